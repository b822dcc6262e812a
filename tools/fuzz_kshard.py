"""Randomised parity sweep of the multi-GPU trainer (host/src/sharded.cpp through include/colibri_sharded.h) on the GPU box: random corpora x random options x 1 / 2 / 4 / 8
ranks (all on device 0: the ranks exchange by device copies; one rank runs RCCL against itself), key-sharded counting wherever the run allows it, the candidate exchange
otherwise — the union of the ranks' exports against the oracle's single-process model of the whole corpus (keys, counts, tokens, types, per-order found / kept).
Not a test (tests/test_kshard.py holds the fixed cases); a tool for hunting corner cases with spare GPU minutes:
    python tools/fuzz_kshard.py --seconds 600 --seed 1 > gpurun_out/fuzz_kshard.json
Every failure is printed with the seed that reproduces it."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import oracle
    from colibri_amd import capi
    from fuzz_parity import make_corpus
    t0 = time.time()
    cases = fails = kshard = 0
    by_world = {1: 0, 2: 0, 4: 0, 8: 0}
    failures = []
    trainers = {}
    case = 0
    while time.time() - t0 < args.seconds:
        seed = args.seed * 1_000_003 + case
        case += 1
        rng = np.random.default_rng(seed)
        payload = make_corpus(rng)
        world = int(rng.choice([1, 2, 2, 4, 4, 8]))
        o = dict(mintokens=int(rng.choice([1, 2, 2, 2, 3, 5])), maxlength=int(rng.choice([1, 2, 3, 4, 5, 5, 6, 9])))
        if rng.integers(0, 4) == 0 and o["mintokens"] >= 2:
            o["mintokens_unigrams"] = o["mintokens"] + int(rng.integers(1, 4))
        try:
            if world not in trainers:  # contexts are reused: a corpus replaces the previous one, as in the product's use
                trainers[world] = capi.ShardedTrainer(world, devices=[0] * world if world > 1 else None)
            tr = trainers[world]
            tr.upload_split(payload)
            st = tr.train(**o)
            got = tr.export_dict()
            want = oracle.train(payload, o["mintokens"], o["maxlength"], mintokens_unigrams=o.get("mintokens_unigrams", 0))
            ok = got == want.counts and (st.totaltokens, st.totaltypes, st.maxn) == (want.tokens, want.types, want.maxn)
            for n in range(1, min(o["maxlength"], 20) + 1):
                ok = ok and (st.found[n], st.kept[n]) == (want.stats[n][0], want.stats[n][2])
            kshard += tr.info.protocol == 0
        except Exception as e:  # noqa: BLE001 — reported with its seed
            ok = False
            failures.append({"seed": seed, "world": world, "options": o, "error": repr(e)[:300]})
            trainers.pop(world, None)
        else:
            if not ok:
                failures.append({"seed": seed, "world": world, "options": o, "error": "model differs from the oracle's"})
        cases += 1
        by_world[world] += 1
        fails += not ok
    for tr in trainers.values():
        tr.close()
    print(json.dumps({"tool": "tools/fuzz_kshard.py", "seed": args.seed, "seconds": round(time.time() - t0, 1), "cases": cases, "key_sharded": kshard, "candidate_exchange": cases - kshard,
                      "by_world": by_world, "failures": fails, "failing_cases": failures[:20]}))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
