"""Randomised parity sweep on the GPU box: random corpora x random options, the HIP path through the C ABI against the oracle
(oracle/colibri_oracle.c). Not a test (tests/ hold the fixed cases); a tool for hunting corner cases with spare GPU minutes:
    python tools/fuzz_parity.py --seconds 300 --seed 1 > gpurun_out/fuzz.json
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


def make_corpus(rng):
    from colibri_amd import synth
    kind = rng.integers(0, 6)
    if kind == 0:
        return synth.random_corpus(rng, nsent=int(rng.integers(1, 400)), maxlen=int(rng.integers(1, 20)), vocab=int(rng.integers(1, 40)), big_classes=bool(rng.integers(0, 2)),
                                   empty_rate=float(rng.choice([0.0, 0.1, 0.5])))
    if kind == 1:
        ntok, vocab = int(rng.integers(100, 60000)), int(rng.integers(2, 5000))
        return synth.zipf_corpus(ntok, vocab, int(rng.integers(0, 1 << 30)), phrases=bool(rng.integers(0, 2)) and ntok >= 20000, header=False)
    if kind == 2:  # few distinct tokens, long sentences: every order recurs heavily
        toks = rng.integers(6, 6 + int(rng.integers(1, 4)), size=int(rng.integers(1, 3000))).astype(np.uint32)
        cut = np.sort(rng.choice(toks.size + 1, size=min(toks.size, int(rng.integers(0, 30))), replace=False))
        return synth.encode_v2(np.insert(toks, cut, np.uint32(0))).tobytes() + (b"\x00" if rng.integers(0, 2) else b"")
    if kind == 3:  # class ids around the kernel-choice boundaries
        top = int(rng.choice([(1 << 21) - 1, 1 << 21, (1 << 22) - 1, 1 << 22, (1 << 28) - 1, 70000, 16384]))
        pool = np.array([6, 7, 8, 127, 128, top - 1, top], dtype=np.uint32)
        toks = pool[rng.integers(0, pool.size, size=int(rng.integers(1, 4000)))]
        lens = rng.integers(1, 12, size=toks.size)
        ends = np.cumsum(lens)
        ends = ends[ends < toks.size]
        return synth.encode_v2(np.insert(toks, ends, np.uint32(0))).tobytes() + b"\x00"
    if kind == 4:  # one repeated sentence
        s = rng.integers(6, 12, size=int(rng.integers(1, 14))).astype(np.uint32)
        return (synth.encode_v2(s).tobytes() + b"\x00") * int(rng.integers(1, 40))
    return synth.random_corpus(rng, nsent=int(rng.integers(1, 60)), maxlen=int(rng.integers(1, 6)), vocab=int(rng.integers(1, 5)), big_classes=False, empty_rate=0.3)


def make_options(rng):
    mode = int(rng.integers(0, 6))  # 0 plain, 1 exhaustive skipgrams, 2 indexed, 3 indexed + skipgrams, 4 plain with word threshold / threshold 1, 5 constrained
    o = dict(mintokens=int(rng.choice([2, 2, 2, 3, 5])), maxlength=int(rng.choice([1, 2, 3, 4, 5, 5, 6, 8])), indexed=0, doskipgrams=0, doskipgrams_exhaustive=0)
    if mode == 1:
        o["doskipgrams_exhaustive"] = 1
        o["minskiptypes"] = int(rng.choice([1, 2, 3]))
        if rng.integers(0, 2):
            o["mintokens_skipgrams"] = o["mintokens"] + int(rng.integers(0, 3))
    elif mode == 2:
        o["indexed"] = 1
    elif mode == 3:
        o["indexed"] = 1
        o["doskipgrams"] = 1
        o["minskiptypes"] = int(rng.choice([1, 2, 3]))
    elif mode == 4:
        if rng.integers(0, 2):
            o["mintokens"] = 1
            o["maxlength"] = min(o["maxlength"], 5)
        else:
            o["mintokens_unigrams"] = o["mintokens"] + int(rng.integers(1, 4))
        o["indexed"] = int(rng.integers(0, 2))
    elif mode == 5:
        o["constrained"] = 1
        o["mintokens"] = int(rng.choice([1, 1, 2, 3]))
        o["minlength"] = int(rng.integers(1, o["maxlength"] + 1))
        o["indexed"] = int(rng.integers(0, 2))
    if mode in (1, 3):
        o["maxlength"] = min(o["maxlength"], 6)
        pick = rng.integers(0, 4)
        if pick == 0:  # secondary word threshold together with skipgrams
            o["mintokens_unigrams"] = o["mintokens"] + int(rng.integers(1, 4))
        elif pick == 1:  # threshold 1: every window and every masked form of it is kept
            o["mintokens"] = 1
            o["maxlength"] = min(o["maxlength"], 5)
            o.pop("mintokens_skipgrams", None)
            if rng.integers(0, 2) and o["doskipgrams_exhaustive"]:
                o["mintokens_skipgrams"] = int(rng.integers(1, 4))
    o["table_mode"] = int(rng.choice([0, 0, 0, 1, 2])) if mode == 0 else 0
    if mode in (0, 2) and o["mintokens"] >= 2 and o["table_mode"] != 2 and rng.integers(0, 3) == 0:  # back-off length below the longest pattern
        o["maxbackofflength"] = int(rng.integers(1, max(2, o["maxlength"])))
    return o


def random_constraint(rng, payload, maxlength, oracle):
    toks = [t for t in oracle.key_tokens(payload) if t != b"\x00"]
    keys = set()
    for _ in range(int(rng.integers(0, 600))):
        n = int(rng.integers(1, maxlength + 2))
        if toks and rng.random() < 0.8:
            i = int(rng.integers(0, len(toks)))
            keys.add(b"".join(toks[i:i + n]))
        else:
            keys.add(bytes(int(x) for x in rng.integers(6, 40, size=n)))
    keys.discard(b"")
    if rng.integers(0, 2):  # a prefix-closed set (what a model built with the look-back is): the probe then stops at a position's first miss
        for k in list(keys):
            t = oracle.key_tokens(k)
            for n in range(1, len(t)):
                keys.add(b"".join(t[:n]))
    if not keys:
        keys.add(b"\x06")  # an empty set LIFTS the constraint at the C ABI (the C++ face handles the empty model itself)
    return sorted(keys)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import oracle
    from colibri_amd import capi
    ctx = capi.Context(0)
    t0, case, failures, by_mode = time.time(), 0, [], {}
    while time.time() - t0 < args.seconds:
        seed = args.seed * 1000003 + case
        rng = np.random.default_rng(seed)
        payload = make_corpus(rng)
        o = make_options(rng)
        case += 1
        try:
            ctx.upload(payload)
            if o.pop("constrained", 0):
                if len(payload) > 60000:
                    payload = payload[:60000]
                    ctx.upload(payload)
                keys = random_constraint(rng, payload, o["maxlength"], oracle)
                want = oracle.train_constrained(payload, keys, o["mintokens"], o["maxlength"], o["minlength"], indexed=bool(o["indexed"]))
                try:
                    ctx.set_constraint(keys)
                    st = ctx.train(capi.Options.defaults(**o))
                    cd, rd = ctx.export_dict()
                finally:
                    ctx.set_constraint([])
                if not (cd == want.counts and (rd is None or rd == want.refs) and int(st.totaltokens) == want.tokens):
                    failures.append({"seed": seed, "options": o, "constrained": len(keys), "bytes": len(payload), "got": len(cd), "want": len(want.counts)})
                by_mode["constrained"] = by_mode.get("constrained", 0) + 1
                continue
            st = ctx.train(capi.Options.defaults(**o))
            cd, rd = ctx.export_dict()
            oo = {k: v for k, v in o.items() if k != "table_mode"}
            want = oracle.train(payload, oo.pop("mintokens"), oo.pop("maxlength"), indexed=bool(oo.pop("indexed")), doskipgrams=bool(oo.pop("doskipgrams")),
                                doskipgrams_exhaustive=bool(oo.pop("doskipgrams_exhaustive")), **oo)
            ok = cd == want.counts and (rd is None or rd == want.refs) and int(st.totaltokens) == want.tokens and int(st.totaltypes) == want.types
            ok = ok and int(st.maxn) == want.maxn and all((int(st.found[n]), int(st.pruned[n]), int(st.kept[n])) == want.stats[n] for n in range(1, o["maxlength"] + 1))
            if not ok:
                failures.append({"seed": seed, "options": o, "bytes": len(payload), "got": len(cd), "want": len(want.counts)})
        except Exception as e:  # noqa: BLE001 — a tool: report and go on
            failures.append({"seed": seed, "options": o, "bytes": len(payload), "error": repr(e)[:300]})
        key = ("idx" if o["indexed"] else "cnt") + ("+skip" if o["doskipgrams"] or o["doskipgrams_exhaustive"] else "") + f"/tm{o['table_mode']}"
        by_mode[key] = by_mode.get(key, 0) + 1
    print(json.dumps({"cases": case, "seconds": round(time.time() - t0, 1), "by_mode": by_mode, "failures": failures}, indent=1))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
