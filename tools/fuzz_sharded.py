"""Randomised parity sweep of the sentence-sharded trainer on ONE GPU: `world` ranks run as threads of this process, each with its own
device context on cuda:0, and exchange through an in-process stand-in for torch.distributed (same call signatures as the RCCL
process group the trainer uses: all_to_all_single / all_gather / all_reduce / barrier). The union of the ranks' exports is compared
with the oracle's single-process model of the whole corpus — bit-exact, every mode. A tool, not a test (tests/test_sharded.py holds
the multi-process cases):
    python tools/fuzz_sharded.py --seconds 300 --seed 1 > gpurun_out/fuzz_sharded.json"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


class ThreadDist:
    """torch.distributed look-alike for `world` threads of one process."""

    class ReduceOp:
        SUM, MIN, MAX = "sum", "min", "max"

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slot = [None] * world
        self.local = threading.local()

    def bind(self, rank):
        self.local.rank = rank

    def get_rank(self):
        return self.local.rank

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "threads"

    def barrier(self):
        self.bar.wait()

    def _post(self, item):
        self.slot[self.local.rank] = item
        self.bar.wait()

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        w, r = self.world, self.local.rank
        if in_splits is None:
            in_splits = [inp.numel() // w] * w
        self._post((inp, list(in_splits)))
        o = 0
        for src in range(w):
            t, splits = self.slot[src]
            a = sum(splits[:r])
            n = splits[r]
            out[o: o + n].copy_(t[a: a + n])
            o += n
        import torch
        torch.cuda.synchronize()
        self.bar.wait()

    def all_gather(self, outs, t):
        self._post(t)
        for src in range(self.world):
            outs[src].copy_(self.slot[src])
        import torch
        torch.cuda.synchronize()
        self.bar.wait()

    def all_reduce(self, t, op=None):
        import torch
        self._post(t.clone())
        stack = torch.stack([self.slot[s] for s in range(self.world)])
        res = stack.sum(0) if op == "sum" else (stack.min(0).values if op == "min" else stack.max(0).values)
        t.copy_(res.to(t.dtype))
        torch.cuda.synchronize()
        self.bar.wait()


def run_case(seed, world, payload, o, capi, oracle, torch, cdist):
    shards = cdist.shard_payload(payload, world)
    dist = ThreadDist(world)
    exports, stats, errors = [None] * world, [None] * world, []

    def worker(rank):
        try:
            dist.bind(rank)
            torch.cuda.set_device(0)
            ctx = capi.Context(0)
            try:
                ctx.upload(shards[rank][0], first_sentence=shards[rank][1])
                eng = capi.HipShardEngine(ctx, torch, torch.device("cuda", 0))
                trainer = cdist.ShardedTrainer(eng, dist, torch, torch.device("cuda", 0))
                stats[rank] = trainer.train(capi.Options.defaults(**o))
                exports[rank] = eng.export_local()
            finally:
                ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e)[:300])
            dist.bar.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        return {"seed": seed, "world": world, "options": o, "bytes": len(payload), "error": errors[0]}
    oo = {k: v for k, v in o.items() if k != "table_mode"}
    want = oracle.train(payload, oo.pop("mintokens"), oo.pop("maxlength"), indexed=bool(oo.pop("indexed")), doskipgrams=bool(oo.pop("doskipgrams")),
                        doskipgrams_exhaustive=bool(oo.pop("doskipgrams_exhaustive")), **oo)
    try:
        counts, refs = cdist.merge_exports(exports)
    except ValueError as e:
        return {"seed": seed, "world": world, "options": o, "bytes": len(payload), "error": str(e)}
    st = stats[0]
    ok = counts == want.counts and (refs is None or refs == want.refs) and int(st.totaltokens) == want.tokens and int(st.totaltypes) == want.types
    ok = ok and int(st.maxn) == want.maxn and all((int(st.found[n]), int(st.kept[n])) == (want.stats[n][0], want.stats[n][2]) for n in range(1, min(o["maxlength"], 15) + 1))
    if not ok:
        return {"seed": seed, "world": world, "options": o, "bytes": len(payload), "got": len(counts), "want": len(want.counts)}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    import oracle
    from colibri_amd import capi
    from colibri_amd import dist as cdist
    from fuzz_parity import make_corpus
    t0, case, failures, by = time.time(), 0, [], {}
    while time.time() - t0 < args.seconds:
        seed = args.seed * 7000003 + case
        rng = np.random.default_rng(seed)
        payload = make_corpus(rng)
        world = int(rng.choice([1, 2, 2, 3, 4, 8]))
        mode = int(rng.integers(0, 4))
        o = dict(mintokens=int(rng.choice([2, 2, 3, 4])), maxlength=int(rng.choice([1, 2, 3, 5, 5, 6])), indexed=int(mode in (2, 3)), doskipgrams=int(mode == 3),
                 doskipgrams_exhaustive=int(mode == 1))
        if mode in (1, 3):
            o["minskiptypes"] = int(rng.choice([1, 2, 3]))
        if mode == 1 and rng.integers(0, 2):
            o["mintokens_skipgrams"] = o["mintokens"] + int(rng.integers(0, 3))
        if mode == 0:
            o["table_mode"] = int(rng.choice([0, 0, 1]))
        extra = int(rng.integers(0, 5))
        if extra == 0:  # threshold 1: everything survives everywhere
            o["mintokens"] = 1
            o["maxlength"] = min(o["maxlength"], 5)
            o.pop("mintokens_skipgrams", None)
        elif extra == 1 and o.get("table_mode", 0) == 0:  # secondary word threshold
            o["mintokens_unigrams"] = o["mintokens"] + int(rng.integers(1, 4))
        case += 1
        bad = run_case(seed, world, payload, o, capi, oracle, torch, cdist)
        if bad:
            failures.append(bad)
        key = f"w{world}/" + ("idx" if o["indexed"] else "cnt") + ("+skip" if mode in (1, 3) else "")
        by[key] = by.get(key, 0) + 1
    print(json.dumps({"cases": case, "seconds": round(time.time() - t0, 1), "by_world_and_mode": by, "failures": failures}, indent=1))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
