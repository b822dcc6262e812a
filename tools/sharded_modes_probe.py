"""Full-size check of the sentence-sharded trainer in every model kind on ONE MI355X: `world` ranks as threads (tools/fuzz_sharded.py's
in-process process group), each with `tokens / world` tokens of the same corpus, against the single-device run of the whole corpus:
number of patterns, sum of counts, number of references and a checksum over (key bytes, count) must be identical. Prints JSON."""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def digest_arrays(key_off, key_bytes, counts):
    """order-independent checksum of {key bytes: count}: sum of 64-bit hashes of (key, count) records"""
    kb = key_bytes.tobytes()
    off = key_off.tolist()
    acc = 0
    for j, c in enumerate(counts.tolist()):
        h = hashlib.blake2b(kb[off[j]: off[j + 1]] + c.to_bytes(4, "little"), digest_size=8).digest()
        acc = (acc + int.from_bytes(h, "little")) & ((1 << 64) - 1)
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=100_000_000)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--modes", default="plain,us,i,is")
    a = ap.parse_args()
    import torch
    from colibri_amd import capi, synth
    from colibri_amd import dist as cdist
    from fuzz_sharded import ThreadDist
    payload = np.frombuffer(synth.zipf_corpus(a.tokens, 1_000_000, 44, header=False), dtype=np.uint8)
    modes = {"plain": {}, "us": dict(doskipgrams_exhaustive=1), "i": dict(indexed=1), "is": dict(indexed=1, doskipgrams=1)}
    out = {"tokens": a.tokens, "world": a.world, "modes": {}}
    for name in a.modes.split(","):
        kw = modes[name]
        rec = {}
        with capi.Context(0) as c:
            c.upload(payload)
            st = c.train(maxlength=5, mintokens=2, **kw)
            ko, kb, cnt, refs = c.export_arrays()
            rec["single"] = {"patterns": int(cnt.size), "sum_counts": int(cnt.astype(np.uint64).sum()), "refs": int(refs[1].size) if refs else 0,
                             "digest": digest_arrays(ko, kb, cnt), "train_ms": round(st.train_ms, 1)}
            del ko, kb, cnt, refs
        shards = cdist.shard_payload(payload.tobytes(), a.world)
        dist = ThreadDist(a.world)
        parts, errors, times = [None] * a.world, [], [0.0] * a.world

        def worker(rank):
            try:
                dist.bind(rank)
                torch.cuda.set_device(0)
                with capi.Context(0) as ctx:
                    ctx.upload(shards[rank][0], first_sentence=shards[rank][1])
                    eng = capi.HipShardEngine(ctx, torch, torch.device("cuda", 0))
                    trainer = cdist.ShardedTrainer(eng, dist, torch, torch.device("cuda", 0))
                    trainer.train(capi.Options.defaults(maxlength=5, mintokens=2, **kw))  # warm-up: allocations
                    dist.barrier()
                    t0 = time.perf_counter()
                    st = trainer.train(capi.Options.defaults(maxlength=5, mintokens=2, **kw))
                    times[rank] = time.perf_counter() - t0
                    ko, kb, cnt, _ = ctx.export_arrays()
                    nrefs = 0
                    if kw.get("indexed"):
                        import ctypes as C
                        ng, nr = C.c_uint64(), C.c_uint64()
                        ctx._check(ctx.L.colibri_shard_index_sizes(ctx.h, C.byref(ng), C.byref(nr)))
                        nrefs = int(nr.value)
                    parts[rank] = {"patterns": int(cnt.size), "sum_counts": int(cnt.astype(np.uint64).sum()), "refs": nrefs, "digest": digest_arrays(ko, kb, cnt),
                                   "npatterns_stat": int(st.npatterns)}
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e)[:400])
                dist.bar.abort()

        th = [threading.Thread(target=worker, args=(r,)) for r in range(a.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            rec["sharded"] = {"error": errors[0]}
        else:
            rec["sharded"] = {"patterns": sum(p["patterns"] for p in parts), "sum_counts": sum(p["sum_counts"] for p in parts), "refs": sum(p["refs"] for p in parts),
                              "digest": sum(p["digest"] for p in parts) & ((1 << 64) - 1), "step_ms_all_ranks_on_one_gpu": round(max(times) * 1e3, 1)}
            rec["identical"] = all(rec["single"][k] == rec["sharded"][k] for k in ("patterns", "sum_counts", "refs", "digest"))
        out["modes"][name] = rec
        print(name, json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
