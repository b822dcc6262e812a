"""Constrained training at the benchmark size: train a 100 M-token corpus normally, install that model as the constraint set,
and count a SECOND corpus of the same size constrained by it (threshold 1, as `colibri-patternmodeller -j train.model -t 1` on test data does).
Prints one JSON object; numbers go into DESIGN.md."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
from colibri_amd import capi, synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
train_corpus = synth.zipf_corpus(T, 1_000_000, 44, header=False)
test_corpus = synth.zipf_corpus(T, 1_000_000, 45, header=False)
L = capi.load()
L.colibri_set_constraint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
res = {"tokens": T}
with capi.Context(0) as ctx:
    ctx.upload(train_corpus)
    st = ctx.train(mintokens=2, maxlength=5)
    key_off, key_bytes, counts, _ = ctx.export_arrays()
    res["constraint_patterns"] = int(counts.size)
    ctx.upload(test_corpus)
    t0 = time.perf_counter()
    assert L.colibri_set_constraint(ctx.h, key_off.ctypes.data_as(C.c_void_p), key_bytes.ctypes.data_as(C.c_void_p), C.c_uint64(counts.size)) == 0
    res["set_constraint_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    best = 1e9
    for rep in range(3):
        st = ctx.train(mintokens=1, maxlength=5)
        best = min(best, st.train_ms)
    res.update({"constrained_train_ms": round(best, 2), "patterns_found": int(st.npatterns), "kept_per_order": [int(st.kept[n]) for n in range(1, 6)],
                "windows_scanned": int(sum(st.windows[1:6])), "G_windows_per_s": round(sum(st.windows[1:6]) / best / 1e6, 1)})
    st2 = ctx.train(mintokens=1, maxlength=5, indexed=1)
    res["constrained_indexed_train_ms"] = round(st2.train_ms, 2)
    res["refs"] = int(st2.nrefs)
print(json.dumps(res))
