import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "colibri-core_amd", "pyhost"))
import numpy as np
from colibri_amd import capi, synth
p = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
small = p[:4 << 20]
with capi.Context(0) as c, capi.Context(0) as c2:
    c.upload(p)
    for idle in (0.0, 0.02, 0.1, 0.3, 0.1, 0.1):
        ts = []
        for rep in range(5):
            time.sleep(idle)
            t0 = time.perf_counter(); c.upload(p); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
        print("idle", idle, "upload + tokenise ms", ts, flush=True)
    ts = []
    for rep in range(5):  # a small upload on another context first: does it wake the copy path?
        time.sleep(0.1)
        c2.upload(small)
        t0 = time.perf_counter(); c.upload(p); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print("idle 0.1 + 4 MB wake-up upload first:", ts, flush=True)
    ts = []
    for rep in range(5):  # CPU busy instead of sleeping
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 0.1: pass
        t0 = time.perf_counter(); c.upload(p); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print("busy-wait 0.1:", ts, flush=True)
