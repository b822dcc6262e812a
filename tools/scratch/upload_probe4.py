import os, sys, time, threading
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "colibri-core_amd", "pyhost"))
import numpy as np
from colibri_amd import capi, synth
p = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
def timed_upload(c):
    t0 = time.perf_counter(); c.upload(p); return round((time.perf_counter() - t0) * 1e3, 2)
def threads_burst():
    def work():
        a = np.random.default_rng(1).integers(0, 1 << 30, size=2_000_000); a.sort()
    th = [threading.Thread(target=work) for _ in range(32)]
    [t.start() for t in th]; [t.join() for t in th]
with capi.Context(0) as c:
    c.upload(p)
    print("baseline", [timed_upload(c) for _ in range(4)], flush=True)
    ts = []
    for rep in range(4):
        c.train(maxlength=5, mintokens=2); ts.append(timed_upload(c))
    print("after train", ts, flush=True)
    ts = []
    for rep in range(4):
        c.train(maxlength=5, mintokens=2); arrs = c.export_arrays(); ts.append(timed_upload(c)); del arrs
    print("after train + export_arrays", ts, flush=True)
    ts = []
    for rep in range(4):
        threads_burst(); ts.append(timed_upload(c))
    print("after 32 host threads", ts, flush=True)
    ts = []
    for rep in range(4):
        big = np.zeros(64_000_000, dtype=np.uint32); big[::1024] = 1; del big; ts.append(timed_upload(c))
    print("after 256 MB alloc/free", ts, flush=True)
