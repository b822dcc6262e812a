import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "colibri-core_amd", "pyhost"))
import numpy as np
from colibri_amd import capi, synth
p = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
print(type(p))
open("/tmp/c.raw", "wb").write(p)
variants = {"synth": p, "file.read": open("/tmp/c.raw", "rb").read(), "np.fromfile": np.fromfile("/tmp/c.raw", dtype=np.uint8), "np.copy": np.frombuffer(p, dtype=np.uint8).copy()}
os.environ["COLIBRI_PLAIN_UPLOAD"] = "1"
with capi.Context(0) as c:
    for name, v in variants.items():
        ts = []
        for rep in range(5):
            t0 = time.perf_counter(); c.upload(v); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
        addr = np.frombuffer(v, dtype=np.uint8).ctypes.data if not isinstance(v, np.ndarray) else v.ctypes.data
        print(name, "addr %% 2MB = %d" % (addr % (2 << 20)), ts, flush=True)
