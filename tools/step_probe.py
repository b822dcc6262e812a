"""wall time of colibri_train on the bench corpus, per step, with the library's own train_ms beside it (COLIBRI_HIP_LIB selects the library)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colibri-core_amd", "pyhost"))
from colibri_amd import capi, synth
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
payload = synth.zipf_corpus(tokens, 1_000_000, 44, header=False)
with capi.Context(0) as ctx:
    ctx.upload(payload)
    for prof in (0, 2, 0):
        opt = capi.Options.defaults(mintokens=2, maxlength=5, profile=prof)
        for _ in range(3):
            ctx.train(opt)
        t0 = time.perf_counter(); lib = 0.0
        for _ in range(20):
            st = ctx.train(opt); lib += st.train_ms
        print(os.environ.get("COLIBRI_HIP_LIB", "current"), "profile", prof, "wall ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), "library ms/step %.3f" % (lib / 20), flush=True)
