import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "colibri-core_amd", "pyhost"))
from colibri_amd import capi, synth
payload = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    for mode in ("pipelined", "plain", "pipelined", "plain"):
        if mode == "plain": os.environ["COLIBRI_PLAIN_UPLOAD"] = "1"
        else: os.environ.pop("COLIBRI_PLAIN_UPLOAD", None)
        ts = []
        for rep in range(6):
            t0 = time.perf_counter(); c.upload(payload); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
        print(mode, "upload + tokenise ms", ts, flush=True)
