import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "colibri-core_amd", "pyhost"))
from colibri_amd import capi, synth
payload = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    for rep in range(3):
        st = c.train(maxlength=5, mintokens=2, indexed=1)
        print("train ms", round(st.train_ms, 3), flush=True)
