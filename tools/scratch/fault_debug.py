import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(root, "colibri-core_amd", "pyhost"), os.path.join(root, "tests")]
from colibri_amd import capi
from test_gpu_parity import _hot_refs_corpora
payload = _hot_refs_corpora()["zipf_6000"]
with capi.Context(0) as c:
    c.upload(payload)
    for attempt in range(2):
        st = c.train(mintokens=2, maxlength=4, indexed=1)
        print("retries", st.retries, "reason", st.fallback_reason, flush=True)
