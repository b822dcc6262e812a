"""ms per train() of the three id-keeping model kinds on the bench corpus (best of four; environment variables select variants): python tools/idmodes_probe.py"""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "colibri-core_amd", "pyhost"))
from colibri_amd import capi, synth
payload = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    for name, kw in (("exhaustive_skipgrams", dict(doskipgrams_exhaustive=1)), ("indexed", dict(indexed=1)), ("indexed_skipgrams", dict(indexed=1, doskipgrams=1))):
        best = 1e9
        for rep in range(4):
            st = c.train(maxlength=5, mintokens=2, **kw); best = min(best, st.train_ms)
        print(name, "train ms", round(best, 3), "patterns", st.npatterns, "refs", st.nrefs, flush=True)
