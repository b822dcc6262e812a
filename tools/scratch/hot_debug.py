import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from colibri_amd import capi
from conftest import small_corpora
payload = small_corpora()[sys.argv[1] if len(sys.argv) > 1 else "zipf200k_phrases"]
kw = dict(indexed=1)
if len(sys.argv) > 2: kw["doskipgrams"] = 1
with capi.Context(0) as c:
    c.upload(payload)
    os.environ["COLIBRI_NO_HOT_REFS"] = "1"
    st0 = c.train(mintokens=2, maxlength=5, **kw)
    want, wrefs = c.export_dict()
    del os.environ["COLIBRI_NO_HOT_REFS"]
    for rep in range(12):
        st = c.train(mintokens=2, maxlength=5, **kw)
        got, refs = c.export_dict()
        bad = [k for k in wrefs if refs.get(k) != wrefs[k]]
        print("rep", rep, "nrefs", st.nrefs, st0.nrefs, "counts equal", got == want, "lists differing", len(bad), "of", len(wrefs), flush=True)
        for k in bad[:6]:
            a, b = wrefs[k], refs.get(k)
            print("   key", k.hex(), "len", len(a), len(b) if b is not None else None, "want", a[:4], "got", (b or [])[:4])
        if bad:
            byorder = {}
            for k in bad: byorder[len(k)] = byorder.get(len(k), 0) + 1
            print("   bad by key bytes", byorder)
