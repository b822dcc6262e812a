import sys, os, time
sys.path.insert(0, '/root/repo/colibri-core_amd/pyhost')
from colibri_amd import capi, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
payload = synth.zipf_corpus(T, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    for name, kw in (("plain", {}), ("exhaustive skipgrams (unindexed)", dict(doskipgrams_exhaustive=1)), ("indexed", dict(indexed=1)), ("indexed + skipgrams T=2", dict(indexed=1, doskipgrams=1))):
        times = []
        for rep in range(3):
            st = c.train(maxlength=5, mintokens=2, **kw)
            times.append(round(st.train_ms, 1))
        best = min(times)
        print(name, 'train ms', round(best, 1), times, 'patterns', st.npatterns, 'refs', st.nrefs, flush=True)
