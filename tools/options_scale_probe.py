"""Scratch probe: the rarely used options at the 100 M-token config (no pathologies?): MAXBACKOFFLENGTH, pattern list, threshold 1, word threshold."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'colibri-core_amd', 'pyhost'))
from colibri_amd import capi, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
payload = synth.zipf_corpus(T, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    for name, kw in (("plain", {}), ("-b 1", dict(maxbackofflength=1)), ("-b 2", dict(maxbackofflength=2)), ("-b 3", dict(maxbackofflength=3)), ("-W 5", dict(mintokens_unigrams=5)),
                     ("-L", dict(dopatternperline=1, mintokens=1, maxlength=100)), ("-t 1 -l 3", dict(mintokens=1, maxlength=3))):
        o = dict(maxlength=5, mintokens=2)
        o.update(kw)
        times = []
        for rep in range(2):
            t0 = time.perf_counter()
            st = c.train(**o)
            times.append(round((time.perf_counter() - t0) * 1e3, 1))
        print(name, 'ms', times, 'patterns', st.npatterns, 'found', [st.found[n] for n in range(1, 7)], flush=True)
