import sys, os, time
sys.path.insert(0, '/root/repo/colibri-core_amd/pyhost')
import numpy as np
from colibri_amd import capi, synth
T = 100_000_000
rng = np.random.default_rng(3)
lens = synth.sentence_lengths(T, rng)
for name, toks in (("all the same token", np.full(T, 6, dtype=np.uint32)),
                   ("two alternating tokens", (np.arange(T) % 2 + 6).astype(np.uint32)),
                   ("period-7 cycle", (np.arange(T) % 7 + 6).astype(np.uint32)),
                   ("all distinct tokens (V = T)", (np.arange(T) + 6).astype(np.uint32))):
    sym = np.insert(toks, np.cumsum(lens), np.uint32(0))
    payload = synth.encode_v2(sym).tobytes()
    with capi.Context(0) as c:
        c.upload(payload)
        for rep in range(2):
            st = c.train(maxlength=5, mintokens=2)
        print(name, 'train ms', round(st.train_ms, 2), 'kept', [st.kept[n] for n in range(1, 6)], 'found', [st.found[n] for n in range(1, 6)], flush=True)
