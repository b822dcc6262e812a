"""A corpus whose bins overflow the count kernels' tables in ONE pass although it is smaller than the single-pass limit: 140 M uniformly drawn tokens over 10^6 types
(nearly every bigram window is its own key: ~1000 distinct keys per final bin, the tables hold 900). The run must notice (Bi2State.overflow 2), repeat with round 3's
pass size (two key slices) and give the table path's model.   usage: COLIBRI_DEBUG_OVERFLOW=1 python tools/overflow_retry_probe.py [tokens]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'colibri-core_amd', 'pyhost'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from colibri_amd import capi, synth  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 140_000_000
    rng = np.random.default_rng(5)
    toks = rng.integers(6, 6 + 1_000_000, size=T, dtype=np.uint32)
    sym = np.insert(toks, np.arange(20, T, 20), np.uint32(0))
    payload = synth.encode_v2(np.append(sym, np.uint32(0)))
    del toks, sym
    from test_gpu_fullsize import row_hashes, summary
    got = {}
    with capi.Context(0) as c:
        c.upload(payload)
        for mode in (0, 1):
            st = c.train(maxlength=5, mintokens=2, table_mode=mode)
            print('table_mode', mode, 'train ms', round(st.train_ms, 1), 'path', c.last_mode(with_passes=True), 'patterns', st.npatterns, 'found', [st.found[n] for n in range(1, 6)],
                  'kept', [st.kept[n] for n in range(1, 6)], flush=True)
            key_off, key_bytes, counts, _ = c.export_arrays()
            got[mode] = (summary(st), [np.sort(h) for h in row_hashes(key_off, key_bytes, counts)])
    same = got[0][0] == got[1][0] and all(np.array_equal(x, y) for x, y in zip(got[0][1], got[1][1]))
    print('SAME MODEL' if same else 'MODELS DIFFER')


if __name__ == '__main__':
    main()
