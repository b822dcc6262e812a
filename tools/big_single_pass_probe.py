"""How far does ONE pass of the second-generation engine go? N x 100 M tokens, plain / indexed / exhaustive skipgrams, with the default slice size (110 M positions: the
plain run in key slices, the id-keeping kinds on the global table) and with COLIBRI_SLICE_POSITIONS raised so that a single pass takes the whole corpus.
usage: [COLIBRI_SLICE_POSITIONS=400000000] python tools/big_single_pass_probe.py [tokens] [kind] [COLIBRI_NO_CHAIN-style env is read by the library]"""
import multiprocessing
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'colibri-core_amd', 'pyhost'))
from colibri_amd import capi, synth  # noqa: E402

KINDS = (("plain", {}), ("indexed", dict(indexed=1)), ("exhaustive skipgrams", dict(doskipgrams_exhaustive=1)))


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
    only = sys.argv[2] if len(sys.argv) > 2 else None
    n = max(1, T // 50_000_000)
    with ProcessPoolExecutor(n, mp_context=multiprocessing.get_context("spawn")) as pool:
        jobs = [pool.submit(synth.zipf_corpus, T // n, 1_000_000, 200 + k, header=False) for k in range(n)]
        payload = np.concatenate([np.frombuffer(j.result(), dtype=np.uint8) for j in jobs])
    print('tokens', T, 'slice positions env', os.environ.get('COLIBRI_SLICE_POSITIONS'), flush=True)
    with capi.Context(0) as c:
        c.upload(payload)
        for name, kw in KINDS:
            if only and name != only:
                continue
            times = []
            for rep in range(2):
                st = c.train(maxlength=5, mintokens=2, **kw)
                times.append(round(st.train_ms, 1))
            print(name, 'train ms', times, 'mode', c.last_mode(with_passes=True), 'patterns', st.npatterns, 'refs', st.nrefs, 'kept', [st.kept[k] for k in range(1, 6)], flush=True)


if __name__ == '__main__':
    main()
