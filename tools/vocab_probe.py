"""Scratch probe: the 100 M-token config under larger vocabularies (class-space boundaries of the kernel choices: 2^21 classes for the
class-keyed orders 2 and 3, 4 M classes for the partitioned order 1). Prints train() ms per vocabulary size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'colibri-core_amd', 'pyhost'))
from colibri_amd import capi, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for V in (10**6, 3 * 10**6, 10**7, 5 * 10**7):
    payload = synth.zipf_corpus(T, V, 44, header=False)
    with capi.Context(0) as c:
        c.upload(payload)
        times = []
        for rep in range(3):
            st = c.train(maxlength=5, mintokens=2)
            times.append(round(st.train_ms, 2))
        print('V', V, 'bytes', len(payload), 'maxclass', c.corpus_info()['maxclass'], 'train ms', times, 'kept', [st.kept[n] for n in range(1, 6)], flush=True)
