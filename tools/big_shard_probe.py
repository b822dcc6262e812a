"""1B-token single-device run: the sliced radix path (default) and, with a second argument, the global-table path; validates the 32-bit sizing limits and the HBM budget on a 288 GB MI355X.
   usage: big_shard_probe.py [tokens] [table_mode]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'colibri-core_amd', 'pyhost'))
import numpy as np
from colibri_amd import capi, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10**9
parts = []
t = time.time()
for k in range(T // 10**8):  # generate in 100M-token pieces to bound host memory
    parts.append(np.frombuffer(synth.zipf_corpus(10**8, 10**6, 100 + k, header=False), dtype=np.uint8))
payload = np.concatenate(parts); del parts
print('gen s', round(time.time() - t, 1), 'bytes', payload.size, flush=True)
c = capi.Context(0)
t = time.time(); c.upload(payload); print('upload+tokenise s', round(time.time() - t, 2), c.corpus_info(), flush=True)
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for rep in range(2):
    st = c.train(maxlength=5, mintokens=2, profile=1, table_mode=MODE)
    W = sum(st.windows[1:6])
    print('mode/passes', c.last_mode(True), 'train ms', round(st.train_ms, 1), 'Mpat/s', round(W / st.train_ms / 1e3, 1), 'kept', [st.kept[n] for n in range(1, 6)], 'patterns', st.npatterns, flush=True)
    print('  kernels', {capi.KERNEL_CLASSES[k]: tuple(round(x, 2) for x in c.kernel_time(k)) for k in range(len(capi.KERNEL_CLASSES)) if c.kernel_time(k)[1]}, flush=True)
c.close()
