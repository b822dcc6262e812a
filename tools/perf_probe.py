"""Scratch perf probe (not the bench contract): times colibri_train on Zipf corpora, prints per-kernel-class HIP-event times."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'colibri-core_amd', 'pyhost'))
import numpy as np
from colibri_amd import capi, synth
sizes = [int(x) for x in sys.argv[1:]] or [10**7, 10**8]
MAXLEN = int(os.environ.get('PROBE_MAXLENGTH', '5'))
for T in sizes:
    V, seed = 10**6, 43 if T == 10**7 else 44
    t = time.time()
    if os.environ.get('PROBE_PHRASES'):  # "nphrases,rate": overwrite that share of the stream with copies of a small phrase inventory (hot n-grams)
        nph, rate = os.environ['PROBE_PHRASES'].split(',')
        rng = np.random.default_rng(seed)
        toks = synth.zipf_tokens(T, V, rng); lens = synth.sentence_lengths(T, rng)
        nph = int(nph); plen = rng.integers(3, 9, size=nph); starts = np.concatenate([[0], np.cumsum(plen)]); pool = toks[:int(starts[-1])].copy()
        ninj = int(T * float(rate) / plen.mean()); which = rng.integers(0, nph, size=ninj); where = rng.integers(0, T - 8, size=ninj)
        for k in range(8):  # vectorised injection, token k of every injected phrase
            m = plen[which] > k
            toks[where[m] + k] = pool[starts[which[m]] + k]
        sym = np.insert(toks, np.cumsum(lens), np.uint32(0)); payload = synth.encode_v2(sym).tobytes()
    else:
        payload = synth.zipf_corpus(T, V, seed, header=False)
    print('gen', T, round(time.time()-t, 2), len(payload), flush=True)
    c = capi.Context(0)
    t = time.time(); c.upload(payload); print('upload+tokenise s', round(time.time()-t, 4), c.corpus_info(), flush=True)
    for rep in range(3):
        st = c.train(maxlength=MAXLEN, mintokens=2, profile=1)
        W = sum(st.windows[1:6])
        print('train ms', round(st.train_ms, 3), 'windows', W, 'Mpat/s', round(W/st.train_ms/1e3, 1), 'kept', [st.kept[n] for n in range(1, 6)], 'found', [st.found[n] for n in range(1, 6)], 'adm', [st.admitted[n] for n in range(1, 6)], flush=True)
        print('  kernels', {capi.KERNEL_CLASSES[k]: tuple(round(x, 3) for x in c.kernel_time(k)) for k in range(len(capi.KERNEL_CLASSES))}, flush=True)
    st = c.train(maxlength=MAXLEN, mintokens=2, profile=0)
    print('train ms (no events)', round(st.train_ms, 3), flush=True)
    t = time.time(); a = c.export_arrays(); print('export s', round(time.time()-t, 4), len(a[2]), flush=True)
    c.close()
