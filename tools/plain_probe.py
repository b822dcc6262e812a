import sys
sys.path.insert(0, '/root/repo/colibri-core_amd/pyhost')
from colibri_amd import capi, synth
payload = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    ts = []
    for rep in range(8):
        st = c.train(maxlength=5, mintokens=2)
        ts.append(round(st.train_ms, 3))
    st = c.train(maxlength=5, mintokens=2, profile=1)
    print('plain', min(ts), ts, st.npatterns, {capi.KERNEL_CLASSES[k]: round(c.kernel_time(k)[0], 3) for k in range(len(capi.KERNEL_CLASSES)) if c.kernel_time(k)[1]})
