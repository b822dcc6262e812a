"""What the last arguments of train() cost at the 100 M-token config (one MI355X): continued training on a model of shorter patterns, and a filter.
   usage: python tools/train_args_probe.py [tokens]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'colibri-core_amd', 'pyhost'))
import numpy as np
from colibri_amd import capi, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
payload = synth.zipf_corpus(T, 1_000_000, 44, header=False)
out = {"tokens": T}
with capi.Context(0) as c:
    c.upload(payload)
    st = c.train(maxlength=5, mintokens=2)
    st = c.train(maxlength=5, mintokens=2)
    out["plain_l5_ms"] = round(st.train_ms, 2)
    full = int(st.npatterns)
    st = c.train(maxlength=2, mintokens=2)
    key_off, key_bytes, counts, _ = c.export_arrays()
    kb = key_bytes.tobytes()
    off = key_off.tolist()
    keys = [kb[off[j]:off[j + 1]] for j in range(len(off) - 1)]
    out["loaded_model_patterns"] = len(keys)
    t0 = time.perf_counter()
    c.set_continuation(keys)
    out["set_continuation_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    for rep in range(2):
        st = c.train(maxlength=5, mintokens=2)
    out["continued_l3_to_l5_ms"] = round(st.train_ms, 2)
    out["continued_new_patterns"] = int(st.npatterns)
    out["continued_total_equals_plain"] = int(st.npatterns) + len(keys) == full
    c.set_continuation([])
    # a filter of three mid-frequency words and one bigram
    flt = [synth.encode_v2(np.array([500], dtype=np.uint32)).tobytes(), synth.encode_v2(np.array([2000], dtype=np.uint32)).tobytes(),
           synth.encode_v2(np.array([6, 77], dtype=np.uint32)).tobytes()]
    c.set_filter(flt)
    for rep in range(2):
        st = c.train(maxlength=5, mintokens=2)
    out["filtered_l5_ms"] = round(st.train_ms, 2)
    out["filtered_patterns"] = int(st.npatterns)
    out["filtered_kept_per_order"] = [int(st.kept[n]) for n in range(1, 6)]
    c.set_filter([])
print(json.dumps(out))
