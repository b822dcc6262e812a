"""ms per plain train() step on the bench corpus under several environment variants, all on ONE box (boxes of the pool differ by +-3 %):
   python tools/plain_variants.py "NAME=VALUE ..." "NAME=VALUE ..." ...     ("-" = no variable).  The corpus is generated once."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
CACHE = "/tmp/plain_variants_corpus.npy"
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from colibri_amd import capi
    payload = np.load(CACHE)
    with capi.Context(0) as ctx:
        ctx.upload(payload)
        opt = capi.Options.defaults(mintokens=2, maxlength=5, profile=0)
        for _ in range(3):
            ctx.train(opt)
        best = 1e9; tot = 0.0
        for _ in range(20):
            st = ctx.train(opt); tot += st.train_ms; best = min(best, st.train_ms)
        print("%-40s mean %.3f ms  best %.3f ms  patterns %d" % (os.environ.get("VARIANT", "-"), tot / 20, best, st.npatterns), flush=True)
    sys.exit(0)
from colibri_amd import synth
tokens = int(os.environ.get("PROBE_TOKENS", "100000000"))
np.save(CACHE, np.frombuffer(synth.zipf_corpus(tokens, 1_000_000, 44, header=False), dtype=np.uint8))
for v in sys.argv[1:] or ["-"]:
    env = dict(os.environ, VARIANT=v)
    if v != "-":
        for kv in v.split():
            k, _, val = kv.partition("=")
            env[k] = val
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env)
