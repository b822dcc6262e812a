"""The drop-in caller's view (host_selftest bench = PatternModel<uint32_t>::train() through the C++ face) on the bench corpus, with the per-phase host timing:
   python tools/cxx_face_probe.py [tokens]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
from colibri_amd import synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
payload = synth.zipf_corpus(T, 1_000_000, 44, header=False)
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
    path = os.path.join(td, "c.colibri.dat")
    with open(path, "wb") as f:
        f.write(bytes([0xA2, 0x02])); f.write(payload if isinstance(payload, (bytes, bytearray)) else payload.tobytes())
    env = dict(os.environ, COLIBRI_HOST_TIMING="1")
    p = subprocess.run([os.path.join(ROOT, "colibri-core_amd", "bin", "host_selftest"), "bench", path, "5", "2", "4"], capture_output=True, text=True, env=env, timeout=600)
    print(p.stderr[-3000:]); print(p.stdout[-2000:])
