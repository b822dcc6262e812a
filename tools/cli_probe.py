"""End-to-end wall time of the drop-in CLIs on the 100 M-token config (file in -> model file out), one MI355X + one host core."""
import os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'colibri-core_amd', 'pyhost'))
from colibri_amd import synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
corpus, model = '/tmp/cli_probe.colibri.dat', '/tmp/cli_probe.colibri.patternmodel'
open(corpus, 'wb').write(synth.zipf_corpus(T, 1_000_000, 44))
cli = os.path.join(ROOT, 'colibri-core_amd', 'bin', 'colibri-patternmodeller')
out = {'tokens': T, 'corpus_bytes': os.path.getsize(corpus), 'runs': {}}
for name, flags in (('unindexed -u', ['-u']), ('indexed', []), ('unindexed + skipgrams -u -s', ['-u', '-s']), ('indexed + skipgrams -s', ['-s']), ('indexed + skipgrams + flexgrams -F S', ['-F', 'S'])):
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([cli, '-f', corpus, '-t', '2', '-l', '5', '-o', model] + flags, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr[-500:]
        best = dt if best is None else min(best, dt)
    out['runs'][name] = {'wall_s': round(best, 2), 'model_bytes': os.path.getsize(model)}
    print(name, out['runs'][name], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
