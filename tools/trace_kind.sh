#!/bin/bash
# per-kernel totals of one train() of a model kind (100 M tokens): bash tools/trace_kind.sh "indexed=1" [rows]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_kind; rm -rf $O; mkdir -p $O
cat > $O/run.py <<PY
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT/colibri-core_amd/pyhost')
from colibri_amd import capi, synth
payload = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    for rep in range(4):
        st = c.train(maxlength=5, mintokens=2, $1)
    print('train ms', round(st.train_ms, 2))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python $O/run.py > $O/log.txt 2>&1
tail -1 $O/log.txt
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/p/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:${2:-24}]:
    print("%-64s calls %4s  %.3f ms/step  avg %.1f us" % (r['Name'].replace('colibri::', '').replace('void ', '')[:64], r['Calls'], float(r['TotalDurationNs']) / 4e6, float(r['AverageNs']) / 1e3))
print("launches per step", sum(int(r['Calls']) for r in rows) / 4)
PY
