#!/bin/bash
# per-kernel totals of the indexed model with a variant library: bash tools/scratch/trace_lib.sh <lib or ""> [rows] [grep]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_lib; rm -rf $O; mkdir -p $O
[ -n "$1" ] && export COLIBRI_HIP_LIB=$GRAFT_REPO_ROOT/$1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python $GRAFT_REPO_ROOT/tools/scratch/kprof_indexed.py > $O/log.txt 2>&1
tail -1 $O/log.txt
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/p/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:${2:-12}]:
    print("%-50s calls %4s  %.3f ms/step" % (r['Name'].replace('colibri::', '').replace('void ', '')[:50], r['Calls'], float(r['TotalDurationNs']) / 3e6))
for r in rows:
    if 'hot' in r['Name'] or 'emit_' in r['Name']: print("   %-50s calls %4s  %.3f ms/step" % (r['Name'].replace('colibri::', '')[:50], r['Calls'], float(r['TotalDurationNs']) / 3e6))
PY
