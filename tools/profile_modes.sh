#!/bin/bash
# rocprofv3 --kernel-trace --stats of one train() per model kind of configs 4 / 5 (100 M tokens): exhaustive skipgrams, indexed, indexed + skipgrams.
# usage (on the GPU box, via gpurun): bash tools/profile_modes.sh <tag>   ->  gpurun_out/prof_<tag>_modes/<kind>_kernel_stats.csv, <kind>_pmc_by_kernel.csv (FETCH_SIZE / WRITE_SIZE in separate passes)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_${1:-r02}_modes; rm -rf $O; mkdir -p $O
for spec in "exhaustive_skipgrams:doskipgrams_exhaustive=1" "indexed:indexed=1" "indexed_skipgrams:indexed=1,doskipgrams=1"; do
  kind=${spec%%:*}; kw=${spec#*:}
  cat > $O/run_$kind.py <<PY
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT/colibri-core_amd/pyhost')
from colibri_amd import capi, synth
payload = synth.zipf_corpus(100_000_000, 1_000_000, 44, header=False)
with capi.Context(0) as c:
    c.upload(payload)
    for rep in range(4):
        st = c.train(maxlength=5, mintokens=2, $kw)
    print('$kind train ms', round(st.train_ms, 2), 'patterns', st.npatterns, 'refs', st.nrefs)
PY
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$kind -o t -- python $O/run_$kind.py > $O/$kind.log 2>&1
  tail -1 $O/$kind.log
  cp $(find $O/$kind -name "*kernel_stats.csv" | head -1) $O/${kind}_kernel_stats.csv
  # HBM bytes per kernel: the TCC counters in their own passes (never combined with a tracing domain)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $O/${kind}_pmc_$ctr -o t -- python $O/run_$kind.py > $O/${kind}_pmc_$ctr.log 2>&1
  done
  python - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/${kind}_pmc_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/${kind}_pmc_by_kernel.csv", "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_sum", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_sum", "WRITE_SIZE_KiB_avg"])
    for k, d in sorted(agg.items(), key=lambda kv: -(sum(kv[1].get("FETCH_SIZE", [0])) + sum(kv[1].get("WRITE_SIZE", [0])))):
        fs, ws = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        n = max(len(fs), len(ws), 1)
        w.writerow([k, n, round(sum(fs), 1), round(sum(fs) / n, 1), round(sum(ws), 1), round(sum(ws) / n, 1)])
PY
done
ls $O
