#!/bin/bash
# HBM bytes per kernel of the plain step (tools/perf_probe.py, 100 M tokens): FETCH_SIZE / WRITE_SIZE in separate --pmc passes (no tracing domain combined); per-kernel averages
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_step; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE ${PMC_EXTRA:-}; do
  rocprofv3 --pmc $C --output-format csv -d $O/$C -o p -- python $GRAFT_REPO_ROOT/tools/perf_probe.py 100000000 > $O/$C.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_step"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(O + "/*/p_counter_collection.csv") + glob.glob(O + "/*/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"].split("(")[0].replace("colibri::", "").replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for d in agg.values() for c in d})
print("%-62s %6s " % ("kernel", "n") + " ".join("%14s" % (c + "_MB/launch") for c in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    n = max(len(v) for v in d.values())
    print("%-62s %6d " % (k, n) + " ".join("%14.1f" % (sum(d.get(c, [0])) / max(1, len(d.get(c, [0]))) / 1024) for c in names))
PY
