#!/bin/bash
# SQ counters per kernel of the plain step (tools/perf_probe.py, 100 M tokens), in separate --pmc passes of <= 8 SQ counters each (never combined with a tracing domain):
# what binds a kernel — VALU issue, LDS issue, or waiting.   bash tools/pmc_sq.sh [kernel-substring ...]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  env ${PMC_ENV:-} rocprofv3 --pmc $P --output-format csv -d $O/p$i -o p -- python $GRAFT_REPO_ROOT/tools/perf_probe.py ${PROBE_TOKENS:-100000000} > $O/p$i.log 2>&1
done
python - "$@" <<'PY'
import csv, glob, os, sys, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_sq"
want = sys.argv[1:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(O + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("colibri::", "").replace("void ", "")[:56]
        if want and not any(w in k for w in want): continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for d in agg.values() for c in d})
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = max(len(v) for v in d.values())
    print("%s  (n=%d; per-launch means, in millions)" % (k, n))
    print("   " + "  ".join("%s=%.2f" % (c.replace("SQ_", ""), sum(d[c]) / len(d[c]) / 1e6) for c in names if c in d))
PY
