#!/bin/bash
# kernel durations of the last plain step under an environment variant: bash tools/trace_variant.sh "NAME=VALUE ..." [kernel-name-substring ...]
cd /tmp && export TMPDIR=/tmp
V="$1"; shift
O=$GRAFT_REPO_ROOT/gpurun_out/trace_variant; rm -rf $O; mkdir -p $O
env $V rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/perf_probe.py 100000000 > $O/log.txt 2>&1
echo "== $V"; python $GRAFT_REPO_ROOT/tools/trace_step.py $(find $O -name "*kernel_trace.csv" | head -1) | grep -e "${1:-chain_emit}" -e "kernels "
