#!/bin/bash
# kernel timeline of the last step of the sharded trainer on one rank (bench.py --force-shard): what the exchange machinery costs on top of the counting
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_shard; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/bench.py --force-shard --steps 2 --warmup 1 --cpu-sample 0 --no-other-configs > $O/log.txt 2>&1
tail -1 $O/log.txt
python $GRAFT_REPO_ROOT/tools/trace_step.py $(find $O -name "*kernel_trace.csv" | head -1) ${1:-uni_onepass_kernel}
