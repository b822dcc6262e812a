cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_step; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/perf_probe.py 100000000 > $O/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_step.py $(find $O -name "*kernel_trace.csv" | head -1)
