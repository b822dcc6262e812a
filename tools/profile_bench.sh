#!/bin/bash
# Collects the rocprofv3 evidence for bench.py (run on the GPU box via gpurun): kernel-trace stats and, in SEPARATE
# passes (never combined with tracing domains), the TCC PMC counters that give HBM bytes per kernel.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${2:---steps 3 --warmup 1 --cpu-sample 0 --no-other-configs}"  # the plain step only (tools/profile_modes.sh profiles the other model kinds)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
find $OUT -type f | head -40
