#!/bin/bash
# End-of-round evidence for profiles/ (run on the GPU box through gpurun; everything lands under gpurun_out/<tag>/):
#   bash tools/profile_round.sh <tag> [tests]
# 1. rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of bench.py's plain step   -> prof_<tag>/ (tools/profile_bench.sh)
# 2. the same per model kind of configs 4 / 5                                                                         -> prof_<tag>_modes/ (tools/profile_modes.sh)
# 3. forced-shard bench lines at 100 M / 125 M tokens, the plain step's kernel timeline, the forced-shard step's timeline
# 4. (second argument "tests") the whole gpu test suite
cd $GRAFT_REPO_ROOT; T=${1:-r05}; O=gpurun_out/$T; mkdir -p $O
bash tools/profile_bench.sh $T > /dev/null 2>&1
bash tools/profile_modes.sh $T > $O/modes.log 2>&1; grep "train ms" $O/modes.log
python bench.py --force-shard --steps 10 --warmup 2 --cpu-sample 0 --no-other-configs > $O/force_shard_100m.json 2> /dev/null
python bench.py --force-shard --tokens 125000000 --steps 10 --warmup 2 --cpu-sample 0 --no-other-configs > $O/force_shard_125m.json 2> /dev/null
grep -o "\"ms_per_step\": [0-9.]*" $O/force_shard_1*.json | head -4
bash tools/trace_step.sh 2>&1 | grep -v "fillBuffer" > $O/step_trace.txt; tail -1 $O/step_trace.txt
bash tools/trace_shard.sh > $O/shard_trace.txt 2>&1; tail -1 $O/shard_trace.txt
if [ "${2:-}" = "tests" ]; then
  timeout 2700 python -m pytest tests -m gpu -x -q > $O/full_tests.txt 2>&1; echo "rc $?" >> $O/full_tests.txt; grep -n "passed\|failed\|rc " $O/full_tests.txt | tail -3
fi
