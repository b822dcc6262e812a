#!/bin/bash
# round 4, end: rocprofv3 stats + PMC of HEAD's plain step (tag r04f), the forced-shard lines, then the whole gpu suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4final3; mkdir -p $O
bash tools/profile_bench.sh r04f > /dev/null 2>&1
python bench.py --force-shard --steps 10 --warmup 2 --cpu-sample 0 --no-other-configs > $O/force_shard_100m.json 2> /dev/null
python bench.py --force-shard --tokens 125000000 --steps 10 --warmup 2 --cpu-sample 0 --no-other-configs > $O/force_shard_125m.json 2> /dev/null
grep -o "\"ms_per_step\": [0-9.]*" $O/force_shard_1*.json | head -4
bash tools/trace_step.sh 2>&1 | grep -v "fillBuffer" > $O/step_trace.txt; tail -1 $O/step_trace.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/full_tests.txt 2>&1; echo "rc $?" >> $O/full_tests.txt; grep -n "passed\|failed\|rc " $O/full_tests.txt | tail -3
