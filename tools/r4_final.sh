#!/bin/bash
# round 4: the evidence of the final library in one GPU call: the default bench line, the forced-shard lines, rocprofv3 stats + PMC of the plain step, the sharded timeline
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
python bench.py --force-shard --steps 10 --warmup 2 --cpu-sample 0 --no-other-configs > $O/force_shard_100m.json 2> /dev/null
python bench.py --force-shard --tokens 125000000 --steps 10 --warmup 2 --cpu-sample 0 --no-other-configs > $O/force_shard_125m.json 2> /dev/null
grep -o "\"ms_per_step\": [0-9.]*" $O/force_shard_1*.json | head -4
bash tools/profile_bench.sh r04b > /dev/null 2>&1
bash tools/trace_shard.sh 2>&1 | grep -v "fillBuffer\|copyBuffer" > $O/shard_trace.txt
bash tools/trace_step.sh 2>&1 | grep -v "fillBuffer" > $O/step_trace.txt; tail -2 $O/step_trace.txt
