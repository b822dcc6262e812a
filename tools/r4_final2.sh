#!/bin/bash
# round 4, second half: the default bench line of the final library (other_configs with the id-keeping modes on the chained engine), rocprofv3 stats + PMC of the
# three id-keeping model kinds, the indexed model's kernel timeline
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4final2; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
bash tools/profile_modes.sh r04d > $O/profile_modes.log 2>&1; tail -4 $O/profile_modes.log
bash tools/trace_mode.sh "indexed=1" > $O/indexed_trace.txt 2>&1; tail -1 $O/indexed_trace.txt
