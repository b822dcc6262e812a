#!/bin/bash
# round 4: the multi-GPU trainer with one rank (100 M and 125 M tokens), then the default bench with the 1 B-token extras (eight ranks on this device)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
python bench.py --force-shard --steps 10 --warmup 2 --cpu-sample 0 --no-other-configs > gpurun_out/r4s/force_100m.json 2> gpurun_out/r4s/force_100m.err; tail -c 1500 gpurun_out/r4s/force_100m.json
python bench.py --force-shard --tokens 125000000 --steps 10 --warmup 2 --cpu-sample 0 --no-other-configs > gpurun_out/r4s/force_125m.json 2> gpurun_out/r4s/force_125m.err; tail -c 1500 gpurun_out/r4s/force_125m.json
if [ "$1" = "z1b" ]; then python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/r4s/bench.json 2> gpurun_out/r4s/bench.err; tail -c 6000 gpurun_out/r4s/bench.json; tail -5 gpurun_out/r4s/bench.err; fi
