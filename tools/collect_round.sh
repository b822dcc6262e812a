#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/ into profiles/ (run in the build container after the gpurun call):  bash tools/collect_round.sh <tag>
set -e
cd "$(dirname "$0")/.."; T=${1:?tag}
python tools/summarise_profile.py gpurun_out/prof_$T $T 100000000 0 0 | head -1
for k in exhaustive_skipgrams indexed indexed_skipgrams; do
  cp gpurun_out/prof_${T}_modes/${k}_kernel_stats.csv profiles/${T}_${k}_kernel_stats.csv
  cp gpurun_out/prof_${T}_modes/${k}_pmc_by_kernel.csv profiles/${T}_${k}_pmc_by_kernel.csv
done
for f in force_shard_100m.json force_shard_125m.json step_trace.txt shard_trace.txt; do cp gpurun_out/$T/$f profiles/${T}_$f; done
[ -f gpurun_out/${T}_bench.json ] && cp gpurun_out/${T}_bench.json profiles/${T}_bench.json
[ -f gpurun_out/$T/full_tests.txt ] && grep -v "^$" gpurun_out/$T/full_tests.txt | tail -15 > profiles/${T}_gpu_suite.txt
ls profiles | grep "^$T" | tr '\n' ' '
