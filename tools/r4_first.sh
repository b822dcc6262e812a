#!/bin/bash
# round 4, first GPU call of a change to the plain path: parity of the plain-mode tests, step time with / without the chained orders, kernel timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_hot_bins.py -m gpu -x -q > gpurun_out/r4a/tests.log 2>&1
tail -5 gpurun_out/r4a/tests.log
timeout 300 python tools/step_probe.py > gpurun_out/r4a/step.log 2>&1; cat gpurun_out/r4a/step.log
COLIBRI_NO_CHAIN=1 timeout 300 python tools/step_probe.py > gpurun_out/r4a/step_nochain.log 2>&1; cat gpurun_out/r4a/step_nochain.log
timeout 300 bash tools/trace_step.sh > gpurun_out/r4a/trace.log 2>&1; tail -90 gpurun_out/r4a/trace.log
