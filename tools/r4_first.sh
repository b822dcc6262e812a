#!/bin/bash
# round 4, the GPU call after a change to the plain path: [parity of the plain-mode tests,] step time under variants, kernel timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
if [ "$1" = "tests" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_hot_bins.py -m gpu -x -q > gpurun_out/r4a/tests.log 2>&1
  tail -3 gpurun_out/r4a/tests.log
  shift
fi
timeout 600 python tools/plain_variants.py "$@" > gpurun_out/r4a/variants.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r4a/variants.log
timeout 300 bash tools/trace_step.sh > gpurun_out/r4a/trace.log 2>&1; grep -v "fillBuffer\|advance_kernel\|offsets_kernel\|kept_scan\|finish_kernel\|binoff\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r4a/trace.log | tail -45
