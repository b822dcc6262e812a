#!/bin/bash
# quick look at the id-keeping modes: timing of the four model kinds, the indexed model's kernel timeline, the full-size tests of those kinds
mkdir -p gpurun_out/idm
timeout 300 python tools/modes_probe.py > gpurun_out/idm/modes.txt 2>&1; cat gpurun_out/idm/modes.txt
bash tools/trace_mode.sh "${1:-indexed=1}" > gpurun_out/idm/trace.txt 2>&1; tail -1 gpurun_out/idm/trace.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "id_keeping" > gpurun_out/idm/fullsize.txt 2>&1; tail -2 gpurun_out/idm/fullsize.txt
