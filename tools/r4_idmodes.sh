#!/bin/bash
# the id-keeping modes: their full-size tests, the timing of the four model kinds, then the parity tests of those kinds
mkdir -p gpurun_out/idm
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "id_keeping" > gpurun_out/idm/fullsize.txt 2>&1; tail -3 gpurun_out/idm/fullsize.txt
timeout 300 python tools/modes_probe.py > gpurun_out/idm/modes.txt 2>&1; cat gpurun_out/idm/modes.txt
if [ -n "$1" ]; then env $1 timeout 300 python tools/modes_probe.py > gpurun_out/idm/modes_old.txt 2>&1; cat gpurun_out/idm/modes_old.txt; fi
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index or skip" > gpurun_out/idm/parity.txt 2>&1; tail -3 gpurun_out/idm/parity.txt
