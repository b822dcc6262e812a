#!/bin/bash
# indexed + skipgrams with the passes enqueued: parity tests of the skipgram kinds, full-size tests, timing
mkdir -p gpurun_out/idm
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "skip or id_keeping" > gpurun_out/idm/parity.txt 2>&1; tail -3 gpurun_out/idm/parity.txt
timeout 300 python tools/modes_probe.py > gpurun_out/idm/modes.txt 2>&1; cat gpurun_out/idm/modes.txt
