#!/bin/bash
# the 1 B-token pin: the new gpu test, then the default bench line (whose other_configs.z1b_* now check themselves against the reference's model)
mkdir -p gpurun_out/z1b
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "one_billion" > gpurun_out/z1b/test.txt 2>&1; echo "test rc $?" >> gpurun_out/z1b/test.txt
tail -5 gpurun_out/z1b/test.txt
timeout 900 python bench.py > gpurun_out/z1b/bench.json 2> gpurun_out/z1b/bench.err; echo "bench rc $?"
tail -c 3000 gpurun_out/z1b/bench.json
