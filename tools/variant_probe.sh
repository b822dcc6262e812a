#!/bin/bash
# run the perf probe against experimental builds of the library (tools/bin/v_*.so)
for v in "" tools/bin/v_per8.so tools/bin/v_per16.so tools/bin/v_noatomic.so tools/bin/v_noelect.so; do
  echo "=== variant: ${v:-default}"
  COLIBRI_HIP_LIB=${v:+$PWD/$v} python tools/perf_probe.py 100000000 2>&1 | grep -E "train ms|kernels" | head -2
done
