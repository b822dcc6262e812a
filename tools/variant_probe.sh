#!/bin/bash
# run the perf probe against experimental builds of the library (tools/bin/v_*.so)
for v in "" "$@"; do
  echo "=== variant: ${v:-default}"
  COLIBRI_HIP_LIB=${v:+$PWD/$v} python tools/perf_probe.py 100000000 2>&1 | grep -E "train ms|kernels" | head -2 | cut -c1-60,330-
done
