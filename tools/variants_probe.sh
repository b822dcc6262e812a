#!/bin/bash
# perf probe (tools/perf_probe.py) against experimental builds on ONE box: bash tools/variants_probe.sh TOKENS v1 v2 ...   (names of tools/bin/v_*.so; "-" = the product library)
T=$1; shift
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = "-" ]; then L=""; else L=$PWD/tools/bin/v_$v.so; fi
  COLIBRI_HIP_LIB=$L python tools/perf_probe.py $T 2>&1 | grep -E "train ms|kernels" | tail -3 | sed -e "s/'tokenise.*'emit'/'emit'/" | cut -c1-400
done
