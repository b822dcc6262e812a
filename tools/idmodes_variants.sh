#!/bin/bash
# the id-keeping kinds' step (tools/idmodes_probe.py) under experimental builds on ONE box: bash tools/idmodes_variants.sh v1 v2 ...   ("-" = the product library)
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = "-" ]; then L=""; else L=$PWD/tools/bin/v_$v.so; fi
  COLIBRI_HIP_LIB=$L python tools/idmodes_probe.py 2>&1 | grep "train ms"
done
