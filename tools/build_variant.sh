#!/bin/bash
# an experimental build of the device library: bash tools/build_variant.sh NAME -DMACRO=VALUE ...   ->  tools/bin/v_NAME.so (select with COLIBRI_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Icolibri-core_amd/csrc "$@" colibri-core_amd/csrc/colibri_hip.hip -o tools/bin/v_$N.so
echo built tools/bin/v_$N.so
