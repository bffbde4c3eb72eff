#!/bin/bash
# instrumented build of the library (-DUAV_X6_PROFILE): tools/libuavppo_prof.so, used via UAVPPO_LIB=...
set -e
cd "$(dirname "$0")/../uav-wrf-les-ppo-lstm_amd/csrc"
mkdir -p build_prof
for f in adam comm ctx env gae gemm gemm_h3 loss lstm lstm_generic mlp mlp_fused rollout wgrad; do
  fl=""; case $f in lstm|wgrad) fl="-ffp-contract=fast";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $fl -DUAV_X6_PROFILE -Wno-unused-function -c $f.hip -o build_prof/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build_prof/*.o -o ../../tools/libuavppo_prof.so
