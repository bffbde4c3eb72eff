#!/bin/bash
# A/B build for the gate-activation change of round 5: tools/libuavppo_A.so = the library with the exp-only tanh
# (-DUAV_TANH_LEGACY), used via UAVPPO_LIB=tools/libuavppo_A.so against the in-tree library (tools/ab_update.py, bench.py).
set -e
cd "$(dirname "$0")/../uav-wrf-les-ppo-lstm_amd/csrc"
mkdir -p build_A
for f in *.hip; do
  b=${f%.hip}; fl=""; case $b in lstm|wgrad) fl="-ffp-contract=fast";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $fl -DUAV_TANH_LEGACY -Wno-unused-function -c $f -o build_A/$b.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build_A/*.o -o ../../tools/libuavppo_A.so
