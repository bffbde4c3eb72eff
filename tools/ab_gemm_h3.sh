#!/bin/bash
# ablation builds of csrc/gemm_h3.hip (H3_ABL = 1..4) linked against the regular objects: tools/bin/libuavppo_h3abl<N>.so
set -e
cd "$(dirname "$0")/../uav-wrf-les-ppo-lstm_amd/csrc"
mkdir -p ../../tools/bin build_abl
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DH3_ABL=$n -Wno-unused-function -c gemm_h3.hip -o build_abl/gemm_h3_$n.o &
done
wait
for n in 1 2 3 4; do
  objs=$(ls build/*.o | grep -v gemm_h3.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build_abl/gemm_h3_$n.o -o ../../tools/bin/libuavppo_h3abl$n.so
done
