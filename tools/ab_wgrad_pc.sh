#!/bin/bash
# ablation builds of csrc/wgrad_pc.hip (PC_ABL = 1..6) linked against the regular objects: tools/bin/libuavppo_pcabl<N>.so
set -e
cd "$(dirname "$0")/../uav-wrf-les-ppo-lstm_amd/csrc"
mkdir -p ../../tools/bin build_abl
for n in ${ABLS:-1 2 3 4}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPC_ABL=$n -Wno-unused-function -c wgrad_pc.hip -o build_abl/wgrad_pc_$n.o &
done
wait
for n in ${ABLS:-1 2 3 4}; do
  objs=$(ls build/*.o | grep -v wgrad_pc.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build_abl/wgrad_pc_$n.o -o ../../tools/bin/libuavppo_pcabl$n.so
done
