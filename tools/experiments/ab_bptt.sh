#!/bin/bash
# ablation / tuning builds of bptt_step_h3_kernel (csrc/lstm_generic.hip): tools/bin/libuavppo_bptt_<tag>.so
# usage: tools/ab_bptt.sh tag "-DBPTT_ABL=1" [tag "defs" ...]
set -e
cd "$(dirname "$0")/../../uav-wrf-les-ppo-lstm_amd/csrc"
mkdir -p ../../tools/bin build_abl
args=("$@")
for ((i = 0; i < ${#args[@]}; i += 2)); do
  tag=${args[i]}; defs=${args[i+1]}
  timeout 600 /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -Wno-unused-function -c lstm_generic.hip -o build_abl/lstm_generic_b_$tag.o &
done
wait
for ((i = 0; i < ${#args[@]}; i += 2)); do
  tag=${args[i]}
  objs=$(ls build/*.o | grep -v lstm_generic.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build_abl/lstm_generic_b_$tag.o -o ../../tools/bin/libuavppo_bptt_$tag.so
done
