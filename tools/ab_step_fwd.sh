#!/bin/bash
# ablation builds of the h = 256 step kernels (csrc/lstm_generic.hip, STEP_ABL = 1..5): tools/bin/libuavppo_stepabl<N>.so
set -e
cd "$(dirname "$0")/../uav-wrf-les-ppo-lstm_amd/csrc"
mkdir -p ../../tools/bin build_abl
for n in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DSTEP_ABL=$n -Wno-unused-function -c lstm_generic.hip -o build_abl/lstm_generic_$n.o &
done
wait
for n in 1 2 3 4 5; do
  objs=$(ls build/*.o | grep -v lstm_generic.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build_abl/lstm_generic_$n.o -o ../../tools/bin/libuavppo_stepabl$n.so
done
# cell_bwd_h3_kernel ablations (CELL_ABL = 1..3): tools/bin/libuavppo_cellabl<N>.so
for n in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DCELL_ABL=$n -Wno-unused-function -c lstm_generic.hip -o build_abl/lstm_generic_c$n.o &
done
wait
for n in 1 2 3; do
  objs=$(ls build/*.o | grep -v lstm_generic.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build_abl/lstm_generic_c$n.o -o ../../tools/bin/libuavppo_cellabl$n.so
done
