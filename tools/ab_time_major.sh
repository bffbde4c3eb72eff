#!/bin/bash
# TIMING experiment of round 5: tools/libuavppo_A.so = the library with the h <= 128 forward kernel writing its stash / y rows and the
# BPTT kernel reading the stash / writing dgates TIME-major ([t][n] instead of [n][t]) -- -DUAV_TM_PROBE, lstm.hip only.
# Apply tools/experiments/r05_time_major_probe.patch to csrc/lstm.hip first (it is not in the tree: the probe flag would change
# the kernel-source digest the committed PMC profiles carry).  Results of that build are garbage (the weight-gradient kernel
# and everything else still index env-major); only the forward and BPTT launch durations mean anything:
#   UAVPPO_LIB=tools/libuavppo_A.so python tools/perf_update.py        -> profiles/r05_time_major_probe.log
set -e
cd "$(dirname "$0")/../uav-wrf-les-ppo-lstm_amd/csrc"
mkdir -p build_A
for f in *.hip; do
  b=${f%.hip}; fl=""; case $b in lstm) fl="-ffp-contract=fast -DUAV_TM_PROBE";; wgrad) fl="-ffp-contract=fast";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $fl -Wno-unused-function -c $f -o build_A/$b.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build_A/*.o -o ../../tools/libuavppo_A.so
