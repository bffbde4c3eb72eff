#!/bin/bash
# copy the summaries of one tools/profile_round.sh run (gpurun_out/<tag>/) into profiles/ under the round's naming
set -e
TAG=${1:?tag}
S=gpurun_out/$TAG
cp $S/stats/bench_kernel_stats.csv profiles/${TAG}_bench_c3_kernel_stats.csv
cp $S/bench_under_rocprof.json profiles/${TAG}_bench_c3_under_rocprof.json
cp $S/hbm_traffic_pmc.json profiles/${TAG}_hbm_traffic_pmc.json
[ -f $S/stats_mlp/bench_kernel_stats.csv ] && cp $S/stats_mlp/bench_kernel_stats.csv profiles/${TAG}_bench_mlp_kernel_stats.csv && cp $S/bench_mlp_under_rocprof.json profiles/${TAG}_bench_mlp_under_rocprof.json
[ -f $S/stats_c5/bench_kernel_stats.csv ] && cp $S/stats_c5/bench_kernel_stats.csv profiles/${TAG}_bench_c5_kernel_stats.csv && cp $S/bench_c5_under_rocprof.json profiles/${TAG}_bench_c5_under_rocprof.json
ls -la profiles/${TAG}_*
