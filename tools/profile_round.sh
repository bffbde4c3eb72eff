#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of bench.py, then FETCH_SIZE and WRITE_SIZE in separate passes.
# run on the GPU box from the repo root: bash tools/profile_round.sh <tag>
set -e
TAG=${1:-r03_d}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/pmc_update.py > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/pmc_update.py > $OUT/pmc_write.log 2>&1
echo "write done"
python3 $ROOT/tools/pmc_to_json.py $OUT $OUT/hbm_traffic_pmc.json
# the reference's own policy (MLP, fused kernels) and the h=256 x2 stack (BASELINE C5): kernel statistics only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mlp -o bench -- python3 $ROOT/bench.py --config mlp --no-cpu-baseline > $OUT/bench_mlp_under_rocprof.json 2> $OUT/stats_mlp.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o bench -- python3 $ROOT/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_under_rocprof.json 2> $OUT/stats_c5.log
echo "mlp / c5 stats done"
find $OUT -name "*kernel_stats.csv" | head -3
