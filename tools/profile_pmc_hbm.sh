#!/bin/bash
# HBM byte counters (FETCH_SIZE, WRITE_SIZE in separate rocprofv3 --pmc passes) per kernel for the C5 or MLP configuration.
# run on the GPU box from the repo root: bash tools/profile_pmc_hbm.sh <tag> <c5|mlp>
set -e
TAG=${1:?tag}; WHAT=${2:?c5|mlp}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_hbm_$WHAT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
case $WHAT in
  c5) DRV=$ROOT/tools/pmc_c5.py; export PMC_C5_T=256; LABEL="BASELINE C5 per GPU: 4096 envs x 256 steps, LSTM h=256 x2, obs 6+2, 2 epochs (tools/pmc_c5.py)";;
  mlp) DRV=$ROOT/tools/pmc_mlp.py; LABEL="the reference's MLP policy at C3's shape: 4096 envs x 128 steps, 2 epochs (tools/pmc_mlp.py)";;
esac
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $DRV > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $DRV > $OUT/pmc_write.log 2>&1
echo "write done"
python3 $ROOT/tools/pmc_to_json.py $OUT $OUT/hbm_traffic_pmc.json "$LABEL" > $OUT/hbm_table.txt
cat $OUT/hbm_table.txt
