#!/bin/bash
# rocprofv3 matrix-pipe / issue counters (north_star: "rocprof ... MFMA utilisation"): separate --pmc passes, program directly after `--`.
# run on the GPU box from the repo root: bash tools/profile_pmc_sq.sh <tag> <c3|c5|mlp>
set -e
TAG=${1:-r04_a}
WHAT=${2:-c3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_sq_$WHAT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
if [ "$WHAT" = calib ]; then
  $ROOT/tools/bin/mfma_calib > $OUT/calib_plain.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- $ROOT/tools/bin/mfma_calib > $OUT/pmc_mfma.log 2>&1
  python3 $ROOT/tools/pmc_sq_to_json.py $OUT $OUT/pmc_sq.json "$WHAT" > $OUT/pmc_sq_table.txt
  cat $OUT/calib_plain.log $OUT/pmc_sq_table.txt
  exit 0
fi
case $WHAT in
  c3) DRV=$ROOT/tools/pmc_update.py;;
  c5) DRV=$ROOT/tools/pmc_c5.py;;
  mlp) DRV=$ROOT/tools/pmc_mlp.py;;
esac
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 $DRV > $OUT/pmc_mfma.log 2>&1
echo "mfma pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq1 -- python3 $DRV > $OUT/pmc_sq1.log 2>&1
echo "sq pass 1 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq2 -- python3 $DRV > $OUT/pmc_sq2.log 2>&1
echo "sq pass 2 done"
python3 $ROOT/tools/pmc_sq_to_json.py $OUT $OUT/pmc_sq.json "$WHAT" > $OUT/pmc_sq_table.txt
cat $OUT/pmc_sq_table.txt
