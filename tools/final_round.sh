#!/bin/bash
# The round's evidence in one call on the GPU box (from the repo root): bash tools/final_round.sh <tag>
# bench lines of every config, the 8-rank gloo rehearsal, kernel stats + HBM byte counters + SQ counters at the CURRENT kernel sources.
set -e
TAG=${1:-r05_z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$2" != "skip-bench" ]; then
python bench.py --steps 20 --warmup 5 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
echo "c3 done"
for c in mlp c4 c2; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
echo "configs done"
fi
# second pass of a round: `only-bench` after tools/collect_final.sh has copied this round's counters into profiles/, so that the bench
# lines carry `roofline.traffic` of the CURRENT kernel sources (bench.py refuses counters whose digest is older than the tree)
[ "$2" = "only-bench" ] && exit 0
# six ranks: the pool's process guard allows at most 6 processes on a GPU (an 8-rank rehearsal is killed by it)
timeout -k 10 300 python bench.py --gpus 6 --backend gloo --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-exchange-probe > $OUT/rehearsal_6rank_gloo_c2.json 2> $OUT/rehearsal_6rank_gloo_c2.err || echo "6-rank rehearsal failed"
echo "rehearsal done"
python tools/perf_train_script.py 60 > $OUT/perf_train_script.log 2>&1 || echo "train script perf failed"
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
echo "profile round done"
bash tools/profile_pmc_sq.sh $TAG c3 > $OUT/sq_c3.log 2>&1
bash tools/profile_pmc_sq.sh $TAG c5 > $OUT/sq_c5.log 2>&1
echo "sq done"
