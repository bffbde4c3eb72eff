#!/bin/bash
# copy one tools/final_round.sh run (gpurun_out/<tag>, gpurun_out/<tag>_sq_{c3,c5}) into profiles/ under <tag>_* (round 5: r05_z;
# the documents' measured tables are regenerated from those files by tools/gen_kernel_table.py <tag>; a re-run after a kernel
# change overwrites them so that every digest matches the tree)
set -e
T=${1:?tag}
bash tools/collect_profiles.sh $T > /dev/null
for c in c2 c3 c4 c5 mlp; do cp gpurun_out/$T/bench_$c.json profiles/${T}_bench_$c.json; done
cp gpurun_out/$T/rehearsal_6rank_gloo_c2.json profiles/${T}_rehearsal_6rank_gloo_c2.json
cp gpurun_out/$T/perf_train_script.log profiles/${T}_perf_train_script.log
for c in c3 c5; do cp gpurun_out/${T}_sq_$c/pmc_sq.json profiles/${T}_pmc_sq_$c.json; cp gpurun_out/${T}_sq_$c/pmc_sq_table.txt profiles/${T}_pmc_sq_${c}_table.txt; done
[ -f gpurun_out/${T}_gpu_tests.log ] && cp gpurun_out/${T}_gpu_tests.log profiles/${T}_gpu_tests.log
python3 tools/gen_kernel_table.py $T > /dev/null
ls profiles/${T}_* | wc -l
