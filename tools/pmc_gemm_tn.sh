#!/bin/bash
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_gemm_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/perf_gemm_tn.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_old -- python3 $ROOT/tools/perf_gemm_tn.py gemm_tn_off > $OUT/pmc_fetch_old.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_fetch", "pmc_fetch_old"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r04_gemm_pmc/{d}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float); names = {}
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != "FETCH_SIZE": continue
            per[row["Dispatch_Id"]] += float(row["Counter_Value"]); names[row["Dispatch_Id"]] = row["Kernel_Name"]
        for k, v in per.items(): acc[names[k][:60]].append(v)
    for k, v in acc.items():
        if sum(v)/len(v) > 1000: print(d, k, "n=%d" % len(v), "FETCH_SIZE x2 = %.3f GB per launch" % (sum(v)/len(v)*1024*2/1e9))
PY
