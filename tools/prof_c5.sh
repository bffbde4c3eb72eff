#!/bin/bash
# kernel stats of one C5 bench run under rocprofv3 -> gpurun_out/prof_c5/ (run on the GPU box: gpurun -- 'bash tools/prof_c5.sh')
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_c5${TAG:+_$TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --config c5 --steps 2 --warmup 1 --no-strong-phase --no-cpu-baseline > $OUT/bench.json 2> $OUT/stats.log || exit 1
python3 - <<P
import csv,glob
f=glob.glob("$OUT/stats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:${TOP:-16}]: print(r["Name"][:90].ljust(92), r["Calls"].rjust(6), r["AverageNs"][:8].rjust(10), r["Percentage"][:5])
P
tail -c 400 $OUT/bench.json
