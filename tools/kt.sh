#!/bin/bash
# usage: tools/kt.sh <tag> <bench args...>   -> gpurun_out/kt_<tag>.txt : per-kernel average durations (rocprofv3 --kernel-trace --stats)
tag=$1; shift
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
d=gpurun_out/kt_$tag; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 40 --warmup 4 --no-cpu --no-parity --single-mode "$@" > $d/bench.json 2> $d/err.txt
f=$(find $d -name '*kernel_stats.csv' | head -1)
python3 - "$f" > gpurun_out/kt_$tag.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith('vp_k') or r['Name'].startswith('void vp_k'):
        print("%-60s calls %5s avg %9.1f us min %9.1f max %9.1f"%(r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -rf $d
cat gpurun_out/kt_$tag.txt
