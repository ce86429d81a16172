#!/bin/bash
# rocprofv3 kernel-trace of one bench configuration; prints our kernels' average durations.  usage: bash tools/kt.sh <tag> <bench args...>
tag=$1; shift
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python3 bench.py --no-cpu --single-mode --steps 40 --warmup 4 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,3), 'M frames/s', round(d['ms_per_step'],4), 'ms/step')"
python3 - <<PY
import csv,glob
f=sorted(glob.glob('gpurun_out/kt_$tag/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'vp_k' in r['Name']:
        print('   %-58s calls %s avg %.1f us min %.1f'%(r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
