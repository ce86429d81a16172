#!/bin/bash
# SQ counters of one bench configuration, per kernel (means per dispatch).  usage: bash tools/sq.sh <tag> <bench args...>
tag=$1; shift
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/sq_$tag -- python3 bench.py --no-cpu --single-mode --steps 24 --warmup 4 "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,statistics
f=sorted(glob.glob('gpurun_out/sq_$tag/*/*counter_collection.csv'))[-1]
acc={}
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    if 'vp_k' in k: acc.setdefault(k,{}).setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k,c in acc.items():
    m={a:statistics.mean(v) for a,v in c.items()}
    print('%-40s waves %6d  valu insts/wave %7.0f  active_valu quad-cyc/inst %.2f  wave_cycles/wave %8.0f  busy %9.0f  valu/wavecyc %.2f wait_any %.2f wait_inst %.2f'%(
      k[:40], m['SQ_WAVES'], m['SQ_INSTS_VALU']/m['SQ_WAVES'], m['SQ_ACTIVE_INST_VALU']/max(m['SQ_INSTS_VALU'],1), m['SQ_WAVE_CYCLES']/m['SQ_WAVES'], m['SQ_BUSY_CYCLES'],
      m['SQ_ACTIVE_INST_VALU']/m['SQ_WAVE_CYCLES'], m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']))
PY
