#!/bin/bash
# round 6: vp_process_blocks_device on the lane-per-window pipeline, microseconds per block by blocks per call: the multi-block plans forced
# (VP_BOTH_MB_MIN=2) against the default policy (vp_capi.hip v2_mb_min_blocks).  Output: gpurun_out/r6_blocks_per_call.txt
o=gpurun_out/r6_blocks_per_call.txt; : > $o
one() {  # $1 = label (plan|default), rest = bench flags
  lab=$1; shift
  python bench.py --warmup 8 --single-mode --no-cpu --no-parity "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['config']['blocks_per_step']
print('%-8s %-44s %2d blocks per call: %7.1f us per block  %7.3f M frames/s' % (sys.argv[1], sys.argv[2], b, d['ms_per_step']*1e3/b, d['value']/1e6))" "$lab" "$*" >> $o
}
for cfg in "--mode voc --streams 1024" "--mode both --streams 1024" "--cfg5 --mode both --streams 512" "--cfg5 --mode voc --streams 512" "--mode voc --streams 512" "--mode both --streams 512"; do
  for b in 1 2 4 8; do
    VP_BOTH_MB_MIN=2 one plan $cfg --blocks-per-step $b --steps $((160/b))
    one default $cfg --blocks-per-step $b --steps $((160/b))
  done
done
cat $o
