#!/bin/bash
# round 6: the combined plan with the pitch kernel beside the pipeline's tail (bench.py --overlap) against the sequential plan, same box
o=gpurun_out/r6_ab_ov.txt; : > $o
for rep in 1 2; do
for ov in "--overlap off" "--overlap on" "--overlap auto"; do
  for cfg in "--mode both --streams 1024" "--cfg5 --mode both --streams 512"; do
    r=$(python bench.py --steps 120 --warmup 20 --single-mode --no-cpu --no-parity $cfg $ov 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f M  %.1f us/step  %s'%(d['value']/1e6, d['ms_per_step']*1e3, d['config']['kernel_builds']))")
    echo "overlap='$ov'  $cfg : $r" >> $o
  done
done
done
cat $o
