#!/bin/bash
# Per-GPU documentation figures for the BASELINE configs other than the headline one.
run() { echo -n "$* : "; python bench.py --no-cpu --single-mode --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 |
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M frames/s', round(d['ms_per_step'],3), 'ms/step', {k: round(v,1) for k,v in d['kernel_us'].items()}, d['config']['kernel_builds'])"; }
if [ "$1" = quick ]; then
run --mode voc --streams 256
run --mode both --streams 256
run --mode both --streams 1024
run --cfg5 --mode both --streams 512
run --cfg5 --mode both --streams 512 --voc-path batched
run --cfg5 --mode pitch --streams 512
run --cfg5 --mode voc --streams 256
exit 0
fi
run --mode voc --streams 256
run --mode voc --streams 256 --iir exact
run --mode voc --streams 256 --lpc-voice 24                          # configs[2] as BASELINE.json words it: LPC order 24
run --mode voc --streams 256 --lpc-voice 24 --iir exact
run --mode voc --streams 256 --lpc-voice 24 --voc-window 1024/256    # ... on the metric's 1024-pt / hop-256 window
run --mode both --streams 256
run --mode pitch --streams 1024
run --mode both --streams 1024          # configs[3]: 8192 streams = 1024 per GPU
run --mode both --streams 1024 --iir exact
run --mode both --streams 1024 --voc-path workgroup
run --mode voc --streams 1024
run --mode voc --streams 1024 --voc-path workgroup
run --mode voc --streams 1024 --iir exact
run --mode voc --streams 512
run --mode voc --streams 512 --voc-path workgroup
run --mode voc --streams 384
run --mode voc --streams 384 --voc-path workgroup
run --cfg5 --mode both --streams 512    # configs[4]: 4096 streams = 512 per GPU
run --cfg5 --mode both --streams 512 --voc-path batched
run --cfg5 --mode pitch --streams 512
run --cfg5 --mode voc --streams 512
run --cfg5 --mode voc --streams 512 --voc-path batched
run --cfg5 --mode both --streams 256
# round 3: several blocks per call with both processes on (the combined plan), and the pitch corrector's multi-block builds
run --mode both --streams 256 --three-channel
run --mode both --streams 256 --three-channel --blocks-per-step 8
run --mode both --streams 1024 --three-channel --blocks-per-step 8
run --mode pitch --streams 256 --blocks-per-step 8
run --mode pitch --streams 1024 --blocks-per-step 8
run --mode voc --streams 256 --three-channel --blocks-per-step 8
