run() { echo -n "$* : "; "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), {k: round(v,1) for k,v in d['kernel_us'].items()})"; }
B="python bench.py --no-cpu --single-mode --steps 100 --warmup 10"
for r in 1 2; do
run $B --mode both --streams 1024
run $B --mode both --streams 1024 --no-overlap
run env VP_V2_NO_SIDE=1 $B --mode both --streams 1024
run env VP_V2_NO_SIDE=1 $B --mode both --streams 1024 --no-overlap
run $B --cfg5 --mode both --streams 512
run $B --cfg5 --mode both --streams 512 --no-overlap
run env VP_V2_NO_SIDE=1 $B --cfg5 --mode both --streams 512
run env VP_V2_NO_SIDE=1 $B --cfg5 --mode both --streams 512 --no-overlap
run $B --mode voc --streams 1024
run env VP_V2_NO_SIDE=1 $B --mode voc --streams 1024
run $B --cfg5 --mode voc --streams 512
run env VP_V2_NO_SIDE=1 $B --cfg5 --mode voc --streams 512
done
