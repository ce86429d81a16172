#!/bin/bash
# Runs on the GPU box: gpurun -- 'bash tools/collect_counters.sh r02'
# For the dominant kernel of every BASELINE config: rocprofv3 --kernel-trace --stats, then PMC passes in runs of their own
# (SQ counters; FETCH_SIZE; WRITE_SIZE -- the TCC pair does not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# each over `python3 bench.py ... --single-mode --no-cpu` (the program directly behind `--`).
# Output: gpurun_out/ctr_<tag>/<config>/{kt,sq,fetch,write}/... ; tools/summarize_counters.py turns it into
# profiles/<tag>_counters.json (what bench.py reads) and copies the kernel-stats csvs.
tag=${1:-r02}
out=$PWD/gpurun_out/ctr_$tag
rm -rf $out          # (a tree merged from several runs would mix their files)
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
SQ2="SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES"
run() {
  name=$1; shift
  mkdir -p $out/$name
  echo "$@" > $out/$name/args.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/kt -- python3 bench.py --steps 40 --warmup 4 --no-cpu --no-parity --single-mode "$@" > $out/$name/bench_under_rocprof.json 2> $out/$name/kt.err
  rocprofv3 --pmc $SQ --output-format csv -d $out/$name/sq -- python3 bench.py --steps 24 --warmup 4 --no-cpu --no-parity --single-mode "$@" > /dev/null 2> $out/$name/sq.err
  rocprofv3 --pmc $SQ2 --output-format csv -d $out/$name/sq2 -- python3 bench.py --steps 24 --warmup 4 --no-cpu --no-parity --single-mode "$@" > /dev/null 2> $out/$name/sq2.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$name/fetch -- python3 bench.py --steps 24 --warmup 4 --no-cpu --no-parity --single-mode "$@" > /dev/null 2> $out/$name/fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$name/write -- python3 bench.py --steps 24 --warmup 4 --no-cpu --no-parity --single-mode "$@" > /dev/null 2> $out/$name/write.err
}
sel=${2:-all}
[ $sel = all -o $sel = cfg2 ] && run cfg2_pitch_s256
# configs[2] AS BENCHED (bench.py's configs2 legs): lpcVoice 24 on the reference's 512/128 window and on the metric's 1024/256 window
[ $sel = all -o $sel = cfg3 ] && run cfg3_voc_s256_lpc24 --mode voc --lpc-voice 24
[ $sel = all -o $sel = cfg3 ] && run cfg3_voc_s256_lpc24_w1024 --mode voc --lpc-voice 24 --voc-window 1024/256
# the +-12-semitone leg of configs[1]
[ $sel = all -o $sel = cfg2s ] && run cfg2_pitch_s256_pm12 --shift 12
# configs[1] with eight queued blocks per call: ONE vp_k_pitch_ws_mb launch covers them (its per-launch figures are per 8 blocks)
[ $sel = all -o $sel = mb8 ] && run cfg2_pitch_s256_mb8 --blocks-per-step 8
[ $sel = all -o $sel = cfg4 ] && run cfg4_both_s1024 --mode both --streams 1024
[ $sel = all -o $sel = cfg5 ] && run cfg5_both_s512 --cfg5 --mode both --streams 512
[ $sel = all -o $sel = cfg2x ] && run cfg2_pitch_s256_exact --iir exact
[ $sel = all -o $sel = stft ] && run stft_s256 --stft-only
[ $sel = all -o $sel = stft32 ] && run stft_s256_f32 --stft-only --stft-precision f32
python3 tools/summarize_counters.py $out $tag
# the summary as this box computed it travels back with gpurun_out/ (copy gpurun_out/ctr_<tag>/_profiles/* into profiles/)
mkdir -p $out/_profiles
cp profiles/${tag}_counters.json profiles/${tag}_*_kernel_stats.csv profiles/${tag}_*_bench_under_rocprof.json $out/_profiles/
