#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh rNN'): rocprofv3 kernel stats of the headline bench,
# the two PMC passes (separately, as the guide prescribes), the other modes' bench lines and the phase stamps.
# Everything lands in gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
python bench.py > $out/${tag}_pitch_cfg2_bench.json 2> $out/bench.err
python bench.py --mode both --no-cpu > $out/${tag}_both_bench.json 2>> $out/bench.err
python bench.py --mode voc --no-cpu > $out/${tag}_voc_bench.json 2>> $out/bench.err
python bench.py --streams 1024 --no-cpu --single-mode > $out/${tag}_pitch_s1024_bench.json 2>> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu --single-mode > $out/${tag}_pitch_cfg2_bench_under_rocprof.json 2> $out/kt.err
bash tools/pmc_traffic.sh $tag > $out/pmc_traffic.txt 2>&1     # FETCH_SIZE / WRITE_SIZE, separate passes, mono and three-channel input
python tools/phase_stamps.py --mode both --iir fast --yin xcorr > $out/${tag}_phase_stamps_both_fast.txt 2>/dev/null
python tools/phase_stamps.py --mode both --iir exact > $out/${tag}_phase_stamps_both_exact.txt 2>/dev/null
find $out -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
ls -la $out
