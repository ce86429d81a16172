#!/bin/bash
# HBM traffic of the headline kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (MI355X_MICROARCH.md), for the
# mono-input bench (default) and the three-channel one.  Output: gpurun_out/pmc_<tag>/{mono,3ch}_{fetch,write}/...
tag=${1:-r01}
out=gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/mono_fetch -- python3 bench.py --steps 40 --warmup 4 --no-cpu --single-mode > /dev/null 2> $out/mono_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/mono_write -- python3 bench.py --steps 40 --warmup 4 --no-cpu --single-mode > /dev/null 2> $out/mono_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/3ch_fetch -- python3 bench.py --steps 40 --warmup 4 --no-cpu --single-mode --three-channel > /dev/null 2> $out/3ch_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/3ch_write -- python3 bench.py --steps 40 --warmup 4 --no-cpu --single-mode --three-channel > /dev/null 2> $out/3ch_write.err
python3 - <<PY
import csv, glob, statistics
for name in ["mono_fetch", "mono_write", "3ch_fetch", "3ch_write"]:
    f = glob.glob("$out/" + name + "/*/*counter_collection.csv")[0]
    rows = list(csv.DictReader(open(f)))
    vals = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"].startswith("vp_k_pitch")]
    print(name, rows[0]["Counter_Name"], len(vals), "mean KB", round(statistics.mean(vals), 1))
PY
