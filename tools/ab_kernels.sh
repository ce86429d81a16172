#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of two builds of the library on the SAME box:
#   gpurun -- 'bash tools/ab_kernels.sh A B [bench.py flags]'     (libraries vocoderproject_amd/libvp_A.so / libvp_B.so, see tools/ab.sh)
a=$1; b=$2; shift 2
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
for v in $a $b; do
  d=$PWD/gpurun_out/abk_$v; rm -rf $d
  VP_AMD_LIB=vocoderproject_amd/libvp_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 60 --warmup 4 --no-cpu --no-parity --single-mode "$@" > /dev/null 2> $d.err
  echo "== $v"
  python3 - $d <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = 0.0
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        if not r["Name"].startswith("vp_k") and "vp_k" not in r["Name"]: continue
        print("  %-52s calls %5s  avg %8.1f us" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3)); tot += float(r["AverageNs"]) / 1e3
    print("  sum of averages %.1f us" % tot)
PY
done
