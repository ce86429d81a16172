#!/bin/bash
# A/B two builds of the library on the SAME GPU box (box-to-box variance is ~5 %, more than most kernel changes):
#   cp vocoderproject_amd/libvp_amd.so vocoderproject_amd/libvp_A.so   (before the change)
#   ... edit, rebuild ...   cp vocoderproject_amd/libvp_amd.so vocoderproject_amd/libvp_B.so
#   gpurun -- 'bash tools/ab.sh A B [bench.py flags]'
a=$1; b=$2; shift 2
for r in 1 2 3; do for v in $a $b; do
  echo -n "$v: "
  VP_AMD_LIB=vocoderproject_amd/libvp_$v.so python bench.py --no-cpu --single-mode --steps 400 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M frames/s', round(d['roofline']['avg_kernel_us'],1), 'us')"
done; done
