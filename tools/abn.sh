#!/bin/bash
# several builds on the same box: bash tools/abn.sh "v00 v10 v01" <bench flags>
libs=$1; shift
for r in 1 2; do for v in $libs; do
  echo -n "$v $*: "
  VP_AMD_LIB=vocoderproject_amd/libvp_$v.so python bench.py --no-cpu --single-mode --steps 200 "$@" 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M frames/s', {k: round(v,1) for k,v in d['kernel_us'].items()})"
done; done
