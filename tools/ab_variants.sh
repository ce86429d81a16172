for r in 1 2; do for v in V0 V1 V2 V3; do
  echo -n "$v pitch: "
  VP_AMD_LIB=vocoderproject_amd/libvp_$v.so python bench.py --no-cpu --single-mode --steps 400 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M', round(d['roofline']['avg_kernel_us'],1), 'us')"
done; done
for v in V0 V1 V2 V3; do
  echo -n "$v voc: "
  VP_AMD_LIB=vocoderproject_amd/libvp_$v.so python bench.py --no-cpu --single-mode --steps 200 --mode voc 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M', round(d['roofline']['avg_kernel_us'],1), 'us')"
done
