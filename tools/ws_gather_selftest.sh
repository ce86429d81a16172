#!/bin/bash
# The gather pass of vp_k_pitch_ws computes two grains per step in a branch-free form and falls back to the general search when a
# sample is not one of that form's two cases -- which real signals all but never produce.  This builds two side libraries in which
# the fallback is forced (-DWS_GATHER_TEST=1: every pair; =2: one lane of every other trip, i.e. mixed trips) and runs the
# wave-specialised kernel's parity tests and the A/B against the phase kernels on them.
#   here:            bash tools/ws_gather_selftest.sh build
#   on the GPU box:  gpurun -- 'bash tools/ws_gather_selftest.sh run'
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for t in 1 2; do
    VP_EXTRA_HIPCC_FLAGS="-DWS_GATHER_TEST=$t" VP_SPLIT_BUILD=1 VP_LIB_OUT=$PWD/vocoderproject_amd/libvp_gather_test$t.so python -m vocoderproject_amd.build --force > /dev/null
    echo "built vocoderproject_amd/libvp_gather_test$t.so"
  done
else
  for t in 1 2; do
    echo "== WS_GATHER_TEST=$t"
    VP_AMD_LIB=vocoderproject_amd/libvp_gather_test$t.so python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -2
    VP_AMD_LIB=vocoderproject_amd/libvp_gather_test$t.so python tools/ws_ab.py 2>&1 | grep -v amdgpu | tail -7
  done
fi
