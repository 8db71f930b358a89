#!/bin/bash
mkdir -p gpurun_out
echo "== new heuristic (main), all shapes"; timeout 600 python tools/gpu_decode_splits.py 2>&1 | grep "SPLITS\|rror" | tee gpurun_out/decode_splits3.txt
for v in main sq3 sq4; do
  lib=ffpa_attn_amd/libffpa_attn_hip.so; [ $v != main ] && lib=ffpa_attn_amd/variants/libffpa_attn_hip_$v.so
  echo "== small head dims, library $v (workgroups per CU for D < 320: main 2, sq3 3, sq4 4)"; SMALL_D=1 AUTO_ONLY=1 FFPA_HIP_LIBRARY=$lib timeout 300 python tools/gpu_decode_splits.py 2>&1 | grep "SPLITS\|rror" | tee -a gpurun_out/decode_splits3.txt
done
timeout 600 python -m pytest tests/test_fwd_gpu.py -m gpu -x -q -k "decode or split or short or graph" 2>&1 | tail -2
