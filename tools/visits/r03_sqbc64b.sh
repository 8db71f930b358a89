#!/bin/bash
mkdir -p gpurun_out
for v in main sq128; do
  lib=ffpa_attn_amd/libffpa_attn_hip.so; [ $v != main ] && lib=ffpa_attn_amd/variants/libffpa_attn_hip_$v.so
  echo "== library $v"; SMALL_D=1 AUTO_ONLY=1 FFPA_HIP_LIBRARY=$lib timeout 300 python tools/gpu_decode_splits.py 2>&1 | grep "SPLITS\|rror" | tee -a gpurun_out/decode_splits6.txt
done
