#!/bin/bash
# same-box A/B of the round-3 library (built from e23b0cd) against this tree's library: interleaved arms, one process
mkdir -p gpurun_out/v20
export TMPDIR=/tmp
timeout 600 python tools/gpu_ab.py --case cfg2,cfg3,cfg4_mask,cfg4_offset0,causal,cross,gqa,non_aligned,dropout,key_bias,d320,d640,d768,d896,d1024_causal,cross_d1024,gqa_d1024,n2048,n2048_d1024 --rounds 7 --reps 3 r03 main > gpurun_out/v20/ab_vs_r03.txt 2>&1
for sh in "1,32,1,320 --nkv 8192" "8,32,1,320 --hkv 8 --nkv 8192" "1,32,1,512 --nkv 8192" "1,32,1,1024 --nkv 8192"; do
  timeout 120 python tools/gpu_ab.py --shape $sh --rounds 7 --reps 20 r03 main 2>&1 | grep "^AB\|^CASE" >> gpurun_out/v20/ab_vs_r03_decode.txt
done
grep -h "^AB\|^CASE custom" gpurun_out/v20/*.txt
