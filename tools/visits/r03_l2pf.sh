#!/bin/bash
# Visit: L2 prefetch two steps ahead (FFPA_FLAG_L2_PREFETCH = 0x10) against the plain launch, same library.
mkdir -p gpurun_out
timeout 1200 python tools/gpu_ab.py --case cross,cfg2,cfg3,cfg4_mask,cfg4_nomask,causal,gqa,d320,d128,n2048,key_bias,dropout --rounds 7 --reps 5 main main:0x10 > gpurun_out/l2pf_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab.txt
