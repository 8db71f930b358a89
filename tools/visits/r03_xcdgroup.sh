#!/bin/bash
# Visit: XCDs per head (FFPA_FLAG_XCD_GROUP: 0x100 = 1, 0x200 = 2, 0x300 = 4, 0x400 = 8; no flag = the launch side's rule)
mkdir -p gpurun_out
timeout 1500 python tools/gpu_ab.py --case cfg3,d960,d896,d832,d768,gqa_d1024,d1024_causal,b4_d1024,cfg2,causal,n16k,n16k_causal,n32k_h8,n32k_h8_d1024 --rounds 5 --reps 4 main:0x100 main:0x200 main:0x300 main:0x400 main > gpurun_out/xcdgroup_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/xcdgroup_ab.txt
