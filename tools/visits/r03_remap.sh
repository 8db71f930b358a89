#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python tools/gpu_ab.py --case cfg2,cross,cfg4_mask,n2048,n1024,causal,gqa,cfg3,cross_d1024,decode,decode_b8,decode_long --rounds 5 --reps 8 main:0x10000 main:0x10002 > gpurun_out/remap_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/remap_ab.txt
