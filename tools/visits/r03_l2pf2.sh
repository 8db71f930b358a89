#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python tools/gpu_ab.py --case cfg3,d768,d640,d1024_causal --rounds 7 --reps 5 main main:0x10 pfd3:0x10 pfd4:0x10 pfd6:0x10 > gpurun_out/l2pf_ab2.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab2.txt
