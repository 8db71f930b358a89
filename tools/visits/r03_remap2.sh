#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python tools/gpu_ab.py --case cfg3,d576,d640,d768,d960,gqa_d1024,b4_d1024,n2048_d1024,d1024_causal,key_bias_d1024,dropout_d1024 --rounds 5 --reps 5 main main:0x2 main:0x22 > gpurun_out/remap_ab2.txt 2>&1
grep "^AB\|rror" gpurun_out/remap_ab2.txt
