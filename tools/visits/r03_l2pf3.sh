#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/gpu_ab.py --case d576,d640,d704,d768,d832,d896,d960,cfg3,cross_d1024,n2048_d1024,gqa_d1024,b4_d1024,key_bias_d1024,dense_bias_d1024,dropout_d1024 --rounds 5 --reps 5 main main:0x10 > gpurun_out/l2pf_ab3.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab3.txt
