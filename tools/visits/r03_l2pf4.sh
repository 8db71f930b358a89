#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/gpu_ab.py --case cfg3,d640,d768,d960,cfg2,cross --rounds 5 --reps 5 main pfw2:0x10 pfw4:0x10 > gpurun_out/l2pf_ab4.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab4.txt
