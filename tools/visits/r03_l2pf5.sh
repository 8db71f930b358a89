#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/gpu_ab.py --case cfg3,d640,d768,d960 --rounds 5 --reps 5 main pfw2:0x10 pfk:0x10 pfv:0x10 > gpurun_out/l2pf_ab5.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab5.txt
