#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/gpu_ab.py --case cfg3,d640,d960 --rounds 5 --reps 5 main pfw2:0x10 pfdummy:0x10 > gpurun_out/l2pf_ab6.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab6.txt
