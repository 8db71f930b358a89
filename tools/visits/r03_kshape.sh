#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/gpu_ab.py --case cfg3,d640,d768 --rounds 5 --reps 5 main ks1 ks2 ks3 > gpurun_out/kshape_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/kshape_ab.txt
