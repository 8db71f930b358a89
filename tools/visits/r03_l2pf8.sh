#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/gpu_ab.py --case cfg3,d640 --rounds 5 --reps 5 main:0x20 main pfs4 pfs16 > gpurun_out/l2pf_ab8.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab8.txt
