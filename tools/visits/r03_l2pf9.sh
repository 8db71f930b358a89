#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/gpu_ab.py --case cross,cfg4_mask,n2048,causal,cfg2 --rounds 7 --reps 10 main pfall:0x10 pfall4:0x10 pfalld4:0x10 > gpurun_out/l2pf_ab9.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab9.txt
