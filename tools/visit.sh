#!/bin/bash
mkdir -p gpurun_out/v17
timeout 300 python tools/gpu_ab.py --case cfg2,causal,cross,d320,cfg4_mask,gqa --rounds 9 --reps 3 main pkfma > gpurun_out/v17/ab_pkfma.txt 2>&1
grep -h "^AB" gpurun_out/v17/*.txt
