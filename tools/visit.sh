#!/bin/bash
mkdir -p gpurun_out/v19
timeout 300 python tools/gpu_ab.py --case cfg3,d640,d1024_causal --rounds 7 --reps 3 main sm1 sm2 sm4 sm8 > gpurun_out/v19/ab_smpos.txt 2>&1
grep -h "^AB" gpurun_out/v19/*.txt
