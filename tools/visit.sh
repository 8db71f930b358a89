#!/bin/bash
mkdir -p gpurun_out/v40
timeout 500 python tools/gpu_ab.py --case cfg2,cfg3,cfg4_nomask,cross,causal --rounds 7 --reps 5 main ntqo > gpurun_out/v40/ab_ntqo.txt 2>&1
grep "^AB" gpurun_out/v40/ab_ntqo.txt
