#!/bin/bash
# Visit: Q loads ahead of the mask ranges + the ranges as one batch of scalar loads (main) vs the previous prologue (prev)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bool_mask_gpu.py tests/test_m16_gpu.py -m gpu -x -q > gpurun_out/prologue_pytest.txt 2>&1; tail -2 gpurun_out/prologue_pytest.txt
timeout 900 python tools/gpu_ab.py --case cfg4_mask,cfg4_offset0,cfg4_nomask,mask_d128,cross,n2048,cfg2,cfg3 --rounds 9 --reps 10 prev main > gpurun_out/prologue_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/prologue_ab.txt
