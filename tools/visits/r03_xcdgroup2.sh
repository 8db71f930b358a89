#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_m16_gpu.py -m gpu -x -q -k "xcd_grouping or l2_prefetch" 2>&1 | tail -2
timeout 900 python tools/gpu_ab.py --case cfg4_mask,cfg4_nomask,d320,d256,mask_d128,cross,cross_d1024,n2048_d1024 --rounds 7 --reps 8 main:0x100 main > gpurun_out/xcdgroup_ab2.txt 2>&1
grep "^AB\|rror" gpurun_out/xcdgroup_ab2.txt
