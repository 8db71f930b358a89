#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/gpu_ab.py --case cfg4_mask,mask_d128,cfg4_offset0 --rounds 9 --reps 10 prev main > gpurun_out/order_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/order_ab.txt
timeout 600 python -m pytest tests/test_bool_mask_gpu.py -m gpu -x -q 2>&1 | tail -2
