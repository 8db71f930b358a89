#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_m16_gpu.py -m gpu -x -q -k "l2_prefetch or plan" > gpurun_out/l2pf_pytest.txt 2>&1
tail -4 gpurun_out/l2pf_pytest.txt
timeout 900 python tools/gpu_ab.py --case cfg3,d640,cfg2 --rounds 5 --reps 5 main:0x20 main > gpurun_out/l2pf_ab7.txt 2>&1
grep "^AB\|rror" gpurun_out/l2pf_ab7.txt
