#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/gpu_ab.py --case decode,decode_b8,decode_d1024,decode_d128,decode_long,decode_q16 --rounds 7 --reps 20 prev:0x10000 main:0x10000 prev main > gpurun_out/sq_ab2.txt 2>&1
grep "^AB\|rror" gpurun_out/sq_ab2.txt
