#!/bin/bash
# Visit: short-query tiles with the V(j+1) request ahead of the wait for K(j+1) (main) vs the previous order (prev); separate merge kernel in both arms.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -x -q -k "decode or split or short" > gpurun_out/sq_pytest.txt 2>&1; tail -3 gpurun_out/sq_pytest.txt
timeout 900 python tools/gpu_ab.py --case decode,decode_b8,decode_d1024,decode_d128,decode_long,decode_q16 --rounds 7 --reps 20 prev:0x10000 main:0x10000 > gpurun_out/sq_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/sq_ab.txt
