#!/bin/bash
# round 5, visit 2: the double-buffered one-barrier tile at every head dim 128 ... 512 (variant builds) against the shipped tiles
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_m16w_gpu.py -x -q > gpurun_out/r05/v2_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r05/v2_pytest.log
timeout 900 python tools/gpu_ab.py --case cfg2,causal,cross,d384,d448,cfg4_mask,cfg4_offset0 --rounds 7 --reps 10 main:0x2000 wall:0x1000 wall3:0x1000 > gpurun_out/r05/v2_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(CASE|AB)" gpurun_out/r05/v2_ab.txt
timeout 900 python tools/gpu_ab.py --case d256,d256_b3,d256_causal,d256_mask,d192,d192_b3,d128,d128_b3 --rounds 7 --reps 10 main:0x2000 wall:0x1000 wall3:0x1000 wrh3:0x1000 > gpurun_out/r05/v2_ab_small.txt 2>&1; echo "ab exit $?"; grep -E "^(CASE|AB)" gpurun_out/r05/v2_ab_small.txt
