#!/bin/bash
# round 5, visit 1: the wide-row tile — parity first, then the same-box A/B against the 32-row tile and its own tuning variants
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_m16w_gpu.py -x -q > gpurun_out/r05/v1_pytest.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r05/v1_pytest.log
timeout 600 python tools/gpu_ab.py --case cfg4_mask,cfg4_offset0,cfg4_nomask,d320,d320_causal,d320_b3,d320_n2048 --rounds 7 --reps 10 main:0x2000 main:0x1000 wq8:0x1000 wq12:0x1000 wpf6:0x1000 wpf2:0x1000 word1:0x1000 > gpurun_out/r05/v1_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(CASE|AB)" gpurun_out/r05/v1_ab.txt
