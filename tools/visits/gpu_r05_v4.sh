#!/bin/bash
# round 5, visit 4: the boolean-mask / mask-range build of the split-D tiles (D > 512) on the softmax pipeline — parity, then A/B against the round-4 loop
# (variant library wq8 was linked against the main objects of the other head dims BEFORE this change: its D > 512 kernels are the round-4 ones)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_m16_gpu.py tests/test_bool_mask_gpu.py tests/test_reference_suite_gpu.py -q -x > gpurun_out/r05/v4_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r05/v4_pytest.log
timeout 900 python tools/gpu_ab.py --case cfg3_mask,mask_d1024,d640_mask,d768_mask,d896_mask,d1024_causal --rounds 7 --reps 6 wq8 main > gpurun_out/r05/v4_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(CASE|AB)" gpurun_out/r05/v4_ab.txt
