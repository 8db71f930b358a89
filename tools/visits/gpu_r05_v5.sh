#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_m16_gpu.py tests/test_bool_mask_gpu.py -q -x > gpurun_out/r05/v5_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r05/v5_pytest.log
timeout 900 python tools/gpu_ab.py --case cfg3_mask,mask_d1024,d768_mask,cfg3 --rounds 7 --reps 6 wq8 main > gpurun_out/r05/v5_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(CASE|AB)" gpurun_out/r05/v5_ab.txt
