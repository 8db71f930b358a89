#!/bin/bash
mkdir -p gpurun_out/v13
export TMPDIR=/tmp
timeout 300 python tools/gpu_prefill_splits.py > gpurun_out/v13/prefill_splits.txt 2>&1
for sh in "1,32,1,320 --nkv 8192" "8,32,1,320 --hkv 8 --nkv 8192" "1,32,1,448 --nkv 8192" "8,32,1,448 --hkv 8 --nkv 8192" "1,8,1,320 --nkv 65536" "1,32,16,320 --nkv 8192"; do
  timeout 120 python tools/gpu_ab.py --shape $sh --rounds 7 --reps 20 main sq64 2>&1 | grep "^AB\|^CASE" >> gpurun_out/v13/sq64.txt
done
(FFPA_HIP_LIBRARY=$PWD/ffpa_attn_amd/variants/libffpa_attn_hip_sq64.so timeout 600 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k "short_query or decode" 2>&1 | tail -4) > gpurun_out/v13/pytest_sq64.txt
timeout 300 python tools/gpu_ab.py --case cfg3 --rounds 7 --reps 3 main pstep1 pstep3 qstep2 pf2 > gpurun_out/v13/ab_cfg3.txt 2>&1
grep -h "^SPLITS\|^AB\|^CASE\|passed\|failed" gpurun_out/v13/*.txt
