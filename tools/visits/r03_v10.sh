#!/bin/bash
# round 3, GPU visit 10: head dims 64 / 128 on the 16x16x32 kernel with 64-key tiles (variant mind64) vs the 32x32x16 kernel (`main`); D = 192 with 64-key tiles
export AB_ARGS="--rounds 5 --reps 5 --case d64,d128,d64_causal,d128_causal,d64_n2048,d128_n2048,key_bias_d128,dense_bias_d128,dropout_d128,mask_d128,d192 main mind64"
bash tools/gpu_round.sh "ab"
FFPA_HIP_LIBRARY=$PWD/ffpa_attn_amd/variants/libffpa_attn_hip_mind64.so timeout 900 python -m pytest tests/test_fwd_gpu.py tests/test_bool_mask_gpu.py tests/test_reference_suite_gpu.py -q -m gpu -x -k "not twin and not graph and not merged_inside" 2>&1 | tail -4
