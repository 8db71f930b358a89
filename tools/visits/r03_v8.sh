#!/bin/bash
# round 3, GPU visit 8: the 16x16x32 kernel at head dims 128 ... 256 (variant FFPA_M16_MIN_D=128) vs the 32x32x16 kernel that serves them
export AB_ARGS="--rounds 5 --reps 5 --case d128,d192,d256,d128_causal,d256_causal,d256_n2048,d128_n2048,key_bias_d256,dense_bias_d256,dropout_d256 main mind128"
bash tools/gpu_round.sh "ab"
FFPA_HIP_LIBRARY=$PWD/ffpa_attn_amd/variants/libffpa_attn_hip_mind128.so timeout 900 python -m pytest tests/test_fwd_gpu.py tests/test_bool_mask_gpu.py -q -m gpu -x -k "every_head_dim or boundary or masks_all or gqa or bias or dropout or causal" 2>&1 | tail -4
