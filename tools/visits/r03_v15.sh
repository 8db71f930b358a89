#!/bin/bash
# round 3, GPU visit 15: bias preload moved behind the K-fragment prefetch + zero accumulators set inside their branch (main) vs the evidence build (prev)
export AB_ARGS="--rounds 5 --reps 5 --case dense_bias,dense_bias_f32,key_bias,dense_bias_d320,dense_bias_d1024,key_bias_d1024,cfg2 main prev"
bash tools/gpu_round.sh "ab"
timeout 600 python -m pytest tests/test_m16_gpu.py tests/test_bool_mask_gpu.py -q -m gpu -x -k "bias or additive or dropout or mask" 2>&1 | tail -2
