#!/bin/bash
# round 3, GPU visit 9: 64-key tiles instead of 128 at head dims 128 ... 320 on the 16x16x32 kernel (variant bc64; D = 128 of `main` is the 32x32x16 kernel)
export AB_ARGS="--rounds 5 --reps 5 --case d128,d192,d256,d320,cfg4_mask,cfg4_nomask,d256_causal,d256_n2048,d128_n2048 main bc64"
bash tools/gpu_round.sh "ab"
