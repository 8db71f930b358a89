#!/bin/bash
# round 3, GPU visit 3: full GPU suite on the build with the key-bias build (MK = 3) and the in-launch split merge, then A/B numbers
export AB_ARGS="--rounds 3 --reps 5 --case cfg2,key_bias,key_bias_d320,key_bias_d1024,dense_bias,cfg4_mask,decode,decode_b8 main"
bash tools/gpu_round.sh "${STAGES:-pytestall ab decode}"
