#!/bin/bash
# round 3, GPU visit 2: A/B of the new one-family build against the round-2 library + prologue / K-piece placement variants
export AB_ARGS="--rounds 3 --reps 5 --case cfg2,cfg3,cfg4_mask,cfg4_nomask,causal,cross,n1024,d320,d448,key_bias,dense_bias,dense_bias_f32,dense_bias_heads,dropout,key_bias_d320,dense_bias_d320,key_bias_d1024,dense_bias_d1024,dropout_d320,dropout_d1024 main r2"
export AB2_ARGS="--rounds 3 --reps 5 --case cross,n1024,cfg4_mask,cfg4_nomask,cfg2,d320 main kfirst"
export AB3_ARGS="--rounds 3 --reps 5 --case cfg3,d1024_causal main kpre8 kpre12 kfirst"
bash tools/gpu_round.sh "${STAGES:-ab ab2 ab3}"
