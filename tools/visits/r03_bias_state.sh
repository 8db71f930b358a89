#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/gpu_ab.py --case cfg2,key_bias,dense_bias,dense_bias_f32,dense_bias_heads,d320,key_bias_d320,dense_bias_d320,cfg3,key_bias_d1024,dense_bias_d1024 --rounds 5 --reps 5 main > gpurun_out/bias_state.txt 2>&1
grep "^AB\|rror" gpurun_out/bias_state.txt
