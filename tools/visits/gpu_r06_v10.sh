#!/bin/bash
# round 6, visit 10: the exponentials of key step ks + 1 in the MFMA gaps of PV step ks (variant ovl: D >= 320; ovl128: D >= 128) and plain FMAs in the D = 1024 pipeline's gaps
# (variant pfma) — bit-identity (maxdiff vs main must be 0) and speed, interleaved on one box
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1800 python tools/gpu_ab.py --case cfg2,causal,cross,gqa,non_aligned,n2048,key_bias,dense_bias,d320,d384,d448,d320_causal --rounds 7 --reps 8 main ovl ovl128 > gpurun_out/r06/v10_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v10_ab.txt
timeout 600 python tools/gpu_ab.py --case cfg4_mask,cfg4_nomask,cfg4_offset0 --rounds 5 --reps 8 main:0x2000 ovl:0x2000 > gpurun_out/r06/v10_ab_cfg4.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v10_ab_cfg4.txt
timeout 900 python tools/gpu_ab.py --case d256,d192,d128,d256_causal,key_bias_d256 --rounds 5 --reps 8 main ovl128 > gpurun_out/r06/v10_ab_small.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v10_ab_small.txt
timeout 900 python tools/gpu_ab.py --case cfg3,d1024_causal,key_bias_d1024,n2048_d1024 --rounds 7 --reps 6 main pfma > gpurun_out/r06/v10_ab_pfma.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v10_ab_pfma.txt
