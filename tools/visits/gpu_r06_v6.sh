#!/bin/bash
# round 6, visit 6: paired causal row tiles over more shapes — where does it pay? (the launch rule's data)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 2400 python tools/gpu_ab.py --case causal2k,causal4k,causal_b4_4k,causal,causal_gqa,causal_h8,causal12k,n16k_causal,causal_cross,d320_causal4k,d320_causal,d320_causal16k,d384_causal,d448_causal,d640_causal,d768_causal,d1024_causal4k,d1024_causal_h8,d1024_causal --rounds 5 --reps 8 main:0x20000 main:0x8000 > gpurun_out/r06/v6_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v6_ab.txt
