#!/bin/bash
# round 5, visit 10: the softmax of the split-D tiles shared by rows between the two waves of a row block — parity (it must be bit-identical), then A/B against the
# library from before (variant "pre")
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python tools/gpu_ab.py --case cfg3,d1024_causal,cfg3_mask,d640,d768,d896,cross_d1024,gqa_d1024,n2048_d1024,n32k_h8_d1024,mask_d1024 --rounds 7 --reps 6 pre main > gpurun_out/r05/v10_ab.txt 2>&1; echo "ab exit $?"; grep -E "^AB" gpurun_out/r05/v10_ab.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05/v10_pytest.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/r05/v10_pytest.log | tail -10
