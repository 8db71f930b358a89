#!/bin/bash
# round 5, visit 8: the pruned kernel headers — full GPU suite, then bit-identity + speed against the library built from the headers before the pruning (variant "pre")
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05/v8_pytest.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/r05/v8_pytest.log | tail -10
timeout 1200 python tools/gpu_ab.py --case cfg2,cfg3,cfg4_mask,causal,cross,d320,d640,d1024_causal,key_bias,dense_bias,dropout,key_bias_d1024,dropout_d1024,d64,d128,decode,decode_d1024,cfg3_mask --rounds 5 --reps 6 pre main > gpurun_out/r05/v8_ab.txt 2>&1; echo "ab exit $?"; grep -E "^AB" gpurun_out/r05/v8_ab.txt
