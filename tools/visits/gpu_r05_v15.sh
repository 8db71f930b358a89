#!/bin/bash
# round 5, visit 15: the key-bias build of the split-D tiles on the softmax pipeline (ring cache) + the P^T / factor exchange moved into own-only slots — A/B against
# the library of the evidence visit (variant r5a: bit-identity + speed), then the GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1200 python tools/gpu_ab.py --case key_bias_d1024,cfg3,cfg3_mask,d1024_causal,d640,d768,dense_bias_d1024,dropout_d1024,key_bias,cfg2 --rounds 5 --reps 6 r5a main > gpurun_out/r05/v15_ab.txt 2>&1; echo "ab exit $?"; grep -E "^AB" gpurun_out/r05/v15_ab.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05/v15_pytest.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/r05/v15_pytest.log | tail -10
