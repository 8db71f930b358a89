#!/bin/bash
# round 5, visit 12: the row-shared softmax in the generic split-D loop (additive bias / dropout / D % 128 == 64 builds) — A/B against the library of the round's
# start of day (variant "pre": bit-identity + speed), ND = 1 controls, then the GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1200 python tools/gpu_ab.py --case key_bias_d1024,dense_bias_d1024,dropout_d1024,d576,d704,d832,d960,cfg2,dropout,key_bias,dense_bias,cfg4_mask --rounds 5 --reps 6 pre main > gpurun_out/r05/v12_ab.txt 2>&1; echo "ab exit $?"; grep -E "^AB" gpurun_out/r05/v12_ab.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05/v12_pytest.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/r05/v12_pytest.log | tail -10
