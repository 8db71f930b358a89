#!/bin/bash
# round 5, visit 3: full GPU suite on the library with the wide-row tile behind its launch rule + the rule's A/B
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05/v3_pytest.log 2>&1; echo "pytest exit $?"; tail -25 gpurun_out/r05/v3_pytest.log
timeout 900 python tools/gpu_ab.py --case cfg4_mask,cfg4_offset0,cfg4_nomask,d320,d320_causal,d320_gqa --rounds 9 --reps 10 main:0x2000 main main:0x1000 > gpurun_out/r05/v3_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(CASE|AB)" gpurun_out/r05/v3_ab.txt
