#!/bin/bash
# round 6, visit 9: the paired-tile kernel built from the .inc split (the one-tile kernels compile from the original text): parity, then round 5 library ("pre") vs main, main also with pairing off (0x20000)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06 tools/probes/bin
timeout 900 python -m pytest tests/test_m16_gpu.py tests/test_fwd_gpu.py -x -q -k "paired or causal" > gpurun_out/r06/v9_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r06/v9_pytest.log
timeout 1800 python tools/gpu_ab.py --case cfg2,cfg3,cross,dropout,key_bias,causal2k,causal4k,causal,d320_causal4k,d320_causal,causal12k,d1024_causal --rounds 7 --reps 8 pre main main:0x20000 > gpurun_out/r06/v9_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v9_ab.txt
