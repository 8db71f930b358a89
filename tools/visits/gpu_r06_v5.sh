#!/bin/bash
# round 6, visit 5: paired causal row tiles (FFPA_FLAG_PAIR_TILES: workgroup i walks row tile n-1-i, then row tile i) — bit-identity first, then the interleaved same-box A/B
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_m16_gpu.py -x -q -k "paired" > gpurun_out/r06/v5_pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r06/v5_pytest.log
timeout 1200 python tools/gpu_ab.py --case causal,d1024_causal,d320_causal,causal4k,d256_causal,d128_causal,n16k_causal --rounds 7 --reps 8 main:0x20000 main:0x8000 > gpurun_out/r06/v5_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(CASE|AB)" gpurun_out/r06/v5_ab.txt
