#!/bin/bash
# round 6, visit 16: KV splits inside the packed-sequence launch with the fill + balance rule: the packed tests, the sweep against the rule, packed decode vs the dense loop
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_varlen_gpu.py -q 2>&1 | tail -15
timeout 900 python tools/gpu_varlen_splits.py 2>&1 | tee gpurun_out/r06/v16_varlen_splits.txt
timeout 900 python tools/gpu_varlen_decode.py 2>&1 | tee gpurun_out/r06/v16_varlen_decode.txt
