#!/bin/bash
# round 6, visit 19: paired row tiles per sequence in the packed kernel (PAIR build): bit-identity test, interleaved A/B on 15 packed causal batches
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_varlen_gpu.py -q -x -k "paired or randomized_packed or reference or oracle_dense" 2>&1 | tail -4
timeout 900 python tools/gpu_varlen_pairs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v19_varlen_pairs.txt
