#!/bin/bash
# round 6, visit 33: the compact grid of ragged packed prefill batches: the packed test file, then the bench batch and more ragged ones with / without it
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_varlen_gpu.py -q 2>&1 | grep -v amdgpu.ids | grep -E "^E  .*Error|^FAILED|passed|failed" | cut -c1-500 | head -20
timeout 900 python tools/gpu_varlen_compact.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v33_compact_grid.txt
