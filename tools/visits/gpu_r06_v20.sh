#!/bin/bash
# round 6, visit 20: KV splits of under-filled PREFILL launches of the packed call: the packed test file, then the sweep of forced counts next to the library's rule
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_varlen_gpu.py -q 2>&1 | grep -v amdgpu.ids | grep -E "^E |^FAILED|passed|failed" | head -60
timeout 1200 python tools/gpu_varlen_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v20_varlen_prefill_splits.txt | grep "^VARLEN" | cut -c1-330
