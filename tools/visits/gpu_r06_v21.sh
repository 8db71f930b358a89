#!/bin/bash
# round 6, visit 21: the packed test file after the prefill-split rule, then the whole GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_varlen_gpu.py -q 2>&1 | grep -v amdgpu.ids | grep -E "^E  .*Error|^FAILED|passed|failed" | cut -c1-400 | head -40
timeout 2000 python -m pytest tests -m gpu -q -x --deselect tests/test_varlen_gpu.py 2>&1 | tail -5
