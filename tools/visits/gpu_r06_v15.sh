#!/bin/bash
# round 6, visit 15: KV splits inside the packed-sequence launch (ABI 6): the packed tests, the split-count sweep on decode-like batches of few long sequences
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_varlen_gpu.py -x -q 2>&1 | tail -15
timeout 900 python tools/gpu_varlen_splits.py 2>&1 | tee gpurun_out/r06/v15_varlen_splits.txt
