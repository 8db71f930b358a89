#!/bin/bash
# round 6, visit 24: per-row-tile KV ranges of dense causal launches, the fitted rule: the whole GPU suite, then plan (req 0) vs one range (req 1: pairs / head chunks) vs forced per-row-tile counts
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | grep -E "^E  .*Error|^FAILED|passed|failed" | cut -c1-600 | head -20
C=c_h8_n4096,c_h8_n4096_d128,c_h8_n4096_d320,c_h6_n4096,c_b2h8_n2048,c_b4h8_n1024,c_h16_n2048,c_h4_n8192,c_h4_n4096_d1024,c_h32g4_n1024,c_h8g4_n4096,c_h8_n4096_ctx,c_h5_n4096,c_h7_n4096,c_h3_n8192
TILE_RANGES=1 ONLY=$C ARMS=0,1,2,3 timeout 900 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v24_dense_tile_ranges.txt
