#!/bin/bash
# round 6, visit 23: per-row-tile KV ranges of DENSE causal launches of one round or less: the new test, forced counts (uniform ranges: visit 22), the packed kernel's side
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_m16_gpu.py -q -k "causal_kv_ranges or head_chunk or paired" 2>&1 | grep -v amdgpu.ids | grep -E "^E  .*Error|^FAILED|passed|failed" | cut -c1-600 | head -20
C=c_h8_n4096,c_h8_n4096_d128,c_h8_n4096_d320,c_h6_n4096,c_b2h8_n2048,c_b4h8_n1024,c_h16_n2048,c_h4_n8192,c_h4_n4096_d1024,c_h32g4_n1024,c_h8g4_n4096,c_h8_n4096_ctx,c_h5_n4096,c_h7_n4096,c_h3_n8192
TILE_RANGES=1 ONLY=$C ARMS=0,1,2,3,4 timeout 900 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v23_dense_tile_ranges.txt
CASES=one_round timeout 900 python tools/gpu_varlen_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v23_varlen_one_round.txt | grep "^VARLEN" | cut -c1-330
