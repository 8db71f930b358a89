#!/bin/bash
# round 6, visit 27: KV ranges of equal LENGTH per row tile (only row tiles longer than a range split): forced lengths next to equal COUNTS (1002, 1003) and the one-range launch,
# on causal launches of half a round to two rounds
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_m16_gpu.py -q -k "causal_kv_ranges" 2>&1 | grep -v amdgpu.ids | grep -E "^E  .*Error|^FAILED|passed|failed" | cut -c1-500 | head
C=c_h8_n4096,c_h8_n4096_d128,c_h8_n4096_d320,c_h6_n4096,c_h5_n4096,c_h7_n4096,c_h4_n8192,c_h3_n8192,c_h8g4_n4096,c_h16_n2048,c_h4_n4096_d1024,c_h8_n4096_ctx,r_h9_n4096,r_h10_n4096,r_h12_n4096,r_h14_n4096,r_h6_n8192,r_h12_n4096_d320,r_h3_n8192_d1024,r_h12_n4096_d128,r_b3h4_n4096,r_h24g4_n2048
TILE_RANGES=1 ONLY=$C ARMS=0,1,1002,2,3,4,6,8 timeout 1500 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v27_range_length.txt
