#!/bin/bash
# round 6, visit 26: causal dense launches of ONE TO TWO rounds of workgroups: the plan (req 0 / 1) next to forced per-row-tile KV ranges
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
C=r_h9_n4096,r_h10_n4096,r_h12_n4096,r_h14_n4096,r_h16_n4096,r_h6_n8192,r_h8_n8192,r_h12_n4096_d320,r_h3_n8192_d1024,r_h24g4_n2048,r_h12_n4096_d128,r_b3h4_n4096
TILE_RANGES=1 ONLY=$C ARMS=0,1,2,3 timeout 900 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v26_two_rounds_tile_ranges.txt
