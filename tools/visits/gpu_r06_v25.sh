#!/bin/bash
# round 6, visit 25: UNDER-FILLED dense causal launches: the plan's uniform KV ranges (req 0) next to per-row-tile ranges at forced counts
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
C=u_h4_n4096,u_h2_n4096,u_h2_n8192,u_h1_n8192,u_h8_n2048,u_h4_n4096_d128,u_h2_n4096_d1024,u_h8g4_n2048,u_h3_n4096,u_h4_n2048_ctx
TILE_RANGES=1 ONLY=$C ARMS=0,1,2,3,4,6,8 timeout 900 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v25_underfilled_tile_ranges.txt
