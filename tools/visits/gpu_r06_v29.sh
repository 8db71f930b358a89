#!/bin/bash
# round 6, visit 29: KV ranges of equal LENGTH with the jobs in DESCENDING length (two ranges for row tiles longer than f % of the longest one, none for the others):
# arms 3000 + f next to one range (1), two ranges for every row tile (2: what ships) and the experiment's auto rule (0)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
C=c_h8_n4096,c_h8_n4096_d128,c_h8_n4096_d320,c_h6_n4096,c_h5_n4096,c_h7_n4096,c_h4_n8192,c_h3_n8192,c_h8g4_n4096,c_h16_n2048,c_h4_n4096_d1024,c_h8_n4096_ctx,r_h9_n4096,r_h10_n4096,r_h12_n4096,r_h14_n4096,r_h6_n8192,r_h12_n4096_d320,r_h3_n8192_d1024,r_h12_n4096_d128,r_b3h4_n4096,r_h24g4_n2048
TILE_RANGES=1 ONLY=$C ARMS=0,1,2,3050,3060,3070,3080,3090 timeout 1500 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v29_range_length_sorted.txt
