#!/bin/bash
# round 6, visit 31: the pair fold (two KV ranges per row tile: the second workgroup of a pair folds the first one's partial; no merge launch): test, then
# one range (1) / two ranges + merge kernel (5002) / two ranges + pair fold (2) / the plan (0)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_m16_gpu.py -q -x -k "causal_kv_ranges" 2>&1 | grep -v amdgpu.ids | grep -E "^E |^FAILED|passed|failed|Error" | cut -c1-600 | head -20
C=c_h8_n4096,c_h8_n4096_d128,c_h8_n4096_d320,c_h6_n4096,c_h5_n4096,c_h7_n4096,c_h4_n8192,c_h3_n8192,c_h8g4_n4096,c_h16_n2048,c_b2h8_n2048,c_h4_n4096_d1024,c_h8_n4096_ctx,r_h9_n4096,r_h12_n4096,r_h6_n8192
TILE_RANGES=1 ONLY=$C ARMS=0,1,5002,2 timeout 900 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v31_pair_fold.txt
