#!/bin/bash
# round 6, visit 30: the sorted equal-length ranges above one round, with the jobs spread over the XCDs differently (12 heads on 8 XCDs: 1.5 heads each in the contiguous remap)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
C=c_h8_n4096,c_h6_n4096,c_h5_n4096,r_h9_n4096,r_h10_n4096,r_h12_n4096,r_h6_n8192,r_b3h4_n4096,r_h12_n4096_d320
for f in 0x0 0x2 0x400 0x300; do
echo "EXTRA_FLAGS=$f" | tee -a gpurun_out/r06/v30_xcd.txt
EXTRA_FLAGS=$f TILE_RANGES=1 ONLY=$C ARMS=1,2,3050,3070,3080 timeout 900 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06/v30_xcd.txt
done
