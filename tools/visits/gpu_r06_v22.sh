#!/bin/bash
# round 6, visit 22: DENSE causal launches of one round or less: forced KV-split counts next to the plan's own (pairs / head chunks / uniform splits)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
ONLY=c_h8_n4096,c_h8_n4096_d128,c_h8_n4096_d320,c_h6_n4096,c_b2h8_n2048,c_b4h8_n1024,c_h16_n2048,c_h4_n8192,c_h4_n4096_d1024,c_h32g4_n1024,c_h8g4_n4096,c_h8_n4096_ctx,c_h5_n4096,c_h7_n4096,c_h3_n8192,h5_n4096_causal ARMS=0,1,2,3 timeout 900 python tools/gpu_prefill_splits.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/v22_dense_causal_one_round.txt
