#!/bin/bash
# round 3, GPU visit 5: (1) in-launch merge with write-through partials vs the merge kernel (decode), (2) computed row offsets vs tables in the
# D = 512 mask / bias / dropout builds, (3) the M0 clobber on the DMA asm
export AB_ARGS="--rounds 7 --reps 20 --case decode,decode_b8 main main:0x10000"
export AB2_ARGS="--rounds 5 --reps 5 --case dense_bias,dense_bias_f32,dropout,causal,cfg2 main rowcalc"
export AB3_ARGS="--rounds 5 --reps 5 --case cfg2,cfg3,d320,cfg4_mask main nom0"
bash tools/gpu_round.sh "ab ab2 ab3"
timeout 300 python -m pytest tests/test_fwd_gpu.py -q -m gpu -k "merged_inside or short_query or hip_graph or decode" 2>&1 | tail -3
