#!/bin/bash
# round 3, GPU visit 13: row tiles of a head walked last-to-first for launches with mask ranges (config 4's explicit causal mask) vs first-to-last
export AB_ARGS="--rounds 7 --reps 10 --case cfg4_mask,cfg4_offset0 main noflip"
bash tools/gpu_round.sh "ab"
timeout 600 python -m pytest tests/test_bool_mask_gpu.py tests/test_m16_gpu.py -q -m gpu -x -k "bounds or ranges or free_range or boolean" 2>&1 | tail -2
