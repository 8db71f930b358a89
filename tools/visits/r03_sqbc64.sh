#!/bin/bash
# Visit: 64-key short-query tiles at D = 384 ... 512 (main) vs 32-key (prev)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fwd_gpu.py tests/test_bool_mask_gpu.py tests/test_reference_suite_gpu.py -m gpu -x -q -k "short or decode or split or merge or graph or dropout or Nq1 or nq1 or bias" 2>&1 | tail -3
timeout 600 python tools/gpu_ab.py --case decode,decode_b8,decode_long,decode_q16 --rounds 7 --reps 20 prev:0x10000 main:0x10000 > gpurun_out/sqbc64_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/sqbc64_ab.txt
AUTO_ONLY=1 timeout 300 python tools/gpu_decode_splits.py 2>&1 | grep "SPLITS\|rror" | tee gpurun_out/decode_splits5.txt
