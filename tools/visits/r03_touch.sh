#!/bin/bash
# Visit: look-ahead touches in the short-query (split-KV) tiles: FFPA_FLAG_L2_PREFETCH (0x10) vs plain, separate merge kernel in both arms (0x10000).
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -x -q -k "decode or split or short" > gpurun_out/touch_pytest.txt 2>&1; tail -3 gpurun_out/touch_pytest.txt
timeout 900 python tools/gpu_ab.py --case decode,decode_b8,decode_d1024,decode_d128,decode_long,decode_q16 --rounds 7 --reps 20 main:0x10000 main:0x10010 > gpurun_out/touch_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/touch_ab.txt
