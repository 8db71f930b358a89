#!/bin/bash
# Visit: KV-split partials merged inside the launch with all splits of a row tile on one XCD (main) vs the separate merge kernel (main:0x10000, prev:0x10000)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -x -q -k "merge or decode or split or short or graph" 2>&1 | tail -2
timeout 900 python tools/gpu_ab.py --case decode,decode_b8,decode_d1024,decode_d128,decode_long,decode_q16 --rounds 9 --reps 20 prev:0x10000 main:0x10000 main > gpurun_out/samexcd_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/samexcd_ab.txt
