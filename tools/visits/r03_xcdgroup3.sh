#!/bin/bash
# Visit: does the best number of XCDs per head follow the K + V footprint?  D = 1024 at 4k / 6k / 8k / 12k tokens (8 heads in flight: 128 / 192 / 256 / 384 MiB), D = 512 at 12k (192 MiB)
mkdir -p gpurun_out
timeout 900 python tools/gpu_ab.py --case n4k_d1024,n6k_d1024,cfg3,n12k_d1024,n12k_d512 --rounds 5 --reps 4 main:0x100 main:0x200 main:0x300 main > gpurun_out/xcdgroup_ab3.txt 2>&1
grep "^AB\|rror" gpurun_out/xcdgroup_ab3.txt
