#!/bin/bash
mkdir -p gpurun_out/v14
export TMPDIR=/tmp
timeout 300 python tools/gpu_prefill_splits.py > gpurun_out/v14/prefill_splits.txt 2>&1
timeout 300 python tools/gpu_ab.py --case cfg3 --rounds 7 --reps 3 main main:0x100 main:0x200 main:0x300 main:0x400 main:0x20 > gpurun_out/v14/ab_cfg3_flags.txt 2>&1
timeout 300 python tools/gpu_ab.py --case d768,d640 --rounds 5 --reps 3 main main:0x100 main:0x200 main:0x20 > gpurun_out/v14/ab_d768_flags.txt 2>&1
grep -h "^SPLITS\|^AB" gpurun_out/v14/*.txt
