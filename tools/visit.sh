#!/bin/bash
mkdir -p gpurun_out/v18
timeout 120 python tools/gpu_diff.py main p1a --splits 1 --shape 1,2,128,64,512 --shape 1,2,128,128,512 --shape 1,2,256,1000,512 --shape 1,2,300,2049,512 > gpurun_out/v18/diff.txt 2>&1
timeout 120 python tools/gpu_diff.py main p1a --causal --splits 1 --shape 1,2,512,512,512 --shape 1,2,200,1000,512 >> gpurun_out/v18/diff.txt 2>&1
timeout 300 python tools/gpu_ab.py --case cfg2,causal,cross,gqa,non_aligned,n2048 --rounds 7 --reps 3 main p1a p1b p1c p1d > gpurun_out/v18/ab_p1.txt 2>&1
timeout 300 python tools/gpu_phase_times.py --dims 512 timing512 timing_p1a > gpurun_out/v18/phase.txt 2>&1
grep -h "^DIFF\|LSE el\|^AB\|^PHASE" gpurun_out/v18/*.txt
