#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -x -q -k "dropout_keep_threshold" > gpurun_out/thr_pytest.txt 2>&1; tail -2 gpurun_out/thr_pytest.txt
timeout 900 python tools/gpu_ab.py --case cfg4_mask,cfg4_nomask,cfg4_offset0,cross,n1024,n2048,causal4k,causal,cfg2,n2048_d1024 --rounds 7 --reps 10 main kfirst > gpurun_out/kfirst_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/kfirst_ab.txt
