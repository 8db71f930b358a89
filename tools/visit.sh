#!/bin/bash
mkdir -p gpurun_out/v7
export TMPDIR=/tmp
timeout 300 python tools/gpu_ab.py --case cfg3 --rounds 7 --reps 3 main pp1 pp3 ppA ppB ppC ppD ppE ppF > gpurun_out/v7/ab_cfg3.txt 2>&1
timeout 120 python tools/gpu_diff.py main pp3 --splits 0 --shape 1,2,256,2048,1024 --shape 1,2,200,1000,768 > gpurun_out/v7/diff.txt 2>&1
(FFPA_HIP_LIBRARY=$PWD/ffpa_attn_amd/variants/libffpa_attn_hip_pp3.so timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/v7/pytest_pp3.txt
grep -h "^DIFF\|LSE el\|^AB\|passed\|failed\|rror" gpurun_out/v7/*.txt
