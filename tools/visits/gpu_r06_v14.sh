#!/bin/bash
# round 6, visit 14: the non-temporal K / V fetch in the packed-sequence kernel (decode batches with one reader per K / V byte): packed tests, interleaved A/B of the two builds
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_varlen_gpu.py -x -q 2>&1 | tail -3
timeout 900 python tools/gpu_varlen_decode.py 2>&1 | tee gpurun_out/r06/v14_varlen_decode_nt.txt
