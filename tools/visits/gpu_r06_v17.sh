#!/bin/bash
# round 6, visit 17: heads x tokens packed into the rows of one tile (short sequences under GQA): the packed tests, packed decode / speculative-decoding batches vs one workgroup per query head
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_varlen_gpu.py -q 2>&1 | tail -15
timeout 900 python tools/gpu_varlen_decode.py 2>&1 | tee gpurun_out/r06/v17_varlen_decode.txt
