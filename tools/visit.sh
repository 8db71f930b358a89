#!/bin/bash
mkdir -p gpurun_out/v33
timeout 600 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k "executed_reference" > gpurun_out/v33/dropfix.txt 2>&1
tail -15 gpurun_out/v33/dropfix.txt
