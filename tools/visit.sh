#!/bin/bash
mkdir -p gpurun_out/v32
timeout 600 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_launch_plans_against_oracle > gpurun_out/v32/plans.txt 2>&1
tail -15 gpurun_out/v32/plans.txt
