#!/bin/bash
mkdir -p gpurun_out/v27
rm -f ffpa_attn_amd/variants/*.so
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/v27/pytest.txt 2>&1
tail -4 gpurun_out/v27/pytest.txt
ARMS=0,1,3 ONLY=h5_n4096,h6_n4096,h5_n4096_short,h5_d320,h20_n1024,h5_n4096_16k,h3_d1024 timeout 300 python tools/gpu_prefill_splits.py > gpurun_out/v27/partial_rule.txt 2>&1
grep -h "^SPLITS" gpurun_out/v27/*.txt
