#!/bin/bash
mkdir -p gpurun_out/v22
rm -f ffpa_attn_amd/variants/*.so
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/v22/pytest.txt 2>&1
tail -5 gpurun_out/v22/pytest.txt
ONLY=h9_n4096,h10_n4096,h11_n4096,h12_n4096,h12_n4096_short,h10_d320,h20_n4096_d1024,h6_d1024,h40_n1024,h17,cross timeout 400 python tools/gpu_prefill_splits.py > gpurun_out/v22/ragged_rule.txt 2>&1
grep -h "^SPLITS" gpurun_out/v22/*.txt
