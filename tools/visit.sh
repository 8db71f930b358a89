#!/bin/bash
mkdir -p gpurun_out/v29
timeout 900 python -m pytest tests/test_fwd_gpu.py tests/test_sharding_gpu.py tests/test_m16_gpu.py -m gpu -q -k "split or underfilled or ragged or shard or rccl or chunk" > gpurun_out/v29/pytest.txt 2>&1
tail -3 gpurun_out/v29/pytest.txt
ARMS=0,2,5 ONLY=h3_n4096,h3_n4096_16k,h3_n2048_d1024,h12_n1024 timeout 300 python tools/gpu_prefill_splits.py > gpurun_out/v29/underfilled_rule.txt 2>&1
grep -h "^SPLITS" gpurun_out/v29/*.txt
