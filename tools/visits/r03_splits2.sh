#!/bin/bash
mkdir -p gpurun_out
AUTO_ONLY=1 timeout 600 python tools/gpu_decode_splits.py 2>&1 | grep "SPLITS\|rror" | tee gpurun_out/decode_splits4.txt
timeout 900 python -m pytest tests/test_fwd_gpu.py tests/test_capi.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --workload decode --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('decode', d['ms_per_step'], d['roofline'])"
