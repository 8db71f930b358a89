#!/bin/bash
mkdir -p gpurun_out/v12
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed|^E  " | head -60) > gpurun_out/v12/pytest.txt
cat gpurun_out/v12/pytest.txt | cut -c1-220
