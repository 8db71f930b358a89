#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05/v7_pytest.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/r05/v7_pytest.log | tail -20
