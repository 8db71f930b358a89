#!/bin/bash
# round 6, visit 2: the error budget again over five seeds (is the kernel's larger MAXIMUM against exact math systematic, or one element's rounding flip?)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1200 python tools/gpu_error_budget.py --seeds 0,1,2,3,4 --out gpurun_out/r06/error_budget.txt > gpurun_out/r06/v2_err.log 2>&1; echo "err exit $?"; grep -v Warning gpurun_out/r06/v2_err.log | grep -E "^(#|kernel|SDPA|emul|what)" | cut -c1-150
