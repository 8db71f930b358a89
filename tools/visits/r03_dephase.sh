#!/bin/bash
# Visit: do wave-phase offsets behind the barriers (FFPA_M16_DEPHASE) and K(j+1) pieces under the PV MFMAs move D = 1024 / D = 512?
mkdir -p gpurun_out
timeout 600 python tools/gpu_ab.py --case cfg3,d768 --rounds 7 --reps 5 main dp1 dp2 dp1k8 dp1k0 k8 dp1q > gpurun_out/dephase_ab.txt 2>&1
timeout 600 python tools/gpu_ab.py --case cfg2,d320 --rounds 7 --reps 5 main dp1s dp1sq >> gpurun_out/dephase_ab.txt 2>&1
grep "^AB\|rror" gpurun_out/dephase_ab.txt
