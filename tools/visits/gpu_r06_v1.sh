#!/bin/bash
# round 6, visit 1: smoke of the rebuilt library (sha 267de97f = round 5's), the error budget against exact math, phase times of the shipped loops
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python tools/gpu_error_budget.py --out gpurun_out/r06/error_budget.txt > gpurun_out/r06/v1_err.log 2>&1; echo "err exit $?"; tail -45 gpurun_out/r06/v1_err.log
timeout 300 python tools/gpu_phase_times.py --dims 512,1024 timing > gpurun_out/r06/v1_phase.txt 2>&1; echo "phase exit $?"; grep PHASE gpurun_out/r06/v1_phase.txt
