#!/bin/bash
# round 6, visit 12: the wide-row tile of D = 320 with the P^T fragments of key step 1 made in the MFMA gaps of PV step 0 (variants wovl1 / wovl2 / wovl3: one, two or all three row
# halves overlapped) — bit-identity + speed on config 4 in its three forms and on dense / causal D = 320 launches with the wide tile forced
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python tools/gpu_ab.py --case cfg4_mask,cfg4_offset0,cfg4_nomask,d320,d320_causal,d320_b3 --rounds 7 --reps 10 main:0x1000 wovl1:0x1000 wovl2:0x1000 wovl3:0x1000 > gpurun_out/r06/v12_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v12_ab.txt
