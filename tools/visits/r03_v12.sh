#!/bin/bash
# round 3, GPU visit 12: D = 1024 schedule knobs on the 16x16x32 kernel (fragment prefetch depths, DMA position relative to the MFMA pair, denser V pieces)
export AB_ARGS="--rounds 5 --reps 3 --case cfg3 main pf1_4 pf1_8 pf2_2 pf2_6 pos0 pos2 s1d2"
bash tools/gpu_round.sh "ab"
