#!/bin/bash
# Visit: what the softmax arithmetic costs in time AND in clock — the shipped kernel vs ablation builds (wrong results by design) under rocprofv3 PMC
# (GRBM_GUI_ACTIVE / duration = effective shader clock of each launch); and the same launches un-profiled.
mkdir -p gpurun_out
timeout 600 python tools/gpu_ab.py --case cfg2,cfg3 --rounds 5 --reps 5 main abl1 abl3 > gpurun_out/clk_ab.txt 2>&1; grep "^AB" gpurun_out/clk_ab.txt
for c in cfg2 cfg3; do
  CLK_ARGS="--case $c main abl1 abl3" bash tools/gpu_round.sh clocks > gpurun_out/clk_$c.log 2>&1
  f=$(find gpurun_out/clk -name "*counter_collection.csv" | head -1)
  echo "== $c"; python tools/parse_clk.py $f main abl1 abl3 | tee gpurun_out/clk_$c.txt
done
