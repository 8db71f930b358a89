#!/bin/bash
mkdir -p gpurun_out/v42
for shp in "1,32,1,320 --hkv 32 --nkv 8192" "1,32,1,512 --hkv 32 --nkv 5120" "1,32,1,512 --hkv 32 --nkv 6144" "1,32,1,512 --hkv 32 --nkv 4608" "8,32,1,128 --hkv 8 --nkv 16384" "1,32,1,1024 --hkv 32 --nkv 2560"; do
  timeout 200 python tools/gpu_ab.py --shape $shp --rounds 7 --reps 10 main:0x800 main:0x80 main:0x800 main:0x80 >> gpurun_out/v42/ab_threshold.txt 2>&1
done
grep "^AB" gpurun_out/v42/ab_threshold.txt
