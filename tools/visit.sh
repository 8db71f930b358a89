#!/bin/bash
mkdir -p gpurun_out/v39
rm -f ffpa_attn_amd/variants/*.so
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/v39/pytest.txt 2>&1
tail -3 gpurun_out/v39/pytest.txt
timeout 400 python tools/gpu_ab.py --case decode,decode_b8,decode_d1024,decode_d128,decode_long,decode_q16 --rounds 5 --reps 10 main main:0x800 main:0x80 > gpurun_out/v39/ab_stream.txt 2>&1
for shp in "1,32,16,512 --hkv 8 --nkv 8192" "1,32,32,512 --hkv 4 --nkv 8192" "4,32,1,512 --hkv 32 --nkv 1024" "2,32,1,512 --hkv 32 --nkv 4096" "1,32,1,512 --hkv 8 --nkv 131072" "1,32,1,320 --hkv 32 --nkv 16384"; do
  timeout 200 python tools/gpu_ab.py --shape $shp --rounds 5 --reps 10 main main:0x800 main:0x80 >> gpurun_out/v39/ab_stream.txt 2>&1
done
grep "^AB" gpurun_out/v39/ab_stream.txt
