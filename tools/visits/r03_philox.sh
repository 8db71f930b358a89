#!/bin/bash
# Visit: dropout parity on the integer keep threshold + A/B against the previous build.
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "dropout or Dropout or philox" > gpurun_out/philox_pytest.txt 2>&1
tail -5 gpurun_out/philox_pytest.txt
timeout 900 python tools/gpu_ab.py --case dropout,dropout_d320,dropout_d1024,dropout_d128,dropout_d256,d64 --rounds 7 --reps 5 prev philox1 main > gpurun_out/philox_ab2.txt 2>&1
grep "^AB\|Error\|error" gpurun_out/philox_ab2.txt
