#!/bin/bash
# The pool's boxes differ by +- 4 % on the same binary (power / clock).  This gate measures config 2 for a few seconds and runs the round's evidence
# visit (r03_final_a.sh + r03_final_b.sh) only on a box of at least the pool's middle speed, so that the committed lines are comparable with round 2's.
mkdir -p gpurun_out
tf=$(timeout 120 python tools/gpu_ab.py --case cfg2 --rounds 5 --reps 5 main 2>/dev/null | grep "^AB" | sed 's/.*median *[0-9.]* ms *\([0-9.]*\) TF.*/\1/')
echo "gate: config 2 at $tf TFLOPS on this box"
python - "$tf" <<'PY' || { echo "gate: slower than ${GATE_TF:-1335} TFLOPS — leaving this box"; exit 7; }
import os, sys
sys.exit(0 if float(sys.argv[1] or 0) >= float(os.environ.get("GATE_TF", "1335")) else 1)
PY
bash tools/visits/r03_final_a.sh; bash tools/visits/r03_final_b.sh
