#!/bin/bash
# Developer tool: compile ONE head dim's kernels with extra -D switches into a scratch directory, keep the ISA, print register / spill
# statistics and run the MFMA hazard check on it.      tools/dev_compile.sh 1024 -DFFPA_M16_PIPE=1 [...]
set -e
D=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${FFPA_DEV_OUT:-/tmp/ffpa_dev}
mkdir -p $OUT/temps_d$D
cd $OUT/temps_d$D
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -I$ROOT/include -I$ROOT/ffpa_attn_amd/csrc -save-temps "$@" -DFFPA_INST_D=$D \
  -c $ROOT/ffpa_attn_amd/csrc/ffpa_fwd_inst.hip -o $OUT/temps_d$D/d$D.o 2>&1 | grep -E "error|static assertion" -A6 | head -40 || true
cd $ROOT
FFPA_ISA_ROOT=$OUT python tools/isa_stats.py $D | grep m16 | head -${FFPA_DEV_LINES:-4}
FFPA_ISA_ROOT=$OUT python - <<EOF
import importlib.util, sys
spec = importlib.util.spec_from_file_location('chk', '$ROOT/tools/check_mfma_hazards.py'); chk = importlib.util.module_from_spec(spec); spec.loader.exec_module(chk)
chk.ROOT = '$OUT'
sys.argv = ['x', '$D']
sys.exit(chk.main())
EOF
