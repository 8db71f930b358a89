#!/bin/bash
# round 6, fourth evidence visit, second part (same library 20179c2c: only tests / tools / docs changed since): the GPU suite and smoke on the final tree, build() as the driver runs it
export TMPDIR=/tmp
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/final/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/final/smoke.txt
timeout 300 python bench.py 2>/dev/null | tail -1 | cut -c1-400 | tee gpurun_out/final/bench_default.json
