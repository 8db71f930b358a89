#!/bin/bash
# round 5, visit 6: host-path changes (inference fast path, scratch reuse, kv_bounds through the Backend object, device-priced plan) — full suite + decode bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05/v6_pytest.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/r05/v6_pytest.log
timeout 300 python -m pytest tests/test_host_path_gpu.py -q -s -k pricing 2>&1 | grep "device pricing"
timeout 300 python tools/gpu_host_overhead.py 2>&1 | grep HOST
for i in 1 2; do timeout 300 python bench.py --workload decode --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DECODE', d['value'], d['unit'], d['ms_per_step'], 'ms/step | steady', (d.get('steady_state') or {}).get('ms_per_step'), '| kernel ms avg', d['roofline'].get('kernel_ms_avg'))"; done
timeout 300 python bench.py --workload cfg4_mask --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CFG4', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], '| steady', (d.get('steady_state') or {}).get('tflops'))"
