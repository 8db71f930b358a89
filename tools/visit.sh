#!/bin/bash
mkdir -p gpurun_out/v23
timeout 200 tools/probes/bin/share_probe > gpurun_out/v23/share_probe.txt 2>&1
cat gpurun_out/v23/share_probe.txt
timeout 300 python bench.py > gpurun_out/v23/bench_default.json 2> gpurun_out/v23/bench_default.err
tail -c 3000 gpurun_out/v23/bench_default.json
