#!/bin/bash
# Copy the evidence of a round from gpurun_out/ (scratch, merged back by gpurun) into profiles/ (tracked): bench lines of every workload,
# the one-shot sweep, and per profiled workload the rocprofv3 kernel-trace statistics + the PMC summary (tools/pmc_summary.py: with the
# binary's sha / git head / clock and the kernel-trace median).   usage: tools/collect_profiles.sh r03
R=${1:?round tag, e.g. r03}
for f in gpurun_out/final/bench_*.json; do w=$(basename $f .json); w=${w#bench_}; [ -s $f ] && tail -1 $f > profiles/${R}_bench_$w.json; done
[ -s gpurun_out/final/sweep.json ] && tail -1 gpurun_out/final/sweep.json > profiles/${R}_sweep.json && grep -v "amdgpu.ids" gpurun_out/final/sweep.txt > profiles/${R}_sweep.txt
for d in gpurun_out/wprof_*/; do w=$(basename $d); w=${w#wprof_}
  [ -s $d/summary.json ] && cp $d/summary.json profiles/${R}_bench_${w}_pmc.json
  s=$(find $d -name "trace_kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s profiles/${R}_bench_${w}_kernel_stats.csv
done
[ -s gpurun_out/final/pytest.log ] && tail -3 gpurun_out/final/pytest.log > profiles/${R}_gpu_pytest_tail.txt
ls profiles | grep "^${R}_" | wc -l
