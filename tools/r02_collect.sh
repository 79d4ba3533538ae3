#!/usr/bin/env bash
# tools/r02_collect.sh -- copy what tools/r02_profiles.sh left under gpurun_out/ into profiles/r02/ (run here, after the gpurun call)
set -u
cd "$(dirname "$0")/.."
D=profiles/r02
for f in c3_rocprof_summary.txt c2_rocprof_summary.txt fir_f32_rocprof_summary.txt spec_rocprof_summary.txt c3_sq_counters.txt fir_f32_sq_counters.txt \
         power_clock_samples.txt final_bench_table.txt bench_default.json; do
  [ -s gpurun_out/r02/$f ] && cp gpurun_out/r02/$f $D/$f
done
mkdir -p $D/final && cp gpurun_out/r02/final/*.json $D/final/ 2>/dev/null
for tag in c3 c2 fir spec; do
  # the kernel-stats table of the process that ran the kernels (the one naming a msdr kernel)
  f=$(grep -l "msdr::" gpurun_out/prof_$tag/trace/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $D/${tag}_kernel_stats.csv
done
[ -s gpurun_out/r02/gpu_tests_final.txt ] && tail -3 gpurun_out/r02/gpu_tests_final.txt > $D/gpu_tests_final.txt
echo "taken at source revision $(cat .git_rev 2>/dev/null || git rev-parse --short HEAD) by tools/r02_profiles.sh (one gpurun call, one box)" > $D/REVISION.txt
ls -la $D | head -60
