#!/usr/bin/env bash
# tools/r02_profiles.sh -- the round's judged evidence on ONE box (run through gpurun): rocprofv3 kernel stats + HBM counter passes for
# the FIR stage, c3 and c2; SQ counters of c3 and the FIR stage; package power / shader clock while each kernel loops; every bench line.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r02
git_rev=$(cat .git_rev 2>/dev/null || echo unknown)
echo "== profiles for source revision $git_rev" > gpurun_out/r02/README_run.txt
bash tools/profile.sh fir --workload fir --no-parity > gpurun_out/r02/fir_f32_rocprof_summary.txt 2>&1
echo "fir profile done"
bash tools/profile.sh c3 --workload c3 --no-parity > gpurun_out/r02/c3_rocprof_summary.txt 2>&1
echo "c3 profile done"
bash tools/profile.sh c2 --workload c2 --no-parity > gpurun_out/r02/c2_rocprof_summary.txt 2>&1
echo "c2 profile done"
bash tools/profile.sh spec --workload spec > gpurun_out/r02/spec_rocprof_summary.txt 2>&1
echo "spec profile done"
bash tools/pmc.sh c3 --workload c3 --no-parity > gpurun_out/r02/c3_sq_counters.txt 2>&1
echo "c3 counters done"
bash tools/pmc.sh fir --workload fir --no-parity > gpurun_out/r02/fir_f32_sq_counters.txt 2>&1
echo "fir counters done"
{
  for w in fir c3 c2 c4 c5; do
    echo "---- $w"
    STEPS=$(case $w in c4) echo 30000;; c5) echo 12000;; *) echo 4000;; esac) bash tools/smi_probe.sh $w -- --workload $w 2>&1 | grep -E "sclk|ms " | sed 's/GPU\[0\]\s*: //g; s/=\+ Power Consumption =\+//'
  done
} > gpurun_out/r02/power_clock_samples.txt 2>&1
echo "power samples done"
bash tools/final_bench.sh > gpurun_out/r02/final_bench_table.txt 2>&1
mkdir -p gpurun_out/r02/final && cp gpurun_out/final/*.json gpurun_out/r02/final/ 2>/dev/null
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err
echo "all done"
