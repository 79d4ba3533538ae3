#!/usr/bin/env bash
# tools/r03_profiles.sh [parts...] -- the round's judged evidence on ONE box (run through gpurun): rocprofv3 kernel stats + HBM counter
# passes (tools/profile.sh: separate --pmc passes, never with a tracing domain) for the FIR stage and c2..c5, SQ counters of c3, c5 and the FIR
# stage, package power / shader clock while each kernel loops, every bench line.  Parts: prof_<w> pmc_<w> power final default (default: all).
# Output: gpurun_out/r03p/ (copy what is judged into profiles/r03/).
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r03p; mkdir -p $O
PARTS="${*:-prof_fir prof_c3 prof_c2 prof_c4 prof_c5 prof_spec pmc_c3 pmc_c5 pmc_fir power final default}"
echo "== profiles for source revision $(cat .git_rev 2>/dev/null || echo unknown), parts: $PARTS" > $O/README_run.txt
for p in $PARTS; do
  case $p in
    prof_fir) bash tools/profile.sh fir --workload fir --no-parity > $O/fir_f32_rocprof_summary.txt 2>&1 ;;
    prof_spec) bash tools/profile.sh spec --workload spec > $O/spec_rocprof_summary.txt 2>&1 ;;
    prof_*) w=${p#prof_}; bash tools/profile.sh $w --workload $w --no-parity > $O/${w}_rocprof_summary.txt 2>&1 ;;
    pmc_fir) bash tools/pmc.sh fir --workload fir --no-parity > $O/fir_f32_sq_counters.txt 2>&1 ;;
    pmc_*) w=${p#pmc_}; bash tools/pmc.sh $w --workload $w --no-parity > $O/${w}_sq_counters.txt 2>&1 ;;
    power) { for w in fir c3 c2 c4 c5; do echo "---- $w"
               STEPS=$(case $w in c4) echo 30000;; c5) echo 12000;; *) echo 4000;; esac) bash tools/smi_probe.sh $w -- --workload $w 2>&1 | grep -E "sclk|ms " | sed 's/GPU\[0\]\s*: //g; s/=\+ Power Consumption =\+//'
             done; } > $O/power_clock_samples.txt 2>&1 ;;
    final) bash tools/final_bench.sh > $O/final_bench_table.txt 2>&1; mkdir -p $O/final && cp gpurun_out/final/*.json $O/final/ 2>/dev/null ;;
    default) python bench.py > $O/bench_default.json 2> $O/bench_default.err ;;
  esac
  echo "$p done"
done
echo "all done"
