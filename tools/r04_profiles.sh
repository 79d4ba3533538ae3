#!/usr/bin/env bash
# tools/r04_profiles.sh [parts...] -- the round's judged evidence on ONE box (run through gpurun): rocprofv3 kernel stats + HBM counter
# passes (tools/profile.sh: separate --pmc passes, never with a tracing domain) for the FIR stage and c2..c5, SQ counters of c3, c5 and the FIR
# stage, every bench line (bench.py samples shader clock and package power per record itself from round 4 on).  Parts: prof_<w> pmc_<w> final default (default: all).
# Output: gpurun_out/r04p/ (copy what is judged into profiles/r04/).
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04p; mkdir -p $O
PARTS="${*:-prof_fir prof_c3 prof_c2 prof_c4 prof_c5 prof_spec pmc_c3 pmc_c5 pmc_fir final default}"
echo "== profiles for source revision $(cat .git_rev 2>/dev/null || echo unknown), parts: $PARTS" > $O/README_run.txt
for p in $PARTS; do
  case $p in
    prof_fir) bash tools/profile.sh fir --workload fir --no-parity > $O/fir_f32_rocprof_summary.txt 2>&1 ;;
    prof_spec) bash tools/profile.sh spec --workload spec > $O/spec_rocprof_summary.txt 2>&1 ;;
    prof_*) w=${p#prof_}; bash tools/profile.sh $w --workload $w --no-parity > $O/${w}_rocprof_summary.txt 2>&1 ;;
    pmc_fir) bash tools/pmc.sh fir --workload fir --no-parity > $O/fir_f32_sq_counters.txt 2>&1 ;;
    pmc_*) w=${p#pmc_}; bash tools/pmc.sh $w --workload $w --no-parity > $O/${w}_sq_counters.txt 2>&1 ;;
    final) bash tools/final_bench.sh > $O/final_bench_table.txt 2>&1; mkdir -p $O/final && cp gpurun_out/final/*.json $O/final/ 2>/dev/null ;;
    default) python bench.py > $O/bench_default.json 2> $O/bench_default.err ;;
  esac
  echo "$p done"
done
echo "all done"
