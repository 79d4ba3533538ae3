#!/usr/bin/env bash
# tools/r05_profiles.sh [parts...] -- round 5's judged evidence on ONE box (run through gpurun, each child under tools/memguard.sh):
# rocprofv3 kernel stats + HBM counter passes (tools/profile.sh: separate --pmc passes, never with a tracing domain) for the headline, the FIR
# stage (256 and 512 taps), c2 / c4 / c5, the records that had no counter pass so far (c3_i16, q15_c3) and the block-cadence records; SQ counters
# of the block kernel; the default bench line.  Output: gpurun_out/r05p/ (copy what is judged into profiles/r05/).
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05p; mkdir -p $O
PARTS="${*:-prof_c3 prof_fir prof_fir512 prof_c2 prof_c4 prof_c5 prof_c3_i16 prof_q15_c3 prof_c3_b128 prof_c4_b128 prof_q15_c3_b128 pmc_c3_b128 default}"
echo "== profiles, parts: $PARTS; library $(python3 -c "import sys; sys.path.insert(0,'minimal-sdr_amd/python'); import msdr,ctypes; l=msdr.load_library(); l.msdr_build_rev.restype=ctypes.c_char_p; print(l.msdr_build_rev().decode())" 2>/dev/null)" > $O/README_run.txt
G="tools/memguard.sh -m 32 -t 500"
for p in $PARTS; do
  case $p in
    prof_fir) $G bash tools/profile.sh fir --workload fir --no-parity > $O/fir_f32_rocprof_summary.txt 2>&1 ;;
    prof_fir512) $G bash tools/profile.sh fir512 --workload fir --taps 512 --no-parity > $O/fir_f32_512_rocprof_summary.txt 2>&1 ;;
    prof_c3_i16) $G bash tools/profile.sh c3_i16 --workload c3 --out-i16 --no-parity > $O/c3_i16_rocprof_summary.txt 2>&1 ;;
    prof_q15_c3) $G bash tools/profile.sh q15_c3 --workload c3 --arith q15 --no-parity > $O/q15_c3_rocprof_summary.txt 2>&1 ;;
    prof_c3_b128) $G bash tools/profile.sh c3_b128 --workload c3 --samples 128 --steps 200 --warmup 20 --no-parity > $O/c3_b128_rocprof_summary.txt 2>&1 ;;
    prof_c4_b128) $G bash tools/profile.sh c4_b128 --workload c4 --samples 128 --steps 200 --warmup 20 --no-parity > $O/c4_b128_rocprof_summary.txt 2>&1 ;;
    prof_q15_c3_b128) $G bash tools/profile.sh q15_c3_b128 --workload c3 --samples 128 --arith q15 --steps 200 --warmup 20 --no-parity > $O/q15_c3_b128_rocprof_summary.txt 2>&1 ;;
    prof_*) w=${p#prof_}; $G bash tools/profile.sh $w --workload $w --no-parity > $O/${w}_rocprof_summary.txt 2>&1 ;;
    pmc_c3_b128) $G bash tools/pmc.sh c3_b128 --workload c3 --samples 128 --channels 65536 --no-parity > $O/c3_b128_64k_sq_counters.txt 2>&1 ;;
    pmc_*) w=${p#pmc_}; $G bash tools/pmc.sh $w --workload $w --no-parity > $O/${w}_sq_counters.txt 2>&1 ;;
    default) tools/memguard.sh -m 40 -t 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cp bench_full.json $O/bench_full.json 2>/dev/null ;;
  esac
  echo "$p done"
done
find gpurun_out -name "*.csv" -size +1M -delete 2>/dev/null
find gpurun_out -name "*.db" -delete 2>/dev/null
echo "all done"
