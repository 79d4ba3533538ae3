#!/usr/bin/env bash
# tools/profile.sh <tag> <bench args...> -- rocprofv3 kernel trace + HBM counters for one bench.py command.
# Run on the GPU box (through gpurun).  Writes gpurun_out/prof_<tag>/{trace,fetch,write}/ and a summary
# gpurun_out/prof_<tag>/summary.txt.  The counter passes are SEPARATE runs (FETCH_SIZE and WRITE_SIZE do
# not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with tracing domains.
set -u
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
export MSDR_BENCH_NO_POWER=1      # no rocm-smi child process under the profiler
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="$* --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS --steps 2 --warmup 1 > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS --steps 2 --warmup 1 > "$OUT/write.log" 2>&1
python3 tools/prof_summary.py "$OUT" "$ARGS" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small artefacts
find "$OUT" -name "*.csv" -size +2M -delete
