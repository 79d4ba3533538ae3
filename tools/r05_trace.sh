#!/usr/bin/env bash
# tools/r05_trace.sh <tag> <bench args...> -- rocprofv3 kernel trace of one bench command; prints the kernel stats head
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp MSDR_BENCH_NO_POWER=1
TAG=$1; shift
O=gpurun_out/trace_$TAG; rm -rf $O; mkdir -p $O
tools/memguard.sh -m 24 -t 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py "$@" --no-cpu --no-parity > $O/log.txt 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-160 "$f" | head -5
find $O -name "*.csv" -size +1M -delete
tail -1 $O/log.txt | cut -c1-300
