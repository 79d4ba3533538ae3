#!/usr/bin/env bash
# tools/r05_fuse_ab.sh -- Q15 block cadence: the two biquad nodes inside chain_q15mb_kernel (default where it applies) against the node kernel behind it (MSDR_Q15_NO_FUSE=1)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for chs in 1024 4096 8192 16384; do
for nf in 1 0; do
  if [ $nf = 1 ]; then export MSDR_Q15_NO_FUSE=1; else unset MSDR_Q15_NO_FUSE; fi
  timeout -k 10 120 tools/memguard.sh -m 24 -t 100 python3 bench.py --workload c3 --arith q15 --channels $chs --samples 128 --steps 400 --warmup 50 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ch $chs no_fuse $nf tick', r.get('tick_us'),'us graph', r.get('graph_tick_us'), d['config']['kernel'][:60], 'mismatches', (d.get('parity') or {}).get('mismatching_samples'))"
done
done
