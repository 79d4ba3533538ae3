#!/usr/bin/env bash
# tools/r05_small.sh -- the reference's own size at the reference's cadence: ONE receiver (and a few), one 128-sample block per call
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for chs in 1 16 64 256; do
  for ar in q15 f32; do
    timeout -k 10 120 python3 bench.py --workload c3 --arith $ar --channels $chs --samples 128 --steps 1000 --warmup 100 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; par=d.get('parity') or {}
print('channels $chs $ar tick', r.get('tick_us'), 'us; graph', r.get('graph_tick_us'), d['config']['kernel'][:55], 'grid', d['config']['grid'], 'parity', par.get('rel_rms_worst', par.get('mismatching_samples')))"
  done
done
