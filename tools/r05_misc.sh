#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for taps in 256 320 512; do
  tools/memguard.sh -m 24 -t 200 python3 bench.py --workload fir --taps $taps --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('fir taps $taps', d['config']['kernel'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'parity', (d.get('parity') or {}).get('rel_rms_worst'))"
done
bash tools/r05_trace.sh q15b --workload c3 --samples 128 --arith q15 --steps 200 --warmup 20
