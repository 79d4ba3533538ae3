#!/usr/bin/env bash
# tools/r05_fr_bench.sh -- long FIRs behind a general (period-128) oscillator table on c4's shape: the full-rate matrix-core layout against the fallback
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for taps in 240 256 320 384 512; do
  for lim in 248 ${FRMAX:-1000}; do
    MSDR_FR_MAX_TAPS=$lim tools/memguard.sh -m 24 -t 200 python3 bench.py --workload c4 --osc-period 128 --taps $taps --steps 20 --warmup 5 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('taps $taps limit $lim', d['config']['kernel'], 'block', d['config'].get('grid'), 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'parity', (d.get('parity') or {}).get('rel_rms_worst'))"
  done
done
