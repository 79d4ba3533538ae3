#!/usr/bin/env bash
# tools/r05_fr_compact.sh -- full-rate layout, B operand as k-step fragments (MSDR_FR_COMPACT=0) or as shifted copies of the taps (=1): c4's shape
# behind a period-128 oscillator table, SSB (2 sections) and the envelope (c3's shape: AM, hI == hQ)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
tools/memguard.sh -m 24 -t 600 python3 -m pytest tests/test_gpu_chain.py tests/test_gpu_out_i16.py -q -x -k "general or nco or full_rate or long" 2>&1 | tail -4
for wl in ${WLS:-c4 c3}; do
for taps in ${TAPS:-96 160 240 256 384 512}; do
  for cp in 0 1; do
    MSDR_FR_COMPACT=$cp tools/memguard.sh -m 24 -t 200 python3 bench.py --workload $wl --osc-period 128 --taps $taps --steps 20 --warmup 5 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$wl taps $taps compact $cp', d['config']['kernel'], 'block', d['config'].get('block'), 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'parity', (d.get('parity') or {}).get('rel_rms_worst'))"
  done
done
done
