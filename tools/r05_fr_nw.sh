#!/usr/bin/env bash
# tools/r05_fr_nw.sh -- full-rate compact layout: waves per workgroup at long filters (c4's shape, period-128 table)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for taps in ${TAPS:-128 256 512}; do
  for nw in auto 3 4 5 6 7 8 12 16; do
    if [ $nw = auto ]; then unset MSDR_MFW_NW; else export MSDR_MFW_NW=$nw; fi
    tools/memguard.sh -m 24 -t 200 python3 bench.py --workload c4 --osc-period 128 --taps $taps --steps 20 --warmup 5 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('taps $taps nw $nw', d['config']['kernel'], 'block', d['config'].get('block'), 'grid', d['config'].get('grid'), 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'parity', (d.get('parity') or {}).get('rel_rms_worst'))"
  done
done
