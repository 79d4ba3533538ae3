#!/usr/bin/env bash
# tools/r05_fill.sh -- channels placed per block tile (MSDR_MB_FILL) at small batches: c3's / c4's shape x 128
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for chs in 1024 2048 4096 8192; do
  for fill in 8 4 2 auto; do
    for wl in c3 c4; do
      if [ $fill = auto ]; then unset MSDR_MB_FILL; else export MSDR_MB_FILL=$fill; fi
      tools/memguard.sh -m 24 -t 120 python3 bench.py --workload $wl --samples 128 --channels $chs --steps 300 --warmup 50 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ch $chs fill $fill $wl', d['value'],'Msps', r['tick_us'],'us/tick', 'grid', d['config']['grid'], 'parity', (d.get('parity') or {}).get('rel_rms_worst'))"
    done
  done
done
