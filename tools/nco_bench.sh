#!/usr/bin/env bash
# tools/nco_bench.sh -- c2 / c4 with oscillator tables of different periods (same box, one after the other)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r02
for wl in c2 c4; do for P in 4 64 128 3; do
  python bench.py --workload $wl --osc-period $P --no-cpu > gpurun_out/r02/nco_${wl}_P$P.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/r02/nco_${wl}_P$P.json'))
print('$wl period $P:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'], d['config']['kernel'], 'parity', d['parity']['rel_rms_worst'])"
done; done
