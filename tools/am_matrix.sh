#!/usr/bin/env bash
# tools/am_matrix.sh -- c3 (4096 AM channels) over taps x biquad stages: chain_amtr_kernel vs chain_mfw_kernel (MSDR_NO_AMTR=1), same box
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r02
for taps in ${TAPS:-64 128 256}; do for st in ${STAGES:-0 1 2 4}; do
  for e in MSDR_X=0 MSDR_NO_AMTR=1; do
    env $e python bench.py --workload c3 --taps $taps --stages $st --no-cpu --no-parity --steps 10 > gpurun_out/r02/amx.json 2>/dev/null
    python -c "
import json
d=json.load(open('gpurun_out/r02/amx.json'))
print('taps', $taps, 'stages', $st, '$e', 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['config'].get('kernel'))"
  done
done; done
