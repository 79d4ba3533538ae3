#!/usr/bin/env bash
# tools/trsweep2.sh -- same-box sweep of "ENV=VAL" settings on the FIR-stage bench (two rounds, interleaved)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r02
for rep in 1 2; do for f in "$@"; do
  env $f python bench.py --workload ${WL:-fir} --no-cpu --no-parity > gpurun_out/r02/sweep_x.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/r02/sweep_x.json'))
print('$f', 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
done; done
