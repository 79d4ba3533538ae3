#!/usr/bin/env bash
# tools/ab_q15.sh -- same-box A/B of minimal-sdr_amd/lib_ab/libmsdr_{old,new}.so on bench configurations given as "workload extra-args" strings
# (default: the Q15 records), three alternating rounds.  ENVX="NAME=value ..." is exported for every run.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r03
[ -n "${ENVX:-}" ] && export $ENVX
if [ $# -eq 0 ]; then set -- "c3 --arith q15 --stages 0" "c2 --arith q15 --stages 0" "fir --arith q15" "c4 --arith q15 --stages 0"; fi
for rep in 1 2 3; do for cfg in "$@"; do for v in old new; do
  MSDR_LIB=$PWD/minimal-sdr_amd/lib_ab/libmsdr_$v.so python bench.py --workload $cfg --no-cpu --no-parity --steps 60 --warmup 30 > gpurun_out/r03/abq.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/r03/abq.json'))
print('$cfg', '$v', 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['config'].get('kernel'))"
done; done; done
