#!/usr/bin/env bash
# tools/ab_named.sh "<lib names>" <workloads...> -- same-box comparison of minimal-sdr_amd/lib_ab/libmsdr_<name>.so, three alternating rounds
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r02
names=$1; shift
for rep in 1 2 3; do for w in "$@"; do for v in $names; do
  MSDR_LIB=$PWD/minimal-sdr_amd/lib_ab/libmsdr_$v.so python bench.py --workload $w --no-cpu --no-parity --steps ${STEPS:-100} --warmup 30 > gpurun_out/r02/ab.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/r02/ab.json'))
print('$w', '$v', 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
done; done; done
