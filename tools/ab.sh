#!/usr/bin/env bash
# tools/ab.sh <bench args...> -- same-box A/B of lib/libmsdr.so against lib_alt/libmsdr_base.so (a copy taken before a change)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for rep in 1 2; do for v in base new; do
  if [ $v = base ]; then export MSDR_LIB=$PWD/minimal-sdr_amd/lib_alt/libmsdr_base.so; else unset MSDR_LIB; fi
  python bench.py "$@" --no-cpu > gpurun_out/ab_x.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/ab_x.json'))
print('$v', '$*', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline']['frac'])"
done; done
