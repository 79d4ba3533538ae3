#!/usr/bin/env bash
# tools/r05_shape_ab.sh -- the MFMA shape of chain_mfw_kernel's burst, same box, alternating builds (make -C minimal-sdr_amd shape-probe):
#   product | shape1 = k-steps in pairs, 2 x 3 v_mfma_f32_32x32x16_f16 (control: the product's instructions in the probe's pipeline)
#           | shape2 = the same pairs as 2 x 2 register-blocked v_mfma_f32_16x16x32_f16 (results meaningless; time / clock / power compared)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
P=$PWD/minimal-sdr_amd
for round in 1 2 3; do
  for wl in c2 c4 c3; do
    for v in product shape1 shape2; do
      lib=$P/lib/libmsdr.so; par=""
      [ $v = shape1 ] && lib=$P/lib_shape1/libmsdr.so
      [ $v = shape2 ] && { lib=$P/lib_shape2/libmsdr.so; par="--no-parity"; }
      MSDR_LIB=$lib tools/memguard.sh -m 24 -t 200 python3 bench.py --workload $wl --steps 30 --warmup 10 --no-cpu $par 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('round $round $wl $v kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'sclk', r.get('sclk_mhz'), 'W', r.get('power_w'), 'parity', (d.get('parity') or {}).get('rel_rms_worst'))"
    done
  done
done
