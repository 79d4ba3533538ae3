#!/usr/bin/env bash
# tools/r05_stamps.sh -- per-phase cycle stamps of chain_mfb_kernel (diagnostic build lib_stamps_block)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1 MSDR_LIB=$PWD/minimal-sdr_amd/lib_stamps_block/libmsdr.so MSDR_STAMP_PRINT=1
for args in "--workload c3" "--workload c4" "--workload c3 --channels 65536" "--workload c4 --channels 65536"; do
  echo "== $args"
  tools/memguard.sh -m 24 -t 120 python3 bench.py $args --samples 128 --steps 4 --warmup 2 --no-cpu --no-parity 2>&1 >/dev/null | grep "mfb stamps" | tail -2
done
