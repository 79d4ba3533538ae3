#!/usr/bin/env bash
# tools/r05_base.sh -- block-cadence (128 samples per call) baseline of the chain: per-tick time and the kernels behind it
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
export MSDR_BENCH_NO_POWER=1
O=gpurun_out/r05base; mkdir -p $O
for wl in c3 c4; do
  python3 bench.py --workload $wl --samples 128 --steps 200 --warmup 20 --no-cpu > $O/${wl}_b128.json 2> $O/${wl}_b128.err
  python3 bench.py --workload $wl --samples 128 --steps 200 --warmup 20 --no-cpu --arith q15 > $O/${wl}_b128_q15.json 2> $O/${wl}_b128_q15.err
done
python3 bench.py --workload c4 --samples 128 --channels 65536 --steps 200 --warmup 20 --no-cpu > $O/c4_b128_64k.json 2> $O/c4_b128_64k.err
python3 bench.py --workload c3 --samples 128 --channels 65536 --steps 200 --warmup 20 --no-cpu > $O/c3_b128_64k.json 2> $O/c3_b128_64k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 bench.py --workload c3 --samples 128 --steps 200 --warmup 20 --no-cpu --no-parity > $O/prof_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3q -- python3 bench.py --workload c3 --samples 128 --steps 200 --warmup 20 --no-cpu --no-parity --arith q15 > $O/prof_c3q.log 2>&1
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
ls -R $O | head -40
