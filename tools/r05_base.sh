set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05base
for wl in c3 c4; do
  python bench.py --workload $wl --samples 128 --steps 200 --warmup 20 --no-cpu > gpurun_out/r05base/${wl}_b128.json 2> gpurun_out/r05base/${wl}_b128.err || true
  python bench.py --workload $wl --samples 128 --steps 200 --warmup 20 --no-cpu --arith q15 > gpurun_out/r05base/${wl}_b128_q15.json 2> gpurun_out/r05base/${wl}_b128_q15.err || true
done
python bench.py --workload c4 --samples 128 --channels 65536 --steps 200 --warmup 20 --no-cpu > gpurun_out/r05base/c4_b128_64k.json 2> gpurun_out/r05base/c4_b128_64k.err || true
python bench.py --workload c3 --samples 128 --channels 65536 --steps 200 --warmup 20 --no-cpu > gpurun_out/r05base/c3_b128_64k.json 2> gpurun_out/r05base/c3_b128_64k.err || true
rocprofv3 --kernel-trace --stats -d gpurun_out/r05base/prof_c3 -o c3 -- python bench.py --workload c3 --samples 128 --steps 200 --warmup 20 --no-cpu --no-parity > /dev/null 2>&1 || true
rocprofv3 --kernel-trace --stats -d gpurun_out/r05base/prof_c3q -o c3q -- python bench.py --workload c3 --samples 128 --steps 200 --warmup 20 --no-cpu --no-parity --arith q15 > /dev/null 2>&1 || true
ls -R gpurun_out/r05base | head -50
