#!/usr/bin/env bash
# tools/r05_blk.sh -- block-cadence tests + bench records on the GPU box
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
O=gpurun_out/r05blk; mkdir -p $O
tools/memguard.sh -m 24 -t 600 python3 -m pytest tests/test_gpu_block.py -q -m gpu > $O/pytest_block.log 2>&1
echo "pytest rc=$?" >> $O/pytest_block.log
tail -25 $O/pytest_block.log
export MSDR_BENCH_NO_POWER=1
for wl in c3 c4; do
  tools/memguard.sh -m 24 -t 300 python3 bench.py --workload $wl --samples 128 --steps 200 --warmup 20 --no-cpu > $O/${wl}_b128.json 2> $O/${wl}_b128.err
done
tools/memguard.sh -m 24 -t 300 python3 bench.py --workload c4 --samples 128 --channels 65536 --steps 200 --warmup 20 --no-cpu > $O/c4_b128_64k.json 2> $O/c4_b128_64k.err
tools/memguard.sh -m 24 -t 300 python3 bench.py --workload c3 --samples 128 --channels 65536 --steps 200 --warmup 20 --no-cpu > $O/c3_b128_64k.json 2> $O/c3_b128_64k.err
for f in $O/*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print(sys.argv[1], d['value'], 'Msps', round(d['ms_per_step']*1000,2), 'us/step; kernel_us', round(r['kernel_ms']*1000,2), d['config'].get('kernel'), d.get('parity'))
except Exception as e: print(sys.argv[1], 'ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
