cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MSDR_BENCH_NO_POWER=1
for chs in 1024 4096 8192 16384 32768 65536; do
 for blk in 0 1; do
  MSDR_BIQUAD_BLK=$blk tools/memguard.sh -m 24 -t 120 python3 bench.py --workload c3 --arith q15 --channels $chs --samples 128 --steps 300 --warmup 50 --no-cpu --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ch $chs blk $blk', d['value'],'Msps', r.get('tick_us'),'us/tick')"
 done
done
