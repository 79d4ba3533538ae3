import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
from gpuhelp import msdr
import test_gpu_chain as T
orc = orclib.Oracle(); ctx = msdr.Context(0)
rng = np.random.default_rng(6)
n = 1 << 20
x = rng.integers(-8000, 8001, (1, n)).astype(np.int16)
hi, hq = T._hilbert_pair(256)
oi, oq = T._q15_nco(4, 1)
bq = T._f32_biquads(orc, 2)
want = orc.chain_f32(x[0], orclib.LSB, hi, hq, oi, oq, bq)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 1, hi, hq, mixer=msdr.MIXER_NCO, mode=orclib.LSB, osc_i=oi, osc_q=oq, biquad_coeffs=bq)
    got = T.run_chain(ctx, chain, x, np.float32)[0]
    info = chain.info()
    e = np.abs(got.astype(np.float64) - want)
    thr = 1e-4 * np.sqrt((want ** 2).mean())
    bad = np.nonzero(e > thr)[0]
    if len(bad):
        tiles = np.unique(bad // 1024)
        print(rep, "BAD samples", len(bad), "tiles", tiles[:20], "seg_len_tiles", (n // 1024 + info["time_segments"] - 1) // info["time_segments"], "first bad offsets in tile", (bad[:8] % 1024), "max err", e.max())
    else:
        print(rep, "ok", info["time_segments"], info["grid"], info["block"])
