"""Debug aid: error profile of the wave-stream matrix-core kernel against the oracle (run on the GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
from gpuhelp import msdr
import test_gpu_chain as T

orc = orclib.Oracle()
ctx = msdr.Context(0)
rng = np.random.default_rng(41)
ch, n = 40, int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ntaps = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lp = (np.sinc(2 * 2800 / 24000 * (np.arange(ntaps) - (ntaps - 1) / 2)) * np.kaiser(ntaps, 7.0)).astype(np.float32)
lp /= lp.sum()
hi, hq = T._hilbert_pair(ntaps)
modes = np.array([orclib.AM if (c * 2654435761) & 1 else orclib.LSB for c in range(ch)], np.int32)
tapsets = np.array([0 if m == orclib.AM else 1 for m in modes], np.int32)
x = rng.integers(-8000, 8001, (ch, n)).astype(np.int16)
bq = T._f32_biquads(orc, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, [lp, hi], [lp, hq], mixer=msdr.MIXER_FS4, modes=modes, tapsets=tapsets, biquad_coeffs=bq)
got = T.run_chain(ctx, chain, x, np.float32)
print(chain.info())
cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
for c in range(4):
    ti, tq = ([lp, hi][tapsets[c]], [lp, hq][tapsets[c]])
    want = orc.chain_f32(x[c], modes[c], ti, tq, sin4, cos4, bq)
    e = got[c].astype(np.float64) - want
    blk = 256
    prof = [np.sqrt((e[i:i + blk] ** 2).mean()) / np.sqrt((want ** 2).mean()) for i in range(0, n, blk)]
    bad = [(i, "%.1e" % p) for i, p in enumerate(prof) if p > 1e-5]
    print(c, "AM" if modes[c] == orclib.AM else "LSB", "%.2e" % T.rel_rms(got[c], want), bad[:12])
    if c < 2:
        thr = 1e-4 * np.sqrt((want ** 2).mean())
        idx = np.nonzero(np.abs(e) > thr)[0]
        print("   bad idx:", idx[:10], "...", idx[-10:], len(idx))
        for t in range(1024, n, 1024):
            seg = np.abs(e[t:t + 1024]) > thr
            w = np.nonzero(seg)[0]
            print("   tile", t, "bad within-tile offsets min/max/count", (w.min(), w.max(), len(w)) if len(w) else None)
