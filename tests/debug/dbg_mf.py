import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "minimal-sdr_amd/python")
import numpy as np
from gpuhelp import *   # imports torch first
import orclib, msdr
orc = orclib.Oracle()
def run_chain(ctx, chain, x, out_dtype, block=None):
    ch, n = x.shape
    out = np.empty((ch, n), out_dtype)
    step = block or n
    for o in range(0, n, step):
        m = min(step, n - o)
        dx, dy = ctx.to_device(x[:, o:o + m]), ctx.array((ch, m), out_dtype)
        chain.process(dx, dy, m)
        out[:, o:o + m] = dy.download()
    return out

ctx = msdr.Context(0)
rng = np.random.default_rng(1)
lp = (np.sinc(2 * 2800 / 24000 * (np.arange(62) - 30.5)) * np.kaiser(62, 7.0)).astype(np.float32); lp /= lp.sum()
x = rng.integers(-8000, 8001, (1, 512)).astype(np.int16)
cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
for mode in (orclib.AM, orclib.LSB):
    for stages in (0, 1):
        bq = None
        if stages:
            c = orc.biquad_design(orclib.BQ_LOWPASS, np.float32(5400 * 0.9), 0.54).astype(np.float64) / 2 ** 30
            bq = np.array([[c[0], c[1], c[2], -c[3], -c[4]]], np.float32)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, 1, lp, lp, mixer=msdr.MIXER_FS4, mode=mode, biquad_coeffs=bq)
        got = run_chain(ctx, chain, x, np.float32, None)
        want = orc.chain_f32(x[0], mode, lp, lp, sin4, cos4, bq)
        err = np.abs(got[0] - want) / (np.abs(want).max())
        bad = np.nonzero(err > 1e-4)[0]
        print("mode", mode, "stages", stages, chain.info()["kernel"], "bad", len(bad), bad[:40])
        if len(bad):
            i = bad[0]
            print("  got ", got[0][i:i+8]); print("  want", want[i:i+8])
