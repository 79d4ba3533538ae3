import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "minimal-sdr_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import msdr, orclib, bench
orc = orclib.Oracle()
ctx = msdr.Context(0)
wl = bench.workload("c5", msdr, 0)
ch, n = 16, 1 << 19
wl["modes"], wl["tapsets"] = wl["modes"][:ch], wl["tapsets"][:ch]
rng = np.random.default_rng(5)
x = rng.integers(-8000, 8001, (ch, n)).astype(np.int16)
for segs in (1, 4, 0):
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, wl["ci"], wl["cq"], mixer=wl["mixer"], modes=wl["modes"], tapsets=wl["tapsets"], biquad_coeffs=wl["bq"], time_segments=segs)
    dx, dy = ctx.to_device(x), ctx.array((ch, n), np.float32)
    chain.process(dx, dy, n)
    got = dy.download()
    print("segs", segs, chain.info())
    cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
    for c in range(ch):
        s = wl["tapsets"][c]
        want = orc.chain_f32(x[c], wl["modes"][c], wl["ci"][s], wl["cq"][s], sin4, cos4, wl["bq"])
        e = np.sqrt(((got[c] - want) ** 2).sum() / (want ** 2).sum())
        bad = np.nonzero(np.abs(got[c] - want) > 1e-4 * np.abs(want).max())[0]
        print(c, "mode", wl["modes"][c], "set", s, "relrms %.3g" % e, "nbad", bad.size, bad[:4], bad[-4:] if bad.size else "", "nan", np.isnan(got[c]).sum())
