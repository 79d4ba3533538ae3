"""Randomised differential test of the arm_fir_f32 stage (fir_f32tq_kernel / fir_f32mf_kernel / fir_kernel<FirF32>) against a float64
convolution: every channel, every output; on a mismatch the (call, channel, tile) map of the error is printed.
    gpurun -- python tests/debug/fuzz_fir_f32.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr, rel_rms  # noqa: E402  (imports torch first)
from scipy.signal import fftconvolve  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = msdr.Context(0)
B = 128
t_end = time.time() + budget
cases = bad = 0
while time.time() < t_end:
    cases += 1
    ntaps = int(rng.integers(1, 320))
    ch = int(rng.choice([1, 5, 70, 70, 70, 300]))
    n = int(rng.integers(1, 80)) * B
    h = (rng.standard_normal(ntaps) * 10.0 ** rng.uniform(-3, 1)).astype(np.float32)
    x = (rng.standard_normal((ch, n)) * 10.0 ** rng.uniform(-6, 6)).astype(np.float32)
    if rng.integers(0, 2):
        x[:, n // 2:] *= np.float32(10.0 ** rng.uniform(-4, 4))
    cuts = sorted(set([0, n] + [int(c) * B for c in rng.integers(1, max(2, n // B), int(rng.integers(0, 4)))]))
    fir = msdr.FirF32(ctx, h, ch)
    got = np.empty_like(x)
    for o, e in zip(cuts[:-1], cuts[1:]):
        m = e - o
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:e])), ctx.to_device(np.full((ch, m), np.nan, np.float32))    # (outputs never written stay NaN)
        fir.process(dx, dy, m)
        got[:, o:e] = dy.download()
    want = fftconvolve(x.astype(np.float64), h.astype(np.float64)[::-1][None, :], axes=1)[:, :n]
    err = np.sqrt(((got - want) ** 2).sum(axis=1) / np.maximum((want ** 2).sum(axis=1), 1e-300))
    err = np.where(np.isnan(err), np.inf, err)
    if not err.max() < 3e-6:
        bad += 1
        print("MISMATCH", dict(seed=seed, case=cases, ntaps=ntaps, ch=ch, n=n, cuts=cuts, kernel=fir.kernel_name(), worst=float(err.max()), rows=int((err > 3e-6).sum())), flush=True)
        shown = 0
        for c in np.nonzero(err > 3e-6)[0][:6]:
            for o, e in zip(cuts[:-1], cuts[1:]):
                for t in range(o, e, 1024):
                    hi = min(e, t + 1024)
                    d = got[c, t:hi] - want[c, t:hi]
                    scale = np.sqrt((want[c, o:e] ** 2).mean()) + 1e-300
                    badm = ~(np.abs(d) <= 1e-5 * scale)
                    if badm.any() and shown < 40:
                        shown += 1
                        idx = np.nonzero(badm)[0]
                        print("   channel %d call [%d, %d) tile at %d (+%d): %d bad samples in +%d .. +%d, %d of them NaN (never written); got %.6g want %.6g" % (
                            c, o, e, t, hi - t, idx.size, idx[0], idx[-1], int(np.isnan(got[c, t:hi]).sum()), got[c, t + idx[0]], want[c, t + idx[0]]), flush=True)
    fir.close()
    if cases % 200 == 0:
        print("cases", cases, "bad", bad, flush=True)
print("fuzz done: %d cases, %d mismatches (seed %d)" % (cases, bad, seed))
sys.exit(1 if bad else 0)
