"""Where does the fp32 chain's parallel IIR evaluation lose accuracy?  GPU vs the sequential fp32 oracle vs a float64 model,
for cascades of low-pass / notch / high-pass sections (run from the repository root on a GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr, rel_rms  # noqa: E402
import orclib  # noqa: E402
from scipy.signal import lfilter  # noqa: E402

orc = orclib.Oracle()
ctx = msdr.Context(0)
rng = np.random.default_rng(0)
B = 128
n = 40 * B
x = rng.integers(-12000, 12001, (2, n)).astype(np.int16)
k = np.arange(100)
proto = np.sinc(1920 / 24000 * (k - 49.5)) * np.kaiser(100, 6.0); proto /= proto.sum()
hi = (2 * proto * np.cos(2 * np.pi * 1330 / 24000 * (k - 49.5) + np.pi / 4)).astype(np.float32)
hq = (2 * proto * np.cos(2 * np.pi * 1330 / 24000 * (k - 49.5) - np.pi / 4)).astype(np.float32)
cos4, sin4 = np.array([1, 0, -1, 0], np.float32)[np.arange(B) % 4], np.array([0, 1, 0, -1], np.float32)[np.arange(B) % 4]
names = {orclib.BQ_LOWPASS: "LP", orclib.BQ_NOTCH: "NT", orclib.BQ_HIGHPASS: "HP", orclib.BQ_BANDPASS: "BP"}


def truth(xc, mode, bq):
    xs = xc.astype(np.float64) / 32768.0
    t = np.arange(xs.size)
    i_, q_ = xs * cos4.astype(np.float64)[t % 4], xs * sin4.astype(np.float64)[t % 4]
    fi = np.convolve(i_, hi[::-1].astype(np.float64))[:xs.size]
    fq = np.convolve(q_, hq[::-1].astype(np.float64))[:xs.size]
    d = fi - fq if mode == orclib.LSB else np.sqrt(fi * fi + fq * fq)
    for c in bq.astype(np.float64):
        d = lfilter(c[:3], [1.0, -c[3], -c[4]], d)
    return d


print("%-28s %-4s %10s %10s %10s" % ("cascade (kind f Q)", "mode", "gpu-orc", "gpu-truth", "orc-truth"))
for mode, mname in ((orclib.LSB, "LSB"), (orclib.AM, "AM")):
    for spec in ([(0, 5400, 0.54)], [(0, 5400, 0.54), (3, 3000, 15)], [(1, 300, 0.7)], [(1, 300, 0.7), (1, 300, 0.7)], [(1, 1000, 4), (1, 1500, 4), (1, 800, 2)], [(2, 1000, 4), (2, 1500, 4)],
                 [(0, 800, 8)], [(0, 800, 8), (0, 900, 8)], [(3, 1000, 8), (3, 2000, 8), (3, 3000, 8), (3, 4000, 8)], [(1, 2000, 1), (0, 3000, 1), (1, 500, 3), (0, 6000, 5)],
                 [(0, 5400, 0.54), (0, 5400, 1.3), (0, 5400, 0.54), (0, 5400, 1.3)]):
        rows = []
        for kind, f, q in spec:
            c_ = orc.biquad_design(kind, np.float32(f * orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0), q).astype(np.float64) / 2 ** 30
            rows.append([c_[0], c_[1], c_[2], -c_[3], -c_[4]])
        bq = np.array(rows, np.float32)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, 2, hi, hq, mode=mode, biquad_coeffs=bq)
        dx, dy = ctx.to_device(x), ctx.array((2, n), np.float32)
        chain.process(dx, dy, n)
        got = dy.download()[0]
        want = orc.chain_f32(x[0], mode, hi, hq, sin4, cos4, bq)
        tr = truth(x[0], mode, bq)
        label = " ".join("%s%d/%g" % (names[k_], f_, q_) for k_, f_, q_ in spec)
        print("%-28s %-4s %10.2e %10.2e %10.2e  %s" % (label[:28], mname, rel_rms(got, want), rel_rms(got, tr), rel_rms(want, tr), chain.info()["kernel"]))
        chain.close()
