"""Where does one fuzz_f32_truth.py case differ?  (GPU box, from the repository root)
    gpurun -- python tests/debug/dbg_truth_case.py seed case
Rebuilds the case, then prints for its first AM (else first) channel: the error against float64 per 1/8 of the block, the same with
time_segments forced to 1, and the error of the chain without its cascade (the cascade's input)."""
import os, sys
import numpy as np
from scipy.signal import lfilter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr, rel_rms  # noqa: E402
import orclib  # noqa: E402

seed, case, B = int(sys.argv[1]), int(sys.argv[2]), 128
orc = orclib.Oracle()
ctx = msdr.Context(0)
rng = np.random.default_rng([seed, case])
ntaps = int(rng.integers(2, 300)); ch = int(rng.choice([1, 3, 40])); n = int(rng.integers(2, 80)) * B
hi = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
hq = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
modes = rng.choice([orclib.AM, orclib.LSB, orclib.USB, orclib.CW], ch).astype(np.int32)
if rng.integers(0, 2):
    hq = hi.copy()
mixer = int(rng.integers(0, 2)); P = int(rng.choice([1, 2, 4, 8, 16, 32, 64])); k = np.arange(B)
if mixer:
    oi = (np.round(32767 * np.sin(2 * np.pi * k / P)).astype(np.int16) / 32768.0).astype(np.float32)
    oq = (np.round(32767 * np.cos(2 * np.pi * k / P)).astype(np.int16) / 32768.0).astype(np.float32)
else:
    oi, oq = np.array([0, 1, 0, -1], np.float32)[k % 4], np.array([1, 0, -1, 0], np.float32)[k % 4]
stages = int(rng.integers(1, 5)); rows = []
for _ in range(stages):
    kind = int(rng.choice([orclib.BQ_LOWPASS, orclib.BQ_NOTCH, orclib.BQ_HIGHPASS]))
    c_ = orc.biquad_design(kind, np.float32(rng.uniform(800, 9000)), float(rng.uniform(0.5, 8))).astype(np.float64) / 2 ** 30
    rows.append([c_[0], c_[1], c_[2], -c_[3], -c_[4]])
bq = np.array(rows, np.float32)
kindx = rng.integers(0, 3)
x = (rng.integers(-32768, 32768, (ch, n)) if kindx == 0 else rng.integers(-300, 301, (ch, n)) if kindx == 1
     else np.sign(rng.standard_normal((ch, n))) * 32767).astype(np.int16)
segs = int(rng.choice([0, 0, 1, 3]))
am = np.where(modes == orclib.AM)[0]
c = int(am[0]) if am.size else 0
print("taps %d ch %d n %d stages %d mixer %d input kind %d time_segments %d channel %d mode %d" % (ntaps, ch, n, stages, mixer, kindx, segs, c, modes[c]))


def f64(bq_):
    nn = np.arange(n); xf = x[c].astype(np.float64) / 32768
    ai = lfilter(hi.astype(np.float64)[::-1], [1.0], xf * oq.astype(np.float64)[nn % oq.size])
    aq = lfilter(hq.astype(np.float64)[::-1], [1.0], xf * oi.astype(np.float64)[nn % oi.size])
    m = int(modes[c])
    d = ai - aq if m == orclib.LSB else ai + aq if m == orclib.USB else np.sqrt(ai * ai + aq * aq)
    for r in ([] if bq_ is None else np.asarray(bq_, np.float64)):
        d = lfilter(r[:3], [1.0, -r[3], -r[4]], d)
    return d


def run(bq_, ts):
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, hi, hq, mixer=mixer, modes=modes, osc_i=oi if mixer else None, osc_q=oq if mixer else None,
                       biquad_coeffs=bq_, time_segments=ts)
    dx, dy = ctx.to_device(x), ctx.array((ch, n), np.float32)
    chain.process(dx, dy, n)
    got = dy.download()[c]
    info = chain.info()
    chain.close()
    return got, info


for name, bq_, ts in (("as drawn", bq, segs), ("time_segments 1", bq, 1), ("no cascade", None, segs)):
    got, info = run(bq_, ts)
    t = f64(bq_)
    want = orc.chain_f32(x[c], modes[c], hi, hq, oi, oq, bq_)
    w = np.array_split(np.arange(n), 8)
    print("%-16s %s segs %d: gpu-f64 %.2e oracle-f64 %.2e | f64 rms %.3e mean %.3e" % (name, info["kernel"], info["time_segments"], rel_rms(got, t), rel_rms(want, t),
          np.sqrt((t ** 2).mean()), t.mean()))
    print("   abs rms error per eighth, gpu   :", " ".join("%.1e" % np.sqrt(((got[i] - t[i]) ** 2).mean()) for i in w))
    print("   abs rms error per eighth, oracle:", " ".join("%.1e" % np.sqrt(((want[i] - t[i]) ** 2).mean()) for i in w))
    print("   signal rms per eighth           :", " ".join("%.1e" % np.sqrt((t[i] ** 2).mean()) for i in w))
    e = got - t
    print("   gpu error: mean %.2e, first 8 %s" % (e.mean(), np.array2string(e[:8], precision=2)))
    print("   gpu distinct values in the last eighth: %d, oracle: %d" % (np.unique(got[w[-1]]).size, np.unique(want[w[-1]]).size))
