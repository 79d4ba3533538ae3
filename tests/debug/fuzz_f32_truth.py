"""fp32 chain with random cascades, judged against a float64 evaluation of the same chain (run from the repository root on a GPU box):
    gpurun -- python tests/debug/fuzz_f32_truth.py [seconds] [seed] [only_case]
The oracle is a sequential fp32 program: behind a cascade with resonant sections or deep stop bands its own rounding noise is a
visible fraction of what is left of the signal, and two correct fp32 evaluations differ by that much.  For every case this prints
nothing unless |gpu - oracle| exceeds the 1e-5 tolerance; then it compares both with the float64 result:
    e_gpu = |gpu - f64| / |f64|      e_orc = |oracle - f64| / |f64|
A case counts as a DEFECT only if the library is further from the float64 result than the contract of include/msdr.h allows
(e_gpu > 2 e_orc + fp32_noise + 1e-6 lvl, fp32_noise = the cascade's own figure from msdr_biquad_df1_f32_cascade_info; lvl = the level of the
cascade's INPUT over its output, >= 1: where the cascade removes most of its input -- a constant envelope behind a high-pass: two taps and a
sign-only input do that -- the 1e-5 / 1e-6 of the contract refer to the input's level; round 4's seed 4206 drew five such cases, judged as
defects by this script's first version, which lacked the clause: tests/test_gpu_f32_contract.py holds them now).
Each case draws from default_rng([seed, case])."""
import os, sys, time
import numpy as np
from scipy.signal import lfilter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr, rel_rms  # noqa: E402  (imports torch first)
import orclib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
LONG = os.environ.get("FUZZ_LONG", "0") == "1"      # one channel, 2^17 .. 2^20 samples per call, Q up to 20: the automatic time segmentation
orc = orclib.Oracle()
ctx = msdr.Context(0)
B = 128


def truth64(x, mode, hi, hq, oi, oq, bq):
    """orc_chain_f32 (oracle/msdr_oracle.c) with every operation in float64."""
    n = np.arange(x.size)
    xf = x.astype(np.float64) * (1.0 / 32768)
    wi, wq = xf * oq.astype(np.float64)[n % oq.size], xf * oi.astype(np.float64)[n % oi.size]
    ai = lfilter(hi.astype(np.float64)[::-1], [1.0], wi)
    aq = lfilter(hq.astype(np.float64)[::-1], [1.0], wq)
    d = ai - aq if mode == orclib.LSB else ai + aq if mode == orclib.USB else np.sqrt(ai * ai + aq * aq)
    if bq is not None:
        for c in np.asarray(bq, np.float64):
            d = lfilter(c[:3], [1.0, -c[3], -c[4]], d)
    return d


t_end = time.time() + budget
case = over = defects = attenuating = 0
worst = 0.0
while time.time() < t_end:
    case += 1
    if only >= 0:
        if case > 1:
            break
        case = only
    rng = np.random.default_rng([seed, case])
    ntaps = int(rng.integers(2, 300))
    ch = int(rng.choice([1, 3, 40]))
    n = int(rng.integers(2, 80)) * B
    if LONG:
        ch, n = 1, int(rng.integers(1 << 10, 1 << 13)) * B
    hi = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    hq = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    modes = rng.choice([orclib.AM, orclib.LSB, orclib.USB, orclib.CW], ch).astype(np.int32)
    if rng.integers(0, 2):
        hq = hi.copy()
    mixer = int(rng.integers(0, 2))
    P = int(rng.choice([1, 2, 4, 8, 16, 32, 64]))
    k = np.arange(B)
    if mixer:
        oi = (np.round(32767 * np.sin(2 * np.pi * k / P)).astype(np.int16) / 32768.0).astype(np.float32)
        oq = (np.round(32767 * np.cos(2 * np.pi * k / P)).astype(np.int16) / 32768.0).astype(np.float32)
    else:
        oi, oq = np.array([0, 1, 0, -1], np.float32)[k % 4], np.array([1, 0, -1, 0], np.float32)[k % 4]
    stages = int(rng.integers(1, 5))
    rows = []
    for _ in range(stages):
        kind = int(rng.choice([orclib.BQ_LOWPASS, orclib.BQ_NOTCH, orclib.BQ_HIGHPASS]))
        c_ = orc.biquad_design(kind, np.float32(rng.uniform(800, 9000)), float(rng.uniform(0.5, 20 if LONG else 8))).astype(np.float64) / 2 ** 30
        rows.append([c_[0], c_[1], c_[2], -c_[3], -c_[4]])
    bq = np.array(rows, np.float32)
    kindx = rng.integers(0, 3)
    x = (rng.integers(-32768, 32768, (ch, n)) if kindx == 0 else rng.integers(-300, 301, (ch, n)) if kindx == 1
         else np.sign(rng.standard_normal((ch, n))) * 32767).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, hi, hq, mixer=mixer, modes=modes, osc_i=oi if mixer else None, osc_q=oq if mixer else None,
                       biquad_coeffs=bq, time_segments=int(rng.choice([0, 0, 1, 3])))
    dx, dy = ctx.to_device(x), ctx.array((ch, n), np.float32)
    chain.process(dx, dy, n)
    got = dy.download()
    for c in rng.choice(ch, min(ch, 3), replace=False):
        want = orc.chain_f32(x[c], modes[c], hi, hq, oi, oq, bq)
        e_go = rel_rms(got[c], want)
        if e_go < 1e-5 and only < 0:
            continue
        over += 1
        pre = orc.chain_f32(x[c], modes[c], hi, hq, oi, oq, None)             # the cascade's input
        lvl = max(1.0, float(np.sqrt((pre.astype(np.float64) ** 2).mean() / max((want.astype(np.float64) ** 2).mean(), 1e-300))))
        t = truth64(x[c], int(modes[c]), hi, hq, oi, oq, bq)
        e_gpu, e_orc = rel_rms(got[c], t), rel_rms(want, t)
        if e_go < 1e-5 * lvl and only < 0:
            # inside the contract by its input-level clause; the float64 figures are printed all the same (a reader can see how the two fp32
            # evaluations sit against the exact result)
            attenuating += 1
            print("attenuating case %d: gpu-oracle %.2e (level %.1f) | gpu-f64 %.2e  oracle-f64 %.2e | taps %d stages %d %s"
                  % (case, e_go, lvl, e_gpu, e_orc, ntaps, stages, chain.info()["kernel"]), flush=True)
            continue
        worst = max(worst, e_gpu / max(e_orc, 1e-12))
        bad = e_gpu > 2 * e_orc + msdr.biquad_cascade_info(bq)[1] + 1e-6 * lvl  # the contract of include/msdr.h (fp32_noise: the cascade's own figure; lvl: its input's level)
        defects += bad
        print("%s case %d: gpu-oracle %.2e | gpu-f64 %.2e  oracle-f64 %.2e | taps %d stages %d P %d mixer %d mode %d %s n %d segs %d"
              % ("DEFECT" if bad else "inherent", case, e_go, e_gpu, e_orc, ntaps, stages, P, mixer, int(modes[c]), chain.info()["kernel"], n,
                 chain.info()["time_segments"]), flush=True)
    chain.close()
print("fuzz_f32_truth done: %d cases, %d over 1e-5 vs the fp32 oracle: %d within 1e-5 of the cascade's input level, %d of the others beyond the contract against float64 (worst e_gpu/e_orc %.2f), seed %d"
      % (case, over, attenuating, defects, worst, seed))
