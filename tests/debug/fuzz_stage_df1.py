"""arm_biquad_cascade_df1_f32 stage mirror (msdr_biquad_df1_f32_*) with random cascades, call lengths and channel counts, judged like
fuzz_f32_truth.py: against the fp32 oracle (1e-5) and, where that is exceeded, against a float64 evaluation.
    gpurun -- python tests/debug/fuzz_stage_df1.py [seconds] [seed] [only_case]
Long single streams exercise the time segmentation of the stage (warm-up from the pole radius)."""
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
orc = orclib.Oracle()
ctx = msdr.Context(0)
t_end = time.time() + budget
case = over = defects = 0
worst = 0.0
while time.time() < t_end:
    case += 1
    if only >= 0:
        if case > 1:
            break
        case = only
    rng = np.random.default_rng([seed, case])
    ch = int(rng.choice([1, 1, 3, 64, 100]))
    n = int(rng.integers(1, 1 << 19)) if ch == 1 else int(rng.integers(1, 5000))
    stages = int(rng.integers(1, 5))
    rows = []
    for _ in range(stages):
        kind = int(rng.choice([orclib.BQ_LOWPASS, orclib.BQ_NOTCH, orclib.BQ_HIGHPASS]))
        c_ = orc.biquad_design(kind, np.float32(rng.uniform(300, 9000)), float(rng.uniform(0.5, 20))).astype(np.float64) / 2 ** 30
        rows.append([c_[0], c_[1], c_[2], -c_[3], -c_[4]])
    bq = np.array(rows, np.float32)
    amp = float(rng.choice([1e-3, 1.0, 3e4]))
    x = (amp * rng.standard_normal((ch, n))).astype(np.float32)
    if rng.integers(0, 2):
        x += np.float32(amp)                                   # a DC term (what an envelope carries)
    st = msdr.BiquadDf1F32(ctx, bq, ch)
    got = np.empty_like(x)
    o = 0
    while o < n:
        m = int(min(n - o, rng.choice([1, 2, 3, 100, 128, 1000, 4096, 100000, n])))
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.float32)
        st.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
        o += m
    for c in rng.choice(ch, min(ch, 2), replace=False):
        want = orc.biquad_df1_blocks(bq, x[c], 128)
        e_go = rel_rms(got[c], want)
        if e_go < 1e-5 and only < 0:
            continue
        over += 1
        t = x[c].astype(np.float64)
        for r in bq.astype(np.float64):
            t = lfilter(r[:3], [1.0, -r[3], -r[4]], t)
        e_gpu, e_orc = rel_rms(got[c], t), rel_rms(want, t)
        worst = max(worst, e_gpu / max(e_orc, 1e-12))
        bad = e_gpu > 2 * e_orc + msdr.biquad_cascade_info(bq)[1] + 1e-6       # the contract of include/msdr.h (fp32_noise: the cascade's own figure)
        defects += bad
        print("%s case %d: gpu-oracle %.2e | gpu-f64 %.2e  oracle-f64 %.2e | ch %d n %d stages %d amp %g" % ("DEFECT" if bad else "inherent", case, e_go, e_gpu, e_orc, ch, n, stages, amp), flush=True)
    st.close()
print("fuzz_stage_df1 done: %d cases, %d over 1e-5 vs the fp32 oracle, %d of them further from float64 than the oracle is (worst e_gpu/e_orc %.2f), seed %d"
      % (case, over, defects, worst, seed))
