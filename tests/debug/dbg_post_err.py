import sys, os
sys.path.insert(0, "tests")
import numpy as np
import test_gpu_chain_post as T
from gpuhelp import msdr, rel_rms
import orclib
ctx = msdr.Context(0); orc = orclib.Oracle()
for ntaps in (62, 256):
    for stages in (0, 2):
        rng = np.random.default_rng(7 + ntaps + stages)
        lp = T._lowpass(ntaps)
        modes = [orclib.SYNCAM, orclib.AM, orclib.LSB, orclib.SYNCAM, orclib.USB, orclib.SYNCAM]
        anr = [0, 1, 2, 2, 0, 1]
        n = 6000
        x = T._am_if(rng, len(modes), n, 35.0)
        bq = T._bq(orc, stages)
        cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, len(modes), lp, lp, mixer=msdr.MIXER_FS4, modes=modes, biquad_coeffs=bq, flags=msdr.CHAIN_SYNCAM_PLL)
        chain.set_anr(anr)
        splits = [2048, 1000, 129, 64, 10000]
        got = T._run(ctx, chain, x, splits)
        errs = []
        for c in range(len(modes)):
            st = {}
            want = np.concatenate([orc.chain_f32(x[c, o:o + m], modes[c], lp, lp, sin4, cos4, bq, state=st, pll=(modes[c] == orclib.SYNCAM), anr_on=anr[c]) for o, m in T._segments(n, splits)])
            errs.append(float("%.2g" % rel_rms(got[c], want)))
        print(ntaps, stages, errs)
