"""fp32 chain with the SYNCAM PLL (MSDR_CHAIN_SYNCAM_PLL) and the LMS notch / noise reduction (msdr_chain_set_anr) -- DESIGN.md 4.6:
random tap counts / channel counts / modes / oscillators / cascades / call lengths, a retune in the middle now and then, against
orc_chain_f32_post_run.  Channels without the LMS filter: 1e-5 (PLL channels once the loop has settled).  LMS channels are checked
in two parts, because the filter is discontinuous in its input (its leak control decides per sample: a 3e-7 difference in front of it
has been seen to come out as 4e-3): (1) the audio IN FRONT of the filter, from a second chain with the filter off, against the oracle
to 1e-5; (2) the chain's output against the oracle's filter + cascade applied to THAT audio -- the filter's arithmetic alone, 1e-5.
    gpurun -- python tests/debug/fuzz_pll_anr_f32.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpuhelp import msdr  # noqa: E402  (imports torch first)
import orclib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = orclib.Oracle()
ctx = msdr.Context(0)
CORR = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
t_end = time.time() + budget
case = bad = 0
ONLY = int(os.environ.get("FUZZ_ONLY", "0"))        # FUZZ_ONLY=<case>: that case alone, with diagnostics
if ONLY:
    case = ONLY - 1
worst = {"plain": 0.0, "pll": 0.0, "lms": 0.0}


def truth64(x, mode, hi, hq, oi, oq, bq):
    """orc_chain_f32 with every operation in float64 (as tests/debug/fuzz_kernels.py)"""
    from scipy.signal import lfilter
    nn = np.arange(x.size)
    xf = x.astype(np.float64) * (1.0 / 32768)
    wi, wq = xf * oq.astype(np.float64)[nn % oq.size], xf * oi.astype(np.float64)[nn % oi.size]
    ai = lfilter(hi.astype(np.float64)[::-1], [1.0], wi)
    aq = lfilter(hq.astype(np.float64)[::-1], [1.0], wq)
    d = ai - aq if mode == orclib.LSB else ai + aq if mode == orclib.USB else np.sqrt(ai * ai + aq * aq)
    if bq is not None:
        for cc in np.asarray(bq, np.float64):
            d = lfilter(cc[:3], [1.0, -cc[3], -cc[4]], d)
    return d


def _cascade(v, bq):
    """arm_biquad_cascade_df1_f32 as the oracle's chain runs it (sequential fp32, zero state)"""
    out = np.empty_like(v)
    st = np.zeros((len(bq), 4), np.float32)
    c = np.asarray(bq, np.float32)
    for i in range(v.size):
        d = np.float32(v[i])
        for sidx in range(len(bq)):
            q = st[sidx]
            y = np.float32(np.float32(np.float32(np.float32(np.float32(c[sidx, 0] * d) + np.float32(c[sidx, 1] * q[0])) + np.float32(c[sidx, 2] * q[1])) + np.float32(c[sidx, 3] * q[2])) + np.float32(c[sidx, 4] * q[3]))
            q[1] = q[0]; q[0] = d; q[3] = q[2]; q[2] = y
            d = y
        out[i] = d
    return out


def err(got, want, pre=None):
    ref = np.sqrt((want.astype(np.float64) ** 2).sum())
    if pre is not None:
        ref = max(ref, np.sqrt((pre.astype(np.float64) ** 2).sum()))
    return float(np.sqrt(((got.astype(np.float64) - want) ** 2).sum()) / max(ref, 1e-300))


while time.time() < t_end:
    case += 1
    rng = np.random.default_rng([seed, case])
    ntaps = int(rng.integers(2, 260))
    ch = int(rng.choice([1, 3, 8, 70]))
    n = int(rng.integers(300, 5000))
    t = np.arange(n)
    k = np.arange(ntaps) - (ntaps - 1) / 2
    lp = (np.sinc(2 * rng.uniform(1500, 4000) / 24000 * k) * np.kaiser(ntaps, 7.0)).astype(np.float32)
    lp = (lp / lp.sum()).astype(np.float32)
    hq = lp.copy() if rng.integers(0, 3) else (lp * np.float32(0.9)).astype(np.float32)
    modes = rng.integers(0, 5, ch).astype(np.int32)
    anr_on = rng.integers(0, 3, ch).astype(np.int32) if rng.integers(0, 2) else np.zeros(ch, np.int32)
    pll_flag = bool(rng.integers(0, 4))
    stages = int(rng.integers(0, 3))
    bq = None
    if stages:
        rows = []
        for kind, f, q in ((orclib.BQ_LOWPASS, 5400 * CORR, 0.54), (orclib.BQ_NOTCH, 3000 * CORR, 15.0))[:stages]:
            c_ = orc.biquad_design(kind, np.float32(f), q).astype(np.float64) / 2 ** 30
            rows.append([c_[0], c_[1], c_[2], -c_[3], -c_[4]])
        bq = np.array(rows, np.float32)
    nco = bool(rng.integers(0, 3) == 0)
    P = 4
    if nco:
        P = int(rng.choice([4, 8, 64, 128]))
        kk = np.arange(128)
        oi = (np.round(32767 * np.sin(2 * np.pi * kk / P)).astype(np.int16) / 32768.0).astype(np.float32)
        oq = (np.round(32767 * np.cos(2 * np.pi * kk / P)).astype(np.int16) / 32768.0).astype(np.float32)
    else:
        oi, oq = np.array([0, 1, 0, -1], np.float32), np.array([1, 0, -1, 0], np.float32)
    # an AM carrier within +-80 Hz of the oscillator (24000 / P Hz): inside the filter, something for the PLL to lock to -- with the carrier
    # in the stop band the channel carries leakage only, the PLL wanders, and two fp32 evaluations part company for good reasons
    x = np.empty((ch, n), np.int16)
    for c in range(ch):
        car = 24000.0 / P + rng.uniform(-80, 80)
        x[c] = (rng.uniform(500, 14000) * (1 + 0.5 * np.sin(2 * np.pi * rng.uniform(100, 900) * t / 24000)) * np.cos(2 * np.pi * car * t / 24000 + c)
                + rng.uniform(0, 2000) * np.cos(2 * np.pi * (24000.0 / P + 1000.0) * t / 24000) + rng.integers(-200, 201, n)).clip(-32768, 32767).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, hq, mixer=msdr.MIXER_NCO if nco else msdr.MIXER_FS4, modes=modes, osc_i=oi if nco else None,
                       osc_q=oq if nco else None, biquad_coeffs=bq, flags=msdr.CHAIN_SYNCAM_PLL if pll_flag else 0)
    if anr_on.any():
        chain.set_anr(anr_on)
    # ragged calls; in every third case channel 0 is retuned to SYNCAM (or away from it) after the first call
    cuts, o = [], 0
    while o < n:
        m = int(min(n - o, rng.choice([1, 63, 64, 65, 128, 1000, 1024, 2500, n])))
        cuts.append((o, m)); o += m
    retune = (case % 3 == 0) and len(cuts) > 1
    mode0_after = orclib.AM if modes[0] == orclib.SYNCAM else orclib.SYNCAM
    got = np.empty((ch, n), np.float32)
    for j, (o, m) in enumerate(cuts):
        if retune and j == 1:
            chain.set_mode(0, mode0_after, 0)
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.float32)
        chain.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
    front = None
    if anr_on.any():            # the same stream through a chain without LMS filter and without cascade: the audio in front of the filter
        ind = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, hq, mixer=msdr.MIXER_NCO if nco else msdr.MIXER_FS4, modes=modes, osc_i=oi if nco else None,
                         osc_q=oq if nco else None, flags=msdr.CHAIN_SYNCAM_PLL if pll_flag else 0)
        front = np.empty((ch, n), np.float32)
        for j, (o, m) in enumerate(cuts):
            if retune and j == 1:
                ind.set_mode(0, mode0_after, 0)
            dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.float32)
            ind.process(dx, dy, m)
            front[:, o:o + m] = dy.download()
        ind.close()
    for c in rng.choice(ch, min(ch, 3), replace=False):
        if retune and c == 0:
            continue            # (a retuned PLL channel restarts its cascade, DESIGN 4.6: checked by the unit test, not here)
        mode = int(modes[c])
        pll = pll_flag and mode == orclib.SYNCAM
        lo = 0
        if pll:
            # The PLL starts on the filter's ramp-up, where I + jQ is a near-zero vector and its ANGLE (atan2, .ino:674) is decided by the
            # last bits: the oracle answers an absolute perturbation of 1e-8 of its I / Q with 0.4 of full scale during the first few
            # hundred samples at 252 taps (and with 4e-7 once it has locked).  PLL channels are compared once the loop has settled.
            lo = ntaps + 2500
            if n < lo + 500:
                continue
        if anr_on[c]:
            # (1) in front of the filter
            pre = orc.chain_f32(x[c], mode, lp, hq, oi, oq, None, pll=pll)
            e1 = err(front[c][lo:], pre[lo:])
            # (2) the filter and the cascade on the GPU's own audio
            w2 = orc.anr_f32(orc.anr_new(), int(anr_on[c]), front[c])
            if bq is not None:
                w2 = _cascade(w2, bq)
            e = max(e1, err(got[c][lo:], w2[lo:], front[c][lo:]))
            tol = 1e-5
            want = w2
        else:
            want = orc.chain_f32(x[c], mode, lp, hq, oi, oq, bq, pll=pll)
            e = err(got[c][lo:], want[lo:])
            tol = 1e-5
        if bq is not None and not anr_on[c]:
            p0 = orc.chain_f32(x[c], mode, lp, hq, oi, oq, None, pll=pll)
            tol *= max(1.0, float(np.sqrt((p0[lo:].astype(np.float64) ** 2).mean() / max((want[lo:].astype(np.float64) ** 2).mean(), 1e-300))))
        key = "lms" if anr_on[c] else "pll" if pll else "plain"
        worst[key] = max(worst[key], e)
        if not e < tol and key == "plain":
            # beyond the tolerance of the fp32 oracle: a defect only if the library is further from float64 than the oracle itself is
            t64 = truth64(x[c], mode, lp, hq, oi, oq, bq)
            if err(got[c], t64) <= 2 * err(want, t64) + 1e-6:
                continue
        if not e < tol:
            bad += 1
            print("MISMATCH", dict(seed=seed, case=case, ntaps=ntaps, ch=ch, n=n, channel=int(c), mode=mode, pll=pll, anr=int(anr_on[c]), stages=stages, nco=nco,
                                   err=e, kernel=chain.info()["kernel"]), flush=True)
            if ONLY:
                d = np.abs(got[c].astype(np.float64) - want)
                print("  P", P, "cuts", cuts[:12], "retune", retune, "first |diff| > 1e-4 at", int(np.argmax(d > 1e-4)), "max", d.max(), "level", np.abs(want).max())
                for lo in range(0, n, max(1, n // 16)):
                    print("   [%5d..] err %.2e" % (lo, err(got[c][lo:lo + n // 16], want[lo:lo + n // 16])))
                # the same rows through an independent chain of the two sideband flavours, same calls
                m2_ = [orclib.USB, orclib.LSB] if pll else [mode, mode]
                ind = msdr.Chain(ctx, msdr.ARITH_F32, 2, lp, hq, mixer=msdr.MIXER_NCO if nco else msdr.MIXER_FS4, modes=m2_,
                                 osc_i=oi if nco else None, osc_q=oq if nco else None)
                g2 = np.empty((2, n), np.float32)
                for (o2, m2) in cuts:
                    dx, dy = ctx.to_device(np.ascontiguousarray(np.stack([x[c, o2:o2 + m2]] * 2))), ctx.array((2, m2), np.float32)
                    ind.process(dx, dy, m2)
                    g2[:, o2:o2 + m2] = dy.download()
                print("  independent chain (the auxiliary chain's rows) vs oracle:", err(g2[0], orc.chain_f32(x[c], m2_[0], lp, hq, oi, oq, None)),
                      err(g2[1], orc.chain_f32(x[c], m2_[1], lp, hq, oi, oq, None)), ind.info()["kernel"])
                if anr_on[c] and not pll:
                    # the oracle's own LMS filter on the GPU's demodulated audio: what is left is the GPU filter's arithmetic
                    a2 = orc.anr_f32(orc.anr_new(), int(anr_on[c]), g2[0])
                    print("  oracle LMS on the GPU's FIR output vs GPU chain (no cascade):", err(got[c], a2, g2[0]) if bq is None else "n/a (cascade behind)")
            break
    chain.close()
    if ONLY:
        break
    if case % 50 == 0:
        print("cases", case, "bad", bad, flush=True)
print("fuzz_pll_anr_f32 done: %d cases, %d mismatches (seed %d); worst error plain %.2g, PLL %.2g, LMS %.2g" % (case, bad, seed, worst["plain"], worst["pll"], worst["lms"]))
sys.exit(1 if bad else 0)
