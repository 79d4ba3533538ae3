"""Rows f2 / f3 (SYNCAM PLL .ino:631-688, LMS notch / noise reduction .ino:702-770) INSIDE the fp32 chain -- an extension: the reference
runs both on int16 samples, the Q15 chain mirrors that bit-exactly (test_gpu_frontend.py); here the same statements run on the fp32
FIR outputs.  Oracle: orc_chain_f32_post_run (oracle/msdr_oracle.c), whose float PLL / LMS steps agree with the q15 functions on
integer-valued input (tests/test_oracle_post.py)."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr, rel_rms  # noqa: F401

pytestmark = pytest.mark.gpu
TOL = 1e-5          # as everywhere in the fp32 chain
# LMS channels are checked in two parts.  The filter's leak control (.ino:754-757) takes a DECISION per sample (nev < nel), so its output
# is discontinuous in its input: the oracle answers a 1e-7 relative perturbation of its own input with 1e-5 .. 5e-5 of the input level
# (tests/test_oracle_post.py), and a 3e-7 difference in front of the filter has been seen to come out as 4e-3
# (tests/debug/fuzz_pll_anr_f32.py).  So: (1) the audio IN FRONT of the filter -- a second chain with the filter and the cascade off --
# against the oracle, 1e-5; (2) the chain's output against the oracle's filter + cascade applied to THAT audio: the filter's arithmetic
# alone (bit-identical without a cascade), 1e-5 of the level in front of the filter.


def _err(got, want, pre=None):
    ref = np.sqrt((want.astype(np.float64) ** 2).sum())
    if pre is not None:
        ref = max(ref, np.sqrt((pre.astype(np.float64) ** 2).sum()))
    return float(np.sqrt(((got.astype(np.float64) - want) ** 2).sum()) / max(ref, 1e-300))


def _cascade(orc, v, bq):
    return orc.biquad_df1_zero_state(bq, v) if bq is not None else v


def _lowpass(ntaps):
    k = np.arange(ntaps) - (ntaps - 1) / 2
    lp = (np.sinc(2 * 2800 / 24000 * k) * np.kaiser(ntaps, 7.0)).astype(np.float32)
    return (lp / lp.sum()).astype(np.float32)


def _am_if(rng, ch, n, off_hz):
    """AM carrier near fs/4 (the PLL has something to lock to), 30 % modulation, a little noise"""
    t = np.arange(n)
    x = np.empty((ch, n), np.int16)
    for c in range(ch):
        f = 0.25 + (off_hz + 17.0 * c) / 24000.0
        env = 1.0 + 0.3 * np.sin(2 * np.pi * (400.0 + 50 * c) / 24000.0 * t)
        x[c] = np.round(9000 * env * np.cos(2 * np.pi * f * t + 0.4 * c) + rng.normal(0, 60, n)).astype(np.int16)
    return x


def _bq(orc, stages):
    out = []
    for k, q in enumerate([0.54, 15.0][:stages]):
        kind = orclib.BQ_NOTCH if k == 1 else orclib.BQ_LOWPASS
        f = (3000 if k == 1 else 5400) * orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
        c = orc.biquad_design(kind, np.float32(f), q).astype(np.float64) / 2 ** 30
        out.append([c[0], c[1], c[2], -c[3], -c[4]])
    return np.array(out, np.float32) if out else None


def _run(ctx, chain, x, splits):
    ch, n = x.shape
    got = np.empty((ch, n), np.float32)
    o = 0
    for m in splits:
        m = min(m, n - o)
        if m <= 0:
            break
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.float32)
        chain.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
        o += m
    assert o == n
    return got


@pytest.mark.parametrize("stages", [0, 2])
@pytest.mark.parametrize("ntaps", [62, 256])
def test_chain_f32_syncam_pll_and_lms_filter(ctx, orc, ntaps, stages):
    rng = np.random.default_rng(7 + ntaps + stages)
    lp = _lowpass(ntaps)
    modes = [orclib.SYNCAM, orclib.AM, orclib.LSB, orclib.SYNCAM, orclib.USB, orclib.SYNCAM]
    anr = [0, 1, 2, 2, 0, 1]
    n = 6000
    x = _am_if(rng, len(modes), n, 35.0)
    bq = _bq(orc, stages)
    cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, len(modes), lp, lp, mixer=msdr.MIXER_FS4, modes=modes, biquad_coeffs=bq, flags=msdr.CHAIN_SYNCAM_PLL)
    chain.set_anr(anr)
    splits = [2048, 1000, 129, 64, 10000]
    got = _run(ctx, chain, x, splits)
    front_chain = msdr.Chain(ctx, msdr.ARITH_F32, len(modes), lp, lp, mixer=msdr.MIXER_FS4, modes=modes, flags=msdr.CHAIN_SYNCAM_PLL)
    front = _run(ctx, front_chain, x, splits)                      # the audio in front of LMS filter and cascade
    for c in range(len(modes)):
        pll = modes[c] == orclib.SYNCAM
        if anr[c]:
            assert _err(front[c], orc.chain_f32(x[c], modes[c], lp, lp, sin4, cos4, None, pll=pll)) < TOL, (c, "front")
            want = _cascade(orc, orc.anr_f32(orc.anr_new(), anr[c], front[c]), bq)
            if bq is None:
                assert np.array_equal(got[c], want), (c, modes[c], anr[c])          # the filter's arithmetic is the oracle's, bit for bit
            assert _err(got[c], want, front[c]) < TOL, (c, modes[c], anr[c], _err(got[c], want, front[c]))
        else:
            st = {}
            want = np.concatenate([orc.chain_f32(x[c, o:o + m], modes[c], lp, lp, sin4, cos4, bq, state=st, pll=pll) for o, m in _segments(n, splits)])
            assert rel_rms(got[c], want) < TOL, (c, modes[c], rel_rms(got[c], want))
    # the PLL really demodulates differently from the envelope (so the test would notice a channel left on the AM branch)
    env = orc.chain_f32(x[0], orclib.AM, lp, lp, sin4, cos4, bq)
    assert rel_rms(got[0], env) > 1e-2


def _segments(n, splits):
    o = 0
    for m in splits:
        m = min(m, n - o)
        if m <= 0:
            break
        yield o, m
        o += m


def test_chain_f32_post_channels_follow_retune_reset_and_switch_off(ctx, orc):
    """msdr_chain_set_mode turns a channel into a PLL channel mid-stream (FIR history and table position are taken over: no filter
    transient; PLL state starts from zero as it would on the Teensy after a mode change), msdr_chain_set_anr(None, 0) removes the
    auxiliary pass again, msdr_chain_reset restarts everything."""
    rng = np.random.default_rng(99)
    lp = _lowpass(100)
    cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
    n = 3072
    x = _am_if(rng, 2, 3 * n, 20.0)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 2, lp, lp, mixer=msdr.MIXER_FS4, modes=[orclib.AM, orclib.AM], flags=msdr.CHAIN_SYNCAM_PLL)
    g0 = _run(ctx, chain, x[:, :n], [n])
    chain.set_mode(1, orclib.SYNCAM, 0)
    g1 = _run(ctx, chain, x[:, n:2 * n], [n])
    st = {}
    w0 = orc.chain_f32(x[1, :n], orclib.AM, lp, lp, sin4, cos4, None, state=st)
    w1 = orc.chain_f32(x[1, n:2 * n], orclib.SYNCAM, lp, lp, sin4, cos4, None, state=st, pll=True)      # FIR history continues, PLL from zero
    assert rel_rms(g0[1], w0) < 1e-5 and rel_rms(g1[1], w1) < TOL, (rel_rms(g0[1], w0), rel_rms(g1[1], w1))
    assert rel_rms(g1[1][:64], w1[:64]) < TOL                       # right at the switch
    st0 = {}
    orc.chain_f32(x[0, :n], orclib.AM, lp, lp, sin4, cos4, None, state=st0)
    assert rel_rms(g1[0], orc.chain_f32(x[0, n:2 * n], orclib.AM, lp, lp, sin4, cos4, None, state=st0)) < 1e-5   # the neighbour is untouched
    chain.reset()
    g2 = _run(ctx, chain, x[:, :n], [n])
    assert rel_rms(g2[1], orc.chain_f32(x[1, :n], orclib.SYNCAM, lp, lp, sin4, cos4, None, pll=True)) < TOL
    chain.set_anr([2, 0])
    chain.reset()
    g3 = _run(ctx, chain, x[:, :n], [n])
    assert _err(g3[0], orc.anr_f32(orc.anr_new(), 2, g0[0]), g0[0]) < TOL        # g0[0]: the same channel's audio without the filter (first run)
    chain.set_anr(None, 0)
    chain.set_mode(1, orclib.AM, 0)
    chain.reset()
    g4 = _run(ctx, chain, x[:, :n], [n])
    assert chain.info()["kernel"].startswith("chain_")
    for c in range(2):
        assert rel_rms(g4[c], orc.chain_f32(x[c, :n], orclib.AM, lp, lp, sin4, cos4, None)) < 1e-5
