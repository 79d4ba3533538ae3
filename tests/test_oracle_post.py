"""The oracle's fp32 flavours of rows f2 / f3 (orc_syncam_step_f32, orc_anr_step_f32: the PLL / LMS statements on float samples, used
inside the fp32 chain) against its q15 functions: on integer-valued input the float PLL's corr[0], truncated, is the q15 output sample
for sample, and the float LMS filter's output x 32768, truncated, is the q15 filter's -- neither loop feeds its OUTPUT back, so the
truncation of the q15 store cannot change the state."""
import numpy as np

import orclib


def test_float_pll_and_lms_agree_with_the_q15_functions_on_integer_input():
    o = orclib.Oracle()
    rng = np.random.default_rng(3)
    n = 8192
    t = np.arange(n)
    env = 1 + 0.5 * np.sin(2 * np.pi * 0.002 * t)
    i16 = np.round(8000 * np.cos(2 * np.pi * 0.01 * t + 0.3) * env + rng.normal(0, 40, n)).astype(np.int16)
    q16 = np.round(8000 * np.sin(2 * np.pi * 0.01 * t + 0.3) * env + rng.normal(0, 40, n)).astype(np.int16)
    a = o.syncam_q15(o.syncam_new(), i16, q16)
    b = o.syncam_f32(o.syncam_new(), i16.astype(np.float32), q16.astype(np.float32))
    assert np.array_equal(a, np.trunc(b).astype(np.int64).astype(np.int16))
    x = rng.integers(-8000, 8001, 2048).astype(np.int16)
    for on in (1, 2):
        qa = o.anr_q15(o.anr_new(), on, x)
        fa = o.anr_f32(o.anr_new(), on, x.astype(np.float32) / 32768)
        assert np.array_equal(qa, np.trunc(fa.astype(np.float64) * 32768).astype(np.int64).astype(np.int16)), on
    # anr_on = 0 leaves data and state alone
    st = o.anr_new()
    assert np.array_equal(o.anr_f32(st, 0, x.astype(np.float32)), x.astype(np.float32)) and st.in_idx == 0


def test_chain_f32_post_run_is_the_plain_chain_when_nothing_is_switched_on():
    o = orclib.Oracle()
    rng = np.random.default_rng(4)
    x = rng.integers(-8000, 8001, 3000).astype(np.int16)
    k = np.arange(61) - 30
    lp = (np.sinc(0.2 * k) * np.kaiser(61, 6.0)).astype(np.float32)
    cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
    plain = o.chain_f32(x, orclib.SYNCAM, lp, lp, sin4, cos4)
    am = o.chain_f32(x, orclib.AM, lp, lp, sin4, cos4)
    assert np.array_equal(plain, am)                                   # without the PLL flag SYNCAM is the AM branch (Teensy 3.2 build)
    pll = o.chain_f32(x, orclib.SYNCAM, lp, lp, sin4, cos4, pll=True)
    assert not np.array_equal(pll, am)
    assert np.array_equal(o.chain_f32(x, orclib.LSB, lp, lp, sin4, cos4, pll=True), o.chain_f32(x, orclib.LSB, lp, lp, sin4, cos4))


def test_lms_filter_is_discontinuous_in_its_input():
    """Why the GPU test of LMS channels cannot ask for 1e-5: the leak control's per-sample decision (nev < nel, .ino:754-757) makes the
    filter's output jump when its input moves in the last bit -- measured here on the oracle against itself, on the PLL-demodulated
    audio of the GPU test's channel 3."""
    o = orclib.Oracle()
    rng = np.random.default_rng(7 + 256)
    n, c = 6000, 3
    t = np.arange(n)
    k = np.arange(256) - 255 / 2
    lp = (np.sinc(2 * 2800 / 24000 * k) * np.kaiser(256, 7.0)).astype(np.float32)
    lp = (lp / lp.sum()).astype(np.float32)
    for cc in range(c + 1):          # (the generator draws the channels in order)
        f = 0.25 + (35.0 + 17.0 * cc) / 24000.0
        env = 1.0 + 0.3 * np.sin(2 * np.pi * (400.0 + 50 * cc) / 24000.0 * t)
        x = np.round(9000 * env * np.cos(2 * np.pi * f * t + 0.4 * cc) + rng.normal(0, 60, n)).astype(np.int16)
    cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
    pre = o.chain_f32(x, orclib.SYNCAM, lp, lp, sin4, cos4, None, pll=True)
    worst = 0.0
    for on in (1, 2):
        a = o.anr_f32(o.anr_new(), on, pre)
        b = o.anr_f32(o.anr_new(), on, (pre.astype(np.float64) * (1 + 1e-7 * rng.standard_normal(n))).astype(np.float32))
        worst = max(worst, float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum() / (pre.astype(np.float64) ** 2).sum())))
    assert 2e-6 < worst < 2e-4, worst
