"""MSDR_CHAIN_OUT_I16: the fp32 chain writing the play queue's sample type (int16, src/Audio/play_queue.h:41) as arm_float_to_q15
converts (prototype arm_math.h:6592; CMSIS-DSP 1.5.x: (q15_t) __SSAT((q31_t)(x * 32768.0f), 16)) -- 4 B per sample through HBM instead of 6.

Two checks per configuration: (1) against the SAME chain without the flag, converted on the host: identical, sample for sample (the
conversion is the only difference); (2) against the oracle's fp32 audio converted the same way: never more than 1 LSB apart, and apart
at all only where the fp32 values straddle an integer (the 5e-7 agreement of the fp32 chains times a few thousand LSB of level)."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr, rel_rms  # noqa: F401
from test_gpu_chain import _f32_biquads, _hilbert_pair, _q15_nco, run_chain

pytestmark = pytest.mark.gpu
COS4, SIN4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)


def to_q15(y):
    """arm_float_to_q15 without ARM_MATH_ROUNDING: the fp32 product truncated toward zero, saturated"""
    v = (np.asarray(y, np.float32) * np.float32(32768.0)).astype(np.float64)
    return np.clip(np.trunc(v), -32768, 32767).astype(np.int16)


def lowpass(n, bw=2800.0):
    k = np.arange(n) - (n - 1) / 2.0
    h = np.sinc(2 * bw / 24000.0 * k) * np.kaiser(n, 7.0)
    return (h / h.sum()).astype(np.float32)


CASES = {
    "am256_2sec_mfw": dict(taps=256, mode=orclib.AM, stages=2, kernel="chain_mfw_kernel<2>"),
    "am256_1sec_amtr": dict(taps=256, mode=orclib.AM, stages=1, kernel="chain_amtr_kernel"),
    "lsb100_2sec": dict(taps=100, mode=orclib.LSB, stages=2, kernel="chain_mfw_kernel<2>"),
    "usb100_4sec": dict(taps=100, mode=orclib.USB, stages=4, kernel="chain_mfw_kernel<4>"),
    "valu_fold": dict(taps=100, mode=orclib.LSB, stages=2, flags=msdr.CHAIN_NO_MFMA, kernel="chain_fold_kernel<4>"),
    "general_table_long_fir": dict(taps=460, mode=orclib.USB, stages=1, nco=(128, 5), kernel="chain_mfw_kernel<1> full-rate NCO streams"),   # (round 5: the compact B layout keeps it on the matrix cores)
    "table_of_another_period": dict(taps=260, mode=orclib.USB, stages=1, nco=(96, 5, 96), kernel="chain_kernel<ArithF32>"),     # a 96-entry table: 96 does not divide 128, no block-periodic streams -- the as-written kernel answers
    "general_table_260_taps": dict(taps=260, mode=orclib.USB, stages=1, nco=(128, 5), kernel="chain_mfw_kernel<1> full-rate NCO streams"),
    "full_rate_table": dict(taps=100, mode=orclib.LSB, stages=2, nco=(128, 5), kernel="chain_mfw_kernel<2> full-rate NCO streams"),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("block", [None, 1000, 333])
def test_int16_audio_out(ctx, orc, name, block):
    cs = CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % 1000 + (block or 0))
    ch, n = 5, 4000
    if cs["mode"] == orclib.AM:
        hi = hq = lowpass(cs["taps"])
    else:
        hi, hq = _hilbert_pair(cs["taps"])
    bq = _f32_biquads(orc, cs["stages"]) if cs["stages"] else None
    kw = dict(mixer=msdr.MIXER_FS4, mode=cs["mode"], biquad_coeffs=bq)
    oi, oq = SIN4, COS4
    if "nco" in cs:
        oi, oq = _q15_nco(*cs["nco"])
        kw.update(mixer=msdr.MIXER_NCO, osc_i=oi, osc_q=oq)
    x = rng.integers(-20000, 20001, (ch, n)).astype(np.int16)
    x[1] = (x[1].astype(np.int32) * 3 // 2).clip(-32768, 32767).astype(np.int16)       # loud enough to saturate the int16 output now and then
    plain = msdr.Chain(ctx, msdr.ARITH_F32, ch, hi, hq, flags=cs.get("flags", 0), **kw)
    i16 = msdr.Chain(ctx, msdr.ARITH_F32, ch, hi, hq, flags=cs.get("flags", 0) | msdr.CHAIN_OUT_I16, **kw)
    want_f = run_chain(ctx, plain, x, np.float32, block=block)
    got = run_chain(ctx, i16, x, np.int16, block=block)
    assert plain.info()["kernel"] == cs["kernel"] and i16.info()["kernel"] == cs["kernel"], (plain.info()["kernel"], i16.info()["kernel"])
    assert np.array_equal(got, to_q15(want_f)), (name, block, int((got != to_q15(want_f)).sum()))
    for c in range(ch):
        ref = to_q15(orc.chain_f32(x[c], cs["mode"], hi, hq, oi, oq, bq))
        diff = np.abs(got[c].astype(np.int32) - ref.astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).mean() < 0.01, (name, c, int(diff.max()), float((diff != 0).mean()))


def test_int16_audio_out_behind_the_post_passes(ctx, orc):
    """The CMSIS-order cascade and the PLL channels run behind the main kernel on fp32 audio: the conversion comes last."""
    from test_gpu_chain_post import _am_if
    rng = np.random.default_rng(3)
    hi, hq = _hilbert_pair(100)
    c30 = lambda kind, f, q: (lambda c: [c[0], c[1], c[2], -c[3], -c[4]])(orc.biquad_design(kind, np.float32(f * orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0), q).astype(np.float64) / 2 ** 30)
    bad = np.array([c30(orclib.BQ_HIGHPASS, 300.0, 0.7), c30(orclib.BQ_HIGHPASS, 300.0, 0.7)], np.float32)
    x = rng.integers(-12000, 12001, (3, 3000)).astype(np.int16)
    kw = dict(mixer=msdr.MIXER_FS4, mode=orclib.LSB, biquad_coeffs=bad)
    a = run_chain(ctx, msdr.Chain(ctx, msdr.ARITH_F32, 3, hi, hq, **kw), x, np.float32)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 3, hi, hq, flags=msdr.CHAIN_OUT_I16, **kw)
    b = run_chain(ctx, chain, x, np.int16, block=1000)
    assert "biquad_df1_seq_kernel" in chain.info()["kernel"]
    d = np.abs(b.astype(np.int32) - to_q15(a).astype(np.int32))          # (one call against three: the seq kernel's state carries, values agree to rounding)
    assert d.max() <= 1
    lp = lowpass(61)
    modes = np.array([orclib.SYNCAM, orclib.AM], np.int32)
    xs = _am_if(rng, 2, 3072, 35.0)
    kw = dict(mixer=msdr.MIXER_FS4, modes=modes, biquad_coeffs=_f32_biquads(orc, 2))
    a = run_chain(ctx, msdr.Chain(ctx, msdr.ARITH_F32, 2, lp, lp, flags=msdr.CHAIN_SYNCAM_PLL, **kw), xs, np.float32)
    b = run_chain(ctx, msdr.Chain(ctx, msdr.ARITH_F32, 2, lp, lp, flags=msdr.CHAIN_SYNCAM_PLL | msdr.CHAIN_OUT_I16, **kw), xs, np.int16)
    assert np.array_equal(b, to_q15(a))
