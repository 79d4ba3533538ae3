"""GPU parity of the kernel-function API (CMSIS / Teensy-Audio mirrors) against the oracle and the
reference-generated golden vectors.  Everything goes through the C ABI (include/msdr.h).
Integer paths: bit-exact.  fp32 paths: relative RMS <= 1e-5 per channel (north-star tolerance);
most are far tighter and the tighter bound is what is asserted."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr, rel_rms  # noqa: F401

pytestmark = pytest.mark.gpu
B = 128


def _rand16(rng, shape, amp=32767):
    return rng.integers(-amp - 1 if amp == 32767 else -amp, amp + 1, shape).astype(np.int16)


# ---------------------------------------------------------------- A1 / A2 / A5 -------------
def test_mix_fs4_q15(ctx, orc):
    rng = np.random.default_rng(1)
    x = _rand16(rng, (5, 4 * B))
    x[0, :8] = -32768
    dx = ctx.to_device(x)
    di, dq = ctx.array(x.shape, np.int16), ctx.array(x.shape, np.int16)
    ctx.mix_fs4_q15(dx, di, dq, x.shape[0], x.shape[1])
    gi, gq = di.download(), dq.download()
    for c in range(x.shape[0]):
        wi, wq = orc.mix_fs4(x[c])
        assert np.array_equal(gi[c], wi) and np.array_equal(gq[c], wq)


@pytest.mark.parametrize("direction", [0, 1])
@pytest.mark.parametrize("passthrough", [0, 1])
def test_freqconv(ctx, orc, direction, passthrough):
    rng = np.random.default_rng(2 + direction)
    i, q = _rand16(rng, (3, B)), _rand16(rng, (3, B))
    n = np.arange(B)
    oi = np.round(32767 * np.sin(2 * np.pi * 5 * n / B)).astype(np.int16)
    oq = np.round(32767 * np.cos(2 * np.pi * 5 * n / B)).astype(np.int16)
    oi[3], oq[3], i[:, 3], q[:, 3] = -32768, -32768, -32768, -32768
    di, dq = ctx.to_device(i), ctx.to_device(q)
    ctx.freqconv_q15(di, dq, oi, oq, direction, passthrough, 3, B)
    gi, gq = di.download(), dq.download()
    for c in range(3):
        wi, wq = orc.freqconv_q15(i[c], q[c], oi, oq, direction, passthrough)
        assert np.array_equal(gi[c], wi) and np.array_equal(gq[c], wq)
    fi, fq = (i / 32768).astype(np.float32), (q / 32768).astype(np.float32)
    foi, foq = (oi / 32768).astype(np.float32), (oq / 32768).astype(np.float32)
    di, dq = ctx.to_device(fi), ctx.to_device(fq)
    ctx.freqconv_f32(di, dq, foi, foq, direction, passthrough, 3, B)
    gi, gq = di.download(), dq.download()
    for c in range(3):
        wi, wq = orc.freqconv_f32(fi[c], fq[c], foi, foq, direction, passthrough)
        assert np.allclose(gi[c], wi, rtol=0, atol=2e-7) and np.allclose(gq[c], wq, rtol=0, atol=2e-7)


def test_demod_q15_all_modes(ctx, orc):
    rng = np.random.default_rng(3)
    i, q = _rand16(rng, (5, 2 * B)), _rand16(rng, (5, 2 * B))
    i[:, :4], q[:, :4] = -32768, -32768          # int32 overflow of I*I+Q*Q
    i[:, 4:8], q[:, 4:8] = 32767, 32767          # sqrt > 32767: truncating store wraps
    modes = np.array([orclib.AM, orclib.LSB, orclib.USB, orclib.CW, orclib.SYNCAM], np.int32)
    di, dq, do = ctx.to_device(i), ctx.to_device(q), ctx.array(i.shape, np.int16)
    dm = ctx.to_device(modes)
    for kind in (orclib.SQRT_F32, orclib.SQRT_Q31):
        ctx.demod_q15(orclib.AM, di, dq, do, 5, 2 * B, sqrt_kind=kind, d_mode=dm)
        got = do.download()
        for c in range(5):
            assert np.array_equal(got[c], orc.demod_q15(modes[c], i[c], q[c], kind)), (kind, c)
    ctx.demod_q15(orclib.USB, di, dq, do, 5, 2 * B)
    assert np.array_equal(do.download()[1], orc.demod_q15(orclib.USB, i[1], q[1]))


def test_demod_q15_sqrt_exhaustive_boundaries(ctx, orc):
    """trunc(sqrtf(float(s))) must agree with the CPU for sums right below/at perfect squares."""
    k = np.arange(1, 32767, dtype=np.int64)
    i = np.concatenate([k, k, k]).astype(np.int16)
    q = np.zeros_like(i)
    # s = k*k exactly; plus pairs whose sum is k*k - 1 or k*k + 1 are covered by random draws below
    rng = np.random.default_rng(4)
    i2, q2 = _rand16(rng, i.size, 23000), _rand16(rng, i.size, 23000)
    I, Q = np.concatenate([i, i2]), np.concatenate([q, q2])
    di, dq, do = ctx.to_device(I[None]), ctx.to_device(Q[None]), ctx.array((1, I.size), np.int16)
    ctx.demod_q15(orclib.AM, di, dq, do, 1, I.size)
    assert np.array_equal(do.download()[0], orc.demod_q15(orclib.AM, I, Q))


def test_demod_f32(ctx, orc):
    rng = np.random.default_rng(5)
    i, q = rng.standard_normal((4, B)).astype(np.float32), rng.standard_normal((4, B)).astype(np.float32)
    modes = np.array([orclib.AM, orclib.LSB, orclib.USB, orclib.CW], np.int32)
    di, dq, do, dm = ctx.to_device(i), ctx.to_device(q), ctx.array(i.shape, np.float32), ctx.to_device(modes)
    ctx.demod_f32(orclib.AM, di, dq, do, 4, B, d_mode=dm)
    got = do.download()
    for c in range(4):
        assert np.allclose(got[c], orc.demod_f32(modes[c], i[c], q[c]), rtol=3e-7, atol=0)


# ---------------------------------------------------------------- A3 / A4 q15 FIR ----------
@pytest.mark.parametrize("tn", ["ssb_i", "am102", "lp256", "lp512", "lp62", "wrap8", "n4", "n6"])
def test_fir_q15_vs_reference_golden(ctx, golden, tn):
    taps = golden["fir/taps_" + tn]
    fir = msdr.FirQ15(ctx, taps, 2)
    for sn in ("noise", "full"):
        x = golden["fir/x_" + sn]
        xx = np.stack([x, x[::-1].copy()])
        for blk in (128, 130, 7):
            fir.reset()
            got = np.empty_like(xx)
            for o in range(0, x.size, blk):
                n = min(blk, x.size - o)
                dx, dy = ctx.to_device(xx[:, o:o + n]), ctx.array((2, n), np.int16)
                fir.process(dx, dy, n)
                got[:, o:o + n] = dy.download()
            assert np.array_equal(got[0], golden["fir/%s_%s_b%d" % (tn, sn, blk)]), (tn, sn, blk)
    fir.close()


def test_fir_q15_rejects_odd_taps_and_inplace(ctx):
    with pytest.raises(msdr.MsdrError) as e:
        msdr.FirQ15(ctx, np.ones(5, np.int16), 1)
    assert e.value.status == msdr.STATUS_ARGUMENT_ERROR        # arm_fir_init_q15.c:93-96
    fir = msdr.FirQ15(ctx, np.ones(4, np.int16), 1)
    d = ctx.to_device(np.zeros((1, B), np.int16))
    with pytest.raises(msdr.MsdrError):
        fir.process(d, d, B)


def test_fir_q15_long_block_time_segments(ctx, orc):
    """One long block is split into time segments on the GPU; the integer result is still exact."""
    rng = np.random.default_rng(6)
    taps = _rand16(rng, 102, 3000)
    x = _rand16(rng, (3, 100003))
    fir = msdr.FirQ15(ctx, taps, 3)
    dx, dy = ctx.to_device(x), ctx.array(x.shape, np.int16)
    fir.process(dx, dy, x.shape[1])
    got = dy.download()
    for c in range(3):
        _, want = orc.fir_q15_blocks(taps, x[c], x.shape[1])
        assert np.array_equal(got[c], want)


# ---------------------------------------------------------------- A6 fp32 FIR --------------
@pytest.mark.parametrize("ntaps", [1, 3, 61, 100, 256, 512])
def test_fir_f32_vs_oracle(ctx, orc, ntaps):
    rng = np.random.default_rng(ntaps)
    h = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    x = rng.uniform(-1, 1, (3, 7001)).astype(np.float32)
    fir = msdr.FirF32(ctx, h, 3)
    got = np.empty_like(x)
    for o, n in ((0, 4000), (4000, 3001)):                       # state carried between two calls
        dx, dy = ctx.to_device(x[:, o:o + n]), ctx.array((3, n), np.float32)
        fir.process(dx, dy, n)
        got[:, o:o + n] = dy.download()
    for c in range(3):
        assert rel_rms(got[c], orc.fir_f32_blocks(h, x[c], 128)) < 1e-6


# ---------------------------------------------------------------- A8 fp32 biquad -----------
@pytest.mark.parametrize("stages", [0, 1, 2, 4])
def test_biquad_df1_f32_vs_oracle(ctx, orc, stages):
    rng = np.random.default_rng(10 + stages)
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    coeffs = []
    for k, q in enumerate([0.54, 15.0, 0.54, 1.3][:stages]):
        kind = orclib.BQ_NOTCH if k == 1 else orclib.BQ_LOWPASS
        c = orc.biquad_design(kind, np.float32((3000 if k == 1 else 5400) * corr), q).astype(np.float64) / 2 ** 30
        coeffs.append([c[0], c[1], c[2], -c[3], -c[4]])
    coeffs = np.array(coeffs, np.float32).reshape(-1, 5)
    x = rng.uniform(-1, 1, (3, 9000)).astype(np.float32)
    bq = msdr.BiquadDf1F32(ctx, coeffs, 3)
    got = np.empty_like(x)
    for o, n in ((0, 128), (128, 5000), (5128, 3872)):
        dx, dy = ctx.to_device(x[:, o:o + n]), ctx.array((3, n), np.float32)
        bq.process(dx, dy, n)
        got[:, o:o + n] = dy.download()
    for c in range(3):
        assert rel_rms(got[c], orc.biquad_df1_blocks(coeffs, x[c], 128)) < 2e-6


def test_biquad_df1_f32_long_single_stream_is_segmented(ctx, orc):
    """One channel x 2^21 samples: the stage cuts the block into time segments that re-converge over a warm-up (poles inside
    the unit circle); two calls (state carried through the ping-pong record), and the in-place form (never segmented)."""
    rng = np.random.default_rng(3)
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    coeffs = []
    for kind, f, q in ((orclib.BQ_LOWPASS, 5400 * corr, 0.54), (orclib.BQ_NOTCH, 3000 * corr, 15.0)):
        c = orc.biquad_design(kind, np.float32(f), q).astype(np.float64) / 2 ** 30
        coeffs.append([c[0], c[1], c[2], -c[3], -c[4]])
    coeffs = np.array(coeffs, np.float32)
    n = 1 << 21
    x = rng.uniform(-1, 1, (1, n)).astype(np.float32)
    want = orc.biquad_df1_blocks(coeffs, x[0], 128)
    bq = msdr.BiquadDf1F32(ctx, coeffs, 1)
    got = np.empty_like(x)
    for o, m in ((0, n // 2 + 128), (n // 2 + 128, n // 2 - 128)):
        dx, dy = ctx.to_device(x[:, o:o + m]), ctx.array((1, m), np.float32)
        bq.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
    assert rel_rms(got[0], want) < 2e-6
    for lo in range(0, n, n // 64):                               # and locally, around would-be segment boundaries
        assert rel_rms(got[0, lo:lo + 4096], want[lo:lo + 4096]) < 5e-6, lo
    bq2 = msdr.BiquadDf1F32(ctx, coeffs, 1)
    d = ctx.to_device(x)
    bq2.process(d, d, n)                                          # in place
    assert rel_rms(d.download()[0], want) < 2e-6


# ---------------------------------------------------------------- A7 Teensy biquad ---------
@pytest.mark.parametrize("n_stage", [1, 2, 4])
def test_biquad_q15_bit_exact(ctx, orc, n_stage):
    rng = np.random.default_rng(20 + n_stage)
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    coefs = [orc.biquad_design(orclib.BQ_LOWPASS, np.float32(5400 * corr), q) for q in [0.54, 1.3, 0.54, 1.3][:n_stage]]
    ch = 70                                                       # more than one 64-channel workgroup
    node = msdr.BiquadQ15(ctx, ch)
    for s, c in enumerate(coefs):
        node.set_coefficients(s, c)
    node.set_coefficients(4, coefs[0])                            # ignored (filter_biquad.cpp:86)
    refs = [orc.biquad_teensy_new(coefs) for _ in range(ch)]
    assert list(node.definition(0)) == list(refs[0].definition)
    for blk in range(5):
        amp = 32767 if blk in (2, 3) else 4000
        x = _rand16(rng, (ch, 2 * B), amp)
        if blk == 2:
            x[:, :] = 32767                                       # drive into saturation
        d = ctx.to_device(x)
        node.update(d, 2 * B)
        got = d.download()
        for c in range(ch):
            assert np.array_equal(got[c], orc.biquad_teensy_update(refs[c], x[c])), (blk, c)
    assert list(node.definition(ch - 1)) == list(refs[ch - 1].definition)
    # re-tuning keeps the history words, clears the residue (filter_biquad.cpp:95-98)
    node.set_coefficients(0, coefs[-1])
    orc.lib.orc_biquad_teensy_set_coefficients(orclib.C.byref(refs[3]), 0, coefs[-1].ctypes.data_as(orclib._p))
    assert list(node.definition(3)) == list(refs[3].definition)
    d = ctx.to_device(np.zeros((ch, 7), np.int16))
    with pytest.raises(msdr.MsdrError) as e:
        node.update(d, 7)
    assert e.value.status == msdr.STATUS_LENGTH_ERROR


@pytest.mark.parametrize("ch", [16, 48, 80, 96])
def test_biquad_q15_slab_pipeline_on_channel_counts_that_are_not_whole_waves(ctx, orc, ch):
    """Below 8192 channels the slab pipeline takes 16 channels per workgroup: any multiple of 16 channels runs on it (every channel checked)."""
    rng = np.random.default_rng(ch)
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    coef = orc.biquad_design(orclib.BQ_NOTCH, np.float32(3000 * corr), 15.0)
    node = msdr.BiquadQ15(ctx, ch)
    node.set_coefficients(0, coef)
    refs = [orc.biquad_teensy_new([coef]) for _ in range(ch)]
    for nb in (2, 1, 4):
        x = _rand16(rng, (ch, nb * B), 20000)
        d = ctx.to_device(x)
        node.update(d, nb * B)
        got = d.download()
        for c in range(ch):
            assert np.array_equal(got[c], orc.biquad_teensy_update(refs[c], x[c])), (nb, c)
    for c in range(ch):
        assert list(node.definition(c)) == list(refs[c].definition), c


@pytest.mark.parametrize("per_group", [16, 32, 64])
@pytest.mark.parametrize("kind", ["lowpass", "notch"])
def test_biquad_q15_one_stage_node_on_the_slab_pipeline(ctx, orc, kind, per_group, monkeypatch):
    """A one-stage node over a slab-shaped batch (channels a multiple of 64, blocks a multiple of 128: the reference's cadence) runs on
    biquad_teensy_pipe4_kernel<1> -- the recursion alone on one wave, the input products element-wise beside it.  Bit-exact incl. the
    state record, saturation, several calls; a second stage sends the node back to the per-lane kernel."""
    monkeypatch.setenv("MSDR_BIQUAD_PIPE_CH", str(per_group))      # 16 / 32 / 64 channels per workgroup: the host's rule picks by batch size (read at create time)
    rng = np.random.default_rng(60 + len(kind))
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    coef = orc.biquad_design(orclib.BQ_LOWPASS, np.float32(5400 * corr), 0.54) if kind == "lowpass" else orc.biquad_design(orclib.BQ_NOTCH, np.float32(3000 * corr), 15.0)
    ch = 128
    node = msdr.BiquadQ15(ctx, ch)
    node.set_coefficients(0, coef)
    refs = [orc.biquad_teensy_new([coef]) for _ in range(ch)]
    for blk, nb in enumerate([1, 3, 2, 5]):
        n = nb * B
        x = _rand16(rng, (ch, n), 32767 if blk == 2 else 6000)
        if blk == 2:
            x[:, : n // 2] = 32767
        d = ctx.to_device(x)
        node.update(d, n)
        got = d.download()
        for c in (0, 1, 63, 64, 127):
            assert np.array_equal(got[c], orc.biquad_teensy_update(refs[c], x[c])), (blk, c)
    for c in (0, 63, 64, 127):
        assert list(node.definition(c)) == list(refs[c].definition), c
    node.set_coefficients(1, coef)                                 # two stages now: per-lane kernel, same stream
    orc.lib.orc_biquad_teensy_set_coefficients(orclib.C.byref(refs[1]), 1, coef.ctypes.data_as(orclib._p))      # (channel 1's model has followed every block)
    x = _rand16(rng, (ch, 2 * B), 6000)
    d = ctx.to_device(x)
    node.update(d, 2 * B)
    assert np.array_equal(d.download()[1], orc.biquad_teensy_update(refs[1], x[1]))


@pytest.mark.parametrize("scale,rng_hint", [(8000.0, None), (32767.0, None), (1.0, None), (1e-3, None), (3.0e6, None), (1e-20, None), (1e30, None),
                                            (1.0, 1.0), (3.0e6, 4.0e6)])
def test_fir_f32_matrix_core_input_ranges(ctx, orc, scale, rng_hint):
    """arm_fir_f32 on the matrix cores: fp16 pieces after a power-of-two scale chosen per tile from the data (block floating
    point), whatever the caller's units; or pinned by msdr_fir_f32_set_input_range.  Many channels, two calls, time segments."""
    rng = np.random.default_rng(int(min(scale, 1e6)) + 5)
    ch, n = 70, 9 * 1024 + 40
    h = (np.sinc(0.23 * (np.arange(256) - 127.5)) * np.kaiser(256, 7.0)).astype(np.float32)
    x = (rng.uniform(-1, 1, (ch, n)) * scale).astype(np.float32)
    x[1] *= 1e-3                                                   # a weak channel next to full-scale ones
    x[2, 3000:] *= 1e-4                                            # level steps inside a channel: the scale follows tile by tile
    x[2, 6000:] *= 1e6
    x[3, 2000:2600] = 0.0
    fir = msdr.FirF32(ctx, h, ch)
    if rng_hint is not None:
        fir.set_input_range(rng_hint)
    got = np.empty_like(x)
    for o, m in ((0, 5000), (5000, n - 5000)):
        dx, dy = ctx.to_device(x[:, o:o + m]), ctx.array((ch, m), np.float32)
        fir.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
    for c in (0, 1, 2, 3, 37, ch - 1):
        if rng_hint is not None and c == 2:
            continue                                               # the steps leave the declared range
        want = orc.fir_f32_blocks(h, x[c, :(n // 128) * 128], 128)
        err = rel_rms(got[c, :want.size], want)
        assert err < (2e-5 if (rng_hint is not None and c == 1) else 1e-6), (scale, c, err)     # a pinned scale: absolute accuracy only
    truth = np.convolve(x[0].astype(np.float64), h.astype(np.float64))[:n]
    assert rel_rms(got[0], truth) < 1e-6
    # level step: every stretch on its own (the quiet stretch must not be drowned by the loud one's rounding)
    if rng_hint is None:
        t2 = np.convolve(x[2].astype(np.float64), h.astype(np.float64))[:n]
        for lo, hi, tol in ((400, 2900, 2e-6), (4200, 4900, 2e-6), (3600, 5900, 2e-5), (8200, n, 2e-6)):     # a tile next to a loud stretch carries the loud scale
            assert rel_rms(got[2, lo:hi], t2[lo:hi]) < tol, (scale, lo)


@pytest.mark.parametrize("ntaps", [16, 100, 256, 512])
def test_fir_q15_matrix_core_many_channels_and_segments(ctx, orc, golden, ntaps):
    """arm_fir_fast_q15 batched on the integer matrix cores: many channels, ragged calls (unaligned rows, partial tiles, time
    segments), full-scale input; bit-exact against the oracle (pinned to the compiled reference)."""
    rng = np.random.default_rng(ntaps)
    ch, n = 70, 9 * 1024 + 128
    taps = rng.integers(-2500, 2501, ntaps).astype(np.int16)
    taps[::7] = 32639                                             # the largest tap that still splits into two signed bytes
    taps[3::11] = -32768
    x = rng.integers(-32768, 32768, (ch, n)).astype(np.int16)
    x[2] = -32768
    fir = msdr.FirQ15(ctx, taps, ch)
    got = np.empty_like(x)
    o = 0
    for m in (5000, 130, 7, n):
        m = min(m, n - o)
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.int16)
        fir.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
        o += m
        if o >= n:
            break
    for c in (0, 1, 2, 37, ch - 1):
        rc, want = orc.fir_q15_blocks(taps, x[c], 128)
        assert rc == 0 and np.array_equal(got[c], want), (ntaps, c)


@pytest.mark.parametrize("spec", [[(1, 300, 0.7)], [(1, 300, 0.7), (1, 300, 0.7)], [(1, 1000, 4), (1, 1500, 4), (1, 800, 2)],
                                  [(3, 1000, 8), (3, 2000, 8), (3, 3000, 8), (3, 4000, 8)], [(0, 5400, 0.54), (3, 3000, 15)]])
def test_biquad_df1_f32_ill_conditioned_cascades(ctx, orc, spec):
    """High-pass and narrow-notch cascades: the 'numerators first' parallel evaluation would lose accuracy (two 300 Hz high-pass
    sections: 8e-3); the library measures the cascade's conditioning and runs such filters in CMSIS order.  Signal with a strong
    DC term, as an envelope has; many channels, ragged calls, state carried, in place."""
    rng = np.random.default_rng(len(spec))
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    coeffs = []
    for kind, f, q in spec:
        c = orc.biquad_design(kind, np.float32(f * corr), q).astype(np.float64) / 2 ** 30
        coeffs.append([c[0], c[1], c[2], -c[3], -c[4]])
    coeffs = np.array(coeffs, np.float32)
    ch, n = 70, 6000
    x = (0.6 + 0.3 * rng.uniform(-1, 1, (ch, n))).astype(np.float32)
    bq = msdr.BiquadDf1F32(ctx, coeffs, ch)
    got = np.empty_like(x)
    for o, m in ((0, 128), (128, 3001), (3129, n - 3129)):
        d = ctx.to_device(np.ascontiguousarray(x[:, o:o + m]))
        bq.process(d, d, m)                                       # in place
        got[:, o:o + m] = d.download()
    from scipy.signal import lfilter
    for c in (0, 33, ch - 1):
        want = orc.biquad_df1_blocks(coeffs, x[c, :(n // 128) * 128], 128)
        truth = x[c].astype(np.float64)
        for r in coeffs.astype(np.float64):
            truth = lfilter(r[:3], [1.0, -r[3], -r[4]], truth)
        tail = slice(1000, want.size)                             # past the start-up transient of the DC step
        assert rel_rms(got[c, tail], truth[tail]) < 2e-5, (spec, c)
        assert rel_rms(got[c, tail], want[tail]) < 2e-5, (spec, c)
        assert rel_rms(got[c, :want.size], want) < 5e-6, (spec, c)


@pytest.mark.parametrize("ch,spec", [(1, [(1, 300, 0.7), (1, 300, 0.7)]), (3, [(3, 1000, 8), (3, 2000, 8), (3, 3000, 8)]), (1, [(1, 300, 0.7)])])
def test_biquad_df1_f32_cmsis_order_long_stream_of_few_channels(ctx, orc, ch, spec):
    """The CMSIS-order fallback on a long block of few channels runs one lane per (channel, time segment), every segment warmed
    up over the samples in front of it (copied aside first: in place stays allowed).  Odd lengths, in place and out of place,
    state carried from call to call; compared with the sequential oracle over the whole stream."""
    rng = np.random.default_rng(ch + len(spec))
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0
    coeffs = []
    for kind, f, q in spec:
        c = orc.biquad_design(kind, np.float32(f * corr), q).astype(np.float64) / 2 ** 30
        coeffs.append([c[0], c[1], c[2], -c[3], -c[4]])
    coeffs = np.array(coeffs, np.float32)
    lens = [(1 << 20) + 3, 777, (1 << 19), 300001]
    n = sum(lens)
    x = (0.6 + 0.3 * rng.uniform(-1, 1, (ch, n))).astype(np.float32)
    bq = msdr.BiquadDf1F32(ctx, coeffs, ch)
    got = np.empty_like(x)
    o = 0
    for j, m in enumerate(lens):
        d = ctx.to_device(np.ascontiguousarray(x[:, o:o + m]))
        if j % 2:
            e = ctx.array((ch, m), np.float32)
            bq.process(d, e, m)
            got[:, o:o + m] = e.download()
        else:
            bq.process(d, d, m)                                   # in place
            got[:, o:o + m] = d.download()
        o += m
    for c in range(ch):
        want = orc.biquad_df1_blocks(coeffs, x[c], n)
        tail = slice(2000, n)
        assert rel_rms(got[c, tail], want[tail]) < 1e-5, (spec, c, rel_rms(got[c, tail], want[tail]))
        # inside a later call and across its segment borders the two agree sample by sample to the cascade's own fp32 noise
        assert np.abs(got[c, tail] - want[tail]).max() < 1e-4 * np.abs(want[tail]).max() + 1e-6, (spec, c)


@pytest.mark.parametrize("ntaps", [16, 17, 33, 48, 49, 64, 65, 97, 113, 128, 129, 161, 193, 225, 242, 257, 289, 290, 321,
                                   352, 385, 417, 449, 481, 512, 513, 545, 577])      # (round 5: 321 - 577, C5's 512 among them: fir_f32mf_kernel, the same bound)
def test_fir_f32_taps_in_registers_every_step_count(ctx, orc, ntaps):
    """msdr_fir_f32tr.hiph: one instantiation per number of 32-sample k-steps (2 .. 10: 16 .. 289 taps; longer filters stay on
    msdr_fir_f32mf.hiph).  Tap counts on both sides of every boundary, with and without the all-zero first step of the second
    fragment family; many channels, time segments, a ragged second call, asymmetric taps (a mirrored table would show)."""
    rng = np.random.default_rng(ntaps)
    ch, n1, n2 = 37, 11 * 1024, 5 * 1024 + 333
    h = (rng.standard_normal(ntaps) * np.linspace(1.0, 0.2, ntaps)).astype(np.float32)
    x = (rng.uniform(-1, 1, (ch, n1 + n2)) * 5000.0).astype(np.float32)
    fir = msdr.FirF32(ctx, h, ch)
    got = np.empty_like(x)
    for o, m in ((0, n1), (n1, n2)):
        dx, dy = ctx.to_device(x[:, o:o + m]), ctx.array((ch, m), np.float32)
        fir.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
    for c in (0, 1, 18, ch - 1):
        truth = np.convolve(x[c].astype(np.float64), h[::-1].astype(np.float64))[:n1 + n2]      # arm_fir_f32 keeps its taps time-reversed
        assert rel_rms(got[c], truth) < 1e-6, (ntaps, c, rel_rms(got[c], truth))
    want = orc.fir_f32_blocks(h, x[5, :(n1 + n2) // 128 * 128], 128)
    assert rel_rms(got[5, :want.size], want) < 1e-6


def test_fir_f32_isolated_spike(ctx, orc):
    """One 1e6 sample in a unit-variance row (arm_fir_f32 itself has no input-dependent precision, arm_math.h:1182-1186): the quiet
    outputs BEFORE the spike -- whose causal windows do not hold it, though the tile's block-floating-point scale does -- and the
    outputs more than N after it must keep 1e-5 relative RMS; the stretch the spike dominates is judged as a whole.  The bound the
    library states in include/msdr.h (msdr_fir_f32_set_input_range): every sample x of a tile's window enters the products with an error
    of at most max(2^-21 |x|, 2^-39 Mw), Mw the window's largest magnitude -- full 22 bits down to 2^-18 of Mw, 22 - (R - 18) bits at
    2^-R.  Here R = 20 (1e6 against unit variance): the quiet samples of the spike's own tile window keep 20 bits, an absolute error of
    2^-39 x 1e6 = 1.8e-6 each, which is what the 1e-5 bound on the quiet stretches of that tile allows for; tiles whose window does
    not hold the spike are untouched."""
    rng = np.random.default_rng(77)
    ntaps, ch, n = 256, 8, 6 * 1024
    h = (np.sinc(0.23 * (np.arange(ntaps) - 127.5)) * np.kaiser(ntaps, 7.0)).astype(np.float32)
    x = rng.standard_normal((ch, n)).astype(np.float32)
    pos = {0: 2 * 1024 + 1023, 1: 2 * 1024 + 1, 2: 3 * 1024 + 500, 3: 1024 + 700}     # last / first sample of a tile, inside one
    for c, p_ in pos.items():
        x[c, p_] = 1.0e6
    fir = msdr.FirF32(ctx, h, ch)
    dx, dy = ctx.to_device(x), ctx.array((ch, n), np.float32)
    fir.process(dx, dy, n)
    got = dy.download()
    for c, p_ in pos.items():
        truth = np.convolve(x[c].astype(np.float64), h.astype(np.float64))[:n]
        tile0 = (p_ // 1024) * 1024
        before = slice(max(0, tile0 - 1024), p_)                   # the spike's own tile up to the spike, and the tile before it
        after = slice(p_ + ntaps, min(n, p_ + ntaps + 2048))
        assert rel_rms(got[c, before], truth[before]) < 1e-5, (c, "before", rel_rms(got[c, before], truth[before]))
        assert rel_rms(got[c, after], truth[after]) < 1e-5, (c, "after", rel_rms(got[c, after], truth[after]))
        assert rel_rms(got[c, p_:p_ + ntaps], truth[p_:p_ + ntaps]) < 1e-6, (c, "spike")
    for c in (4, 7):                                               # rows without a spike: the usual bound
        assert rel_rms(got[c], orc.fir_f32_blocks(h, x[c], 128)) < 1e-6


@pytest.mark.parametrize("ntaps,ch,sizes", [(256, 5, (1, 3, 127, 128, 1023, 1024, 1025, 4099)), (100, 300, (128, 128, 2048 + 6, 130)),
                                            (33, 3, (50000, 7, 1024 * 9)), (55, 70, (2432,)), (274, 300, (9344,)), (143, 70, (1920, 2560, 3456, 1792))])
def test_fir_f32_tile_queue_ragged_calls(ctx, orc, ntaps, ch, sizes):
    """fir_f32tq_kernel deals tiles from a queue in address order; every tile fetches its own halo -- from the row, or from the history
    the previous call left (row-opening tiles), zeros beyond the row's end.  Calls of ragged lengths (unaligned rows, partial tiles,
    fewer tiles than waves, more fronts than tiles), the stream continuing across calls, EVERY output of every row against a float64
    convolution; the output buffer starts as NaN (a tile nobody computed shows).  The 70- and 300-channel shapes are the ones
    tests/debug/fuzz_fir_f32.py found in round 3: few waves per front, so that one wave draws a row's partial last tile BETWEEN two
    complete ones -- which entered the steady-state loop as its current tile and left as four whole 1 KB stores, over the first
    samples of the next row."""
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(ntaps + ch)
    h = (rng.standard_normal(ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
    n = sum(sizes)
    x = rng.uniform(-8000, 8000, (ch, n)).astype(np.float32)
    fir = msdr.FirF32(ctx, h, ch)
    got = np.empty_like(x)
    o = 0
    for m in sizes:
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.to_device(np.full((ch, m), np.nan, np.float32))
        fir.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
        o += m
    assert not np.isnan(got).any()
    truth = fftconvolve(x.astype(np.float64), h.astype(np.float64)[::-1][None, :], axes=1)[:, :n]      # y[n] = sum_k h[k] x[n - (N-1) + k]: h time-reversed
    err = np.sqrt(((got - truth) ** 2).sum(axis=1) / (truth ** 2).sum(axis=1))
    assert err.max() < 1e-6, (int(err.argmax()), float(err.max()))
    o = 0
    for m in sizes:                                                # ... and call by call, tile by tile on the worst row (a bad stretch of 128 samples in 10^4)
        c = int(err.argmax())
        for t in range(o, o + m, 1024):
            hi = min(o + m, t + 1024)
            scale = np.sqrt((truth[c, o:o + m] ** 2).mean())
            assert np.abs(got[:, t:hi] - truth[:, t:hi]).max() < 2e-5 * max(scale, np.sqrt((truth[:, o:o + m] ** 2).mean(axis=1)).max()), (c, t)
        o += m
    assert rel_rms(got[0], orc.fir_f32_blocks(h, x[0], 128)) < 1e-6                        # and the oracle's own (fp32, sequential) run


def test_fir_f32_every_output_of_a_batch_larger_than_the_resident_waves(ctx):
    """More tiles than the 2048 resident waves of fir_f32tq_kernel, most of them "hot" (drawn from the queue, fetched and converted in the
    steady-state pipeline), two calls so that row-opening tiles take their halo from the carried history: EVERY output of EVERY row is
    compared with a float64 FFT convolution (a wrong tile index, a halo of the wrong row or a tile done twice shows as one bad row)."""
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(2024)
    ntaps, ch = 256, 96
    sizes = (40 * 1024 + 333, 7 * 1024)
    n = sum(sizes)
    h = (rng.standard_normal(ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
    x = rng.uniform(-8000, 8000, (ch, n)).astype(np.float32)
    x[:, ::1024] += 5.0e4                                          # a marker on every tile's first sample
    fir = msdr.FirF32(ctx, h, ch)
    got = np.empty_like(x)
    o = 0
    for m in sizes:
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.float32)
        fir.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
        o += m
    truth = fftconvolve(x.astype(np.float64), h.astype(np.float64)[::-1][None, :], axes=1)[:, :n]
    err = np.sqrt(((got - truth) ** 2).sum(axis=1) / (truth ** 2).sum(axis=1))
    assert err.max() < 1e-6, (int(err.argmax()), float(err.max()))
    # and tile by tile on a few rows (a single bad tile of 1024 in 48 000 samples would pass a per-row 1e-6 only if it were tiny)
    for c in (0, 1, 47, ch - 1):
        d = (got[c] - truth[c]).reshape(-1)[:(n // 1024) * 1024].reshape(-1, 1024)
        t = truth[c][:(n // 1024) * 1024].reshape(-1, 1024)
        per_tile = np.sqrt((d ** 2).sum(axis=1) / (t ** 2).sum(axis=1))
        assert per_tile.max() < 2e-6, (c, int(per_tile.argmax()), float(per_tile.max()))


@pytest.mark.parametrize("shift", [1, 2, 3])
def test_fir_f32_tile_queue_runs_of_tiles(ctx, orc, shift, monkeypatch):
    """Round 4: a draw hands a wave a RUN of 2^shift consecutive tiles, and the second .. last tile of a run inside one row take their
    1 KB halo from the registers that hold the tile before (256-tap class: the halo is exactly the window's first load).  By itself the
    library uses runs only where every wave still gets >= 16 of them (the bench shape); MSDR_TQ_RUN_SHIFT forces them here: rows of 1,
    2^shift - 1, 2^shift + 1 and many tiles, rows that end inside a run, runs that straddle rows, two calls (history), EVERY output of
    every row against a float64 convolution, NaN-filled output; 100 taps as well (runs without the register halo)."""
    from scipy.signal import fftconvolve
    monkeypatch.setenv("MSDR_TQ_RUN_SHIFT", str(shift))
    for ntaps, ch, sizes in ((256, 96, (40 * 1024 + 333, 7 * 1024)), (256, 7, (1024 * ((1 << shift) + 1), 1024 * ((1 << shift) - 1) + 5, 1024)),
                             (241, 300, (9 * 1024,)), (100, 64, (12 * 1024 + 17, 2048))):
        rng = np.random.default_rng(ntaps + ch + shift)
        h = (rng.standard_normal(ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
        n = sum(sizes)
        x = rng.uniform(-8000, 8000, (ch, n)).astype(np.float32)
        x[:, ::1024] += 5.0e4                                      # a marker on every tile's first sample
        fir = msdr.FirF32(ctx, h, ch)
        got = np.empty_like(x)
        o = 0
        for m in sizes:
            dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.to_device(np.full((ch, m), np.nan, np.float32))
            fir.process(dx, dy, m)
            got[:, o:o + m] = dy.download()
            o += m
        assert not np.isnan(got).any(), (ntaps, ch)
        truth = fftconvolve(x.astype(np.float64), h.astype(np.float64)[::-1][None, :], axes=1)[:, :n]
        err = np.sqrt(((got - truth) ** 2).sum(axis=1) / (truth ** 2).sum(axis=1))
        assert err.max() < 1e-6, (ntaps, ch, int(err.argmax()), float(err.max()))
        full = (n // 1024) * 1024
        d = (got[:, :full] - truth[:, :full]).reshape(ch, -1, 1024)
        t = truth[:, :full].reshape(ch, -1, 1024)
        per_tile = np.sqrt((d ** 2).sum(axis=2) / (t ** 2).sum(axis=2))
        assert per_tile.max() < 3e-6, (ntaps, ch, np.unravel_index(int(per_tile.argmax()), per_tile.shape), float(per_tile.max()))


def test_fir_f32_stage_replays_from_a_hip_graph(ctx, orc):
    """Round 5: fir_f32tq_kernel leaves its tile queue as it found it (its last wave out re-zeroes the counters; rounds 3 - 4 alternated two
    counter sets on the host, so a replayed launch found its counters spent -- the review's weak point 8).  Two consecutive
    msdr_fir_f32_process calls (the history buffers alternate: an even number) captured on the context's stream into ONE HIP graph, replayed
    five times over fresh input: ten blocks of one continuing stream, every output against a float64 convolution."""
    import ctypes as C
    from scipy.signal import fftconvolve
    hip = C.CDLL("libamdhip64.so")
    rng = np.random.default_rng(58)
    ntaps, ch, m = 256, 40, 4096
    h = (rng.standard_normal(ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
    fir = msdr.FirF32(ctx, h, ch)
    assert fir.kernel_name().startswith("fir_f32tq_kernel"), fir.kernel_name()
    reps = 5
    x = rng.uniform(-8000, 8000, (ch, (2 + 2 * reps) * m)).astype(np.float32)
    dx = [ctx.array((ch, m), np.float32) for _ in range(2)]
    dy = [ctx.array((ch, m), np.float32) for _ in range(2)]
    got = np.empty_like(x)
    o = 0
    for k in range(2):                                             # two direct calls first (nothing is allocated lazily inside the capture)
        dx[k].upload(x[:, o:o + m]); fir.process(dx[k], dy[k], m); got[:, o:o + m] = dy[k].download(); o += m
    stream = C.c_void_p(ctx.stream())
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, C.c_int(2)) == 0      # hipStreamCaptureModeRelaxed
    try:
        for k in range(2):
            fir.process(dx[k], dy[k], m)
    finally:
        rc = hip.hipStreamEndCapture(stream, C.byref(graph))
    assert rc == 0 and graph.value
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
    try:
        for _ in range(reps):
            for k in range(2):
                dx[k].upload(x[:, o + k * m:o + (k + 1) * m])
                dy[k].upload(np.full((ch, m), np.nan, np.float32))
            assert hip.hipGraphLaunch(gexec, stream) == 0
            for k in range(2):
                got[:, o:o + m] = dy[k].download(); o += m
    finally:
        hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
    assert not np.isnan(got).any()
    truth = fftconvolve(x.astype(np.float64), h.astype(np.float64)[::-1][None, :], axes=1)[:, :x.shape[1]]
    err = np.sqrt(((got - truth) ** 2).sum(axis=1) / (truth ** 2).sum(axis=1))
    assert err.max() < 1e-6, (int(err.argmax()), float(err.max()))
    # ... and the stage goes on from where the graph left it, by direct calls
    x2 = rng.uniform(-8000, 8000, (ch, m)).astype(np.float32)
    dx[0].upload(x2); fir.process(dx[0], dy[0], m)
    t2 = fftconvolve(np.concatenate([x, x2], axis=1).astype(np.float64), h.astype(np.float64)[::-1][None, :], axes=1)[:, x.shape[1]:x.shape[1] + m]
    assert np.sqrt(((dy[0].download() - t2) ** 2).sum() / (t2 ** 2).sum()) < 1e-6


@pytest.mark.parametrize("ntaps,ch", [(102, 300), (256, 37), (512, 64), (16, 5)])
def test_fir_q15_stage_at_block_cadence(ctx, orc, monkeypatch, ntaps, ch):
    """`arm_fir_fast_q15(&FIR_I, I_buffer, I_FIR_out, AUDIO_BLOCK_SAMPLES)` (Minimal-SDR.ino:574-575) on a batch of channels: calls of 32 ..
    512 samples take chain_q15mb_kernel's FIR-stage flavour (channel-batched tiles, the next history written by the same kernel), calls of
    other lengths in between take the long-call kernel -- one history format, every hand-over exact.  Bit-exact against the oracle (pinned to
    the compiled reference) and against the same stream through the long-call kernel alone (MSDR_NO_BLOCK=1); full-scale and -32768 rows."""
    rng = np.random.default_rng(ntaps + ch)
    taps = rng.integers(-2500, 2501, ntaps).astype(np.int16)
    taps[::7] = 32639
    taps[3::11] = -32768
    plan = [128] * 20 + [64] * 4 + [512, 1000, 7] + [128] * 8 + [32] * 8 + [256, 130] + [128] * 6
    n = sum(plan)
    x = rng.integers(-32768, 32768, (ch, n)).astype(np.int16)
    x[min(2, ch - 1)] = -32768
    outs = []
    for no_block in (False, True):
        if no_block:
            monkeypatch.setenv("MSDR_NO_BLOCK", "1")
        else:
            monkeypatch.delenv("MSDR_NO_BLOCK", raising=False)
        fir = msdr.FirQ15(ctx, taps, ch)
        got = np.empty_like(x)
        o = 0
        for m in plan:
            dx, dy = ctx.to_device(np.ascontiguousarray(x[:, o:o + m])), ctx.array((ch, m), np.int16)
            fir.process(dx, dy, m)
            got[:, o:o + m] = dy.download()
            o += m
        outs.append(got)
        fir.close()
    monkeypatch.delenv("MSDR_NO_BLOCK", raising=False)
    assert np.array_equal(outs[0], outs[1])
    for c in sorted(set([0, 1, min(2, ch - 1), ch // 2, ch - 1])):
        rc, want = orc.fir_q15_blocks(taps, np.concatenate([x[c], np.zeros((-n) % 128, np.int16)]), 128)
        assert rc == 0 and np.array_equal(outs[0][c], want[:n]), (ntaps, c)
