"""Row f4 (second half) on the GPU: the spectrum display's 128-point q15 real FFT + column heights (UI.cpp:520-592),
through the C ABI, bit-exact against the oracle and against the reference-generated golden vectors."""
import numpy as np
import pytest

from gpuhelp import ctx, msdr  # noqa: F401

pytestmark = pytest.mark.gpu


def _run(ctx, x, stride=None):
    """x: [nfft][128] int16 -> (fft_out [nfft][256], columns [nfft][127])"""
    nfft = x.shape[0]
    stride = stride or 128
    buf = np.zeros((nfft, stride), np.int16)
    buf[:, :128] = x
    d = ctx.to_device(buf)
    o, c = ctx.array((nfft, 256), np.int16), ctx.array((nfft, 128), np.uint8)
    msdr.rfft128_q15(ctx, d, stride, nfft, o, c)
    assert np.array_equal(d.download(), buf)             # the IF block is read-only here
    col = c.download()
    assert (col[:, 127] == 0).all()
    return o.download(), col[:, :127]


def test_rfft128_matches_golden(ctx, golden):
    out, col = _run(ctx, golden["fft/x"])
    assert np.array_equal(out, golden["fft/rfft128_out"])
    want = np.minimum(np.abs(golden["fft/rfft128_out"][:, 127:0:-1].astype(np.int32)) // 200, 16)
    assert np.array_equal(col, want)


@pytest.mark.parametrize("nfft,stride", [(1, 128), (15, 128), (16, 128), (17, 136), (1000, 128), (4096, 256)])
def test_rfft128_matches_oracle(ctx, orc, nfft, stride):
    rng = np.random.default_rng(nfft)
    amp = rng.choice([30, 3000, 32767], nfft)
    x = (rng.integers(-32768, 32768, (nfft, 128)) * amp[:, None] // 32768).astype(np.int16)
    x[::7, ::5] = -32768
    x[3::11, 1::3] = 32767
    out, col = _run(ctx, x, stride)
    for f in range(nfft):
        want, _ = orc.rfft128_q15(x[f])
        assert np.array_equal(out[f], want), f
        assert np.array_equal(col[f], orc.spectrum_columns(want)), f


def test_rfft128_saturating_and_halving_paths(ctx, orc):
    """Inputs that drive the packed saturating adds (__QADD16 / __QASX ...), the halving adds at both ends of the 16-bit range and
    the wrapping twiddle sums: constants and square waves at full scale, impulses, single bins at full scale."""
    rows = []
    for v in (-32768, 32767, -1, 1, 0):
        rows.append(np.full(128, v))
    k = np.arange(128)
    for period in (2, 4, 8, 16, 32, 64, 128):
        sq = np.where((k // (period // 2)) % 2 == 0, 32767, -32768)
        rows += [sq, -1 - sq, np.roll(sq, 1), np.where(k % 2 == 0, sq, 0), np.where(k % 2 == 1, sq, 0)]
    for pos in (0, 1, 2, 63, 64, 126, 127):
        for v in (-32768, 32767):
            imp = np.zeros(128)
            imp[pos] = v
            rows.append(imp)
    for b in (1, 3, 16, 31, 32, 33, 63):
        for ph in (0.0, 0.7):
            rows.append(np.clip(np.round(40000 * np.cos(2 * np.pi * b * k / 128 + ph)), -32768, 32767))
    x = np.array(rows).astype(np.int16)
    out, col = _run(ctx, x)
    for f in range(x.shape[0]):
        want, _ = orc.rfft128_q15(x[f])
        assert np.array_equal(out[f], want), f
        assert np.array_equal(col[f], orc.spectrum_columns(want)), f


def test_rfft128_more_batches_than_workgroups(ctx, orc):
    """The launch is capped at 8192 workgroups of 16 transforms: beyond 131 072 transforms a workgroup walks several batches, the next
    batch's samples already on their way while one is transformed.  Checked: the head, the seams between walks and the ragged tail."""
    nfft = 2 * 8192 * 16 + 16 * 5 + 3
    rng = np.random.default_rng(77)
    x = rng.integers(-32768, 32768, (nfft, 128)).astype(np.int16)
    out, col = _run(ctx, x)
    picks = np.unique(np.concatenate([np.arange(0, 40), np.arange(131072 - 20, 131072 + 40), np.arange(262144 - 20, nfft),
                                      rng.integers(0, nfft, 300)]))
    for f in picks:
        want, _ = orc.rfft128_q15(x[f])
        assert np.array_equal(out[f], want), f
        assert np.array_equal(col[f], orc.spectrum_columns(want)), f


def test_rfft128_outputs_optional_and_argument_checks(ctx):
    x = np.arange(256, dtype=np.int16).reshape(2, 128)
    d = ctx.to_device(x)
    o = ctx.array((2, 256), np.int16)
    msdr.rfft128_q15(ctx, d, 128, 2, o, None)
    c = ctx.array((2, 128), np.uint8)
    msdr.rfft128_q15(ctx, d, 128, 2, None, c)
    msdr.rfft128_q15(ctx, d, 128, 0, o, c)
    with pytest.raises(msdr.MsdrError):
        msdr.rfft128_q15(ctx, d, 100, 2, o, c)            # stride not a multiple of 8
    with pytest.raises(msdr.MsdrError):
        msdr.rfft128_q15(ctx, d, 64, 2, o, c)             # overlapping transforms


def test_show_spectrum_cadence_over_block_stream(ctx, orc):
    """showSpectrum(data) once per dequeued block (Minimal-SDR.ino:532): draws on the 1st call and then every 25th."""
    ch, blocks = 6, 60
    rng = np.random.default_rng(77)
    x = rng.integers(-9000, 9001, (ch, blocks * 128)).astype(np.int16)
    d = ctx.to_device(x)
    sp = msdr.Spectrum(ctx, ch)
    o, c = ctx.array((ch, 256), np.int16), ctx.array((ch, 128), np.uint8)
    drawn = []
    for b in range(blocks):
        if b == 40:
            sp.set_on(False)
        if b == 45:
            sp.set_on(True)
        if sp.show(d.offset(b * 128 * 2), blocks * 128, o, c):
            drawn.append(b)
            out, col = o.download(), c.download()
            for k in range(ch):
                want, _ = orc.rfft128_q15(x[k, b * 128:(b + 1) * 128])
                assert np.array_equal(out[k], want)
                assert np.array_equal(col[k, :127], orc.spectrum_columns(want))
    # counter: 0 -> draws at call 0, then 25; Spectrum_on == 0 skips the calls without counting (UI.cpp:533 returns first)
    assert drawn == [0, 25, 55]
    sp.close()


def test_spectrum_tables_survive_freqconv_on_the_same_context(orc):
    """The reference application runs freq_conv and the spectrum display side by side.  The first msdr_freqconv_* call of a
    context grows its scratch buffer; that must not touch the FFT twiddle tables an earlier msdr_rfft128_q15 uploaded."""
    own = msdr.Context(0)                                 # a fresh context: scratch_bytes starts at 0
    try:
        rng = np.random.default_rng(5)
        x = rng.integers(-20000, 20001, (64, 128)).astype(np.int16)
        before, colb = _run(own, x)
        n = np.arange(128)
        oi = np.round(32767 * np.sin(2 * np.pi * 5 * n / 128)).astype(np.int16)
        oq = np.round(32767 * np.cos(2 * np.pi * 5 * n / 128)).astype(np.int16)
        i0 = rng.integers(-30000, 30001, (8, 128)).astype(np.int16)
        q0 = np.zeros_like(i0)
        di, dq = own.to_device(i0), own.to_device(q0)
        own.freqconv_q15(di, dq, oi, oq, 1, 1, 8, 128)
        gi, gq = di.download(), dq.download()
        for c in range(8):
            wi, wq = orc.freqconv_q15(i0[c], q0[c], oi, oq, 1, 1)
            assert np.array_equal(gi[c], wi) and np.array_equal(gq[c], wq)
        after, cola = _run(own, x)
        assert np.array_equal(before, after) and np.array_equal(colb, cola)
        for f in range(0, 64, 9):
            want, _ = orc.rfft128_q15(x[f])
            assert np.array_equal(after[f], want)
    finally:
        own.close()
