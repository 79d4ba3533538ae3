"""Parity at BASELINE.json's FULL sizes, through properties that do not need the oracle to chew the
whole data set (it would take minutes per case):

  * spot windows: the oracle recomputes randomly placed windows of the output -- including windows that
    straddle the GPU's time-segment boundaries -- from the raw IF with enough pre-roll for FIR history and
    IIR convergence, and must agree to 1e-5 relative RMS (bit-exact for the q15 FIR stage);
  * determinism / idempotence: processing the same stream twice from reset gives bit-identical output;
  * state carry: one long call == the same stream in two calls (bit-identical when no time split moves);
  * channel independence: identical IF in two channels gives identical audio, and permuting channels
    permutes the audio.
Device buffers come from torch (plumbing only); every kernel call goes through the C ABI."""
import numpy as np
import pytest

import orclib
from gpuhelp import msdr, rel_rms  # noqa: F401

pytestmark = pytest.mark.gpu
TOL = 1e-5
FS = 24000.0


@pytest.fixture(scope="module")
def tctx():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = msdr.Context(0, stream.cuda_stream)
    yield torch, ctx
    torch.cuda.synchronize()
    ctx.close()


def _hilbert(n_taps, fc=1330.0, bw=1920.0):
    k = np.arange(n_taps)
    m = (n_taps - 1) / 2.0
    p = np.sinc(bw / FS * (k - m)) * np.kaiser(n_taps, 6.0)
    p /= p.sum()
    w = 2 * np.pi * fc / FS
    return (2 * p * np.cos(w * (k - m) + np.pi / 4)).astype(np.float32), (2 * p * np.cos(w * (k - m) - np.pi / 4)).astype(np.float32)


def _lowpass(n_taps, fc=2800.0):
    k = np.arange(n_taps)
    h = np.sinc(2 * fc / FS * (k - (n_taps - 1) / 2.0)) * np.kaiser(n_taps, 7.0)
    return (h / h.sum()).astype(np.float32)


def _bq(orc):
    corr = orclib.AUDIO_SAMPLE_RATE_EXACT / FS
    out = []
    for kind, f, q in ((orclib.BQ_LOWPASS, 5400 * corr, 0.54), (orclib.BQ_NOTCH, 3000 * corr, 15.0)):
        c = orc.biquad_design(kind, np.float32(f), q).astype(np.float64) / 2 ** 30
        out.append([c[0], c[1], c[2], -c[3], -c[4]])
    return np.array(out, np.float32)


def _q15_nco4():
    n = np.arange(128)
    s = np.round(32767 * np.sin(2 * np.pi * n / 4)).astype(np.int16)
    c = np.round(32767 * np.cos(2 * np.pi * n / 4)).astype(np.int16)
    return (s / 32768.0).astype(np.float32), (c / 32768.0).astype(np.float32)


def _synth(torch, channels, n, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.empty((channels, n), dtype=torch.int16, device="cuda")
    flat = x.view(-1)
    step = 1 << 24
    for o in range(0, channels * n, step):
        m = min(step, channels * n - o)
        flat[o:o + m] = torch.randint(-8000, 8001, (m,), device="cuda", generator=g, dtype=torch.int32).to(torch.int16)
    torch.cuda.synchronize()
    return x


def _window_check(orc, x_row, y_row, lo, length, mode, hi, hq, osc_i, osc_q, bq, preroll, phase_period):
    """Oracle on [lo - preroll, lo + length) of one channel; compares the last `length` samples."""
    start = max(0, lo - preroll)
    start -= start % phase_period                      # keep the oscillator phase: restart on a period boundary
    xs = x_row[start:lo + length].cpu().numpy()
    if osc_i is None:
        oi, oq = np.array([0, 1, 0, -1], np.float32), np.array([1, 0, -1, 0], np.float32)
    else:
        oi, oq = osc_i, osc_q
    want = orc.chain_f32(xs, mode, hi, hq, oi, oq, bq)[lo - start:]
    got = y_row[lo:lo + length].cpu().numpy()
    return rel_rms(got, want)


def test_c2_one_channel_one_gisample(tctx, orc):
    """configs[1]: 1 SSB channel, 2^30 int16 IF samples in ONE call (2 GiB in, 4 GiB out)."""
    torch, ctx = tctx
    n = 1 << 30
    hi, hq = _hilbert(100)
    oi, oq = _q15_nco4()
    bq = _bq(orc)
    x = _synth(torch, 1, n, 2)
    y = torch.empty((1, n), dtype=torch.float32, device="cuda")
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 1, hi, hq, mixer=msdr.MIXER_NCO, mode=orclib.LSB, osc_i=oi, osc_q=oq, biquad_coeffs=bq)
    chain.process(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize()
    info = chain.info()
    assert info["time_segments"] > 1000
    seg = -(-n // info["time_segments"])
    seg = -(-seg // info["tile"]) * info["tile"]
    rng = np.random.default_rng(0)
    # windows: stream start, segment boundaries (straddling), random interior, the very end
    los = [0, seg - 200, 7 * seg - 200, (info["time_segments"] - 1) * seg - 200, n - 4096]
    los += [int(v) for v in rng.integers(seg, n - 8192, 4)]
    for lo in los:
        lo = max(0, min(lo, n - 4096))
        err = _window_check(orc, x[0], y[0], lo, 4096, orclib.LSB, hi, hq, oi, oq, bq, preroll=8192, phase_period=4)
        assert err < TOL, (lo, err)
    # determinism: same stream again from reset
    s1 = (float(y.double().sum()), float(y[:, ::4099].double().abs().sum()))
    y2 = torch.empty_like(y)
    chain.reset()
    chain.process(x.data_ptr(), y2.data_ptr(), n)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    assert s1 == (float(y2.double().sum()), float(y2[:, ::4099].double().abs().sum()))


def test_c3_4096_am_channels_256_taps(tctx, orc):
    """configs[2]: 4096 AM channels x 2^18 samples, 256-tap low-pass pair."""
    torch, ctx = tctx
    ch, n = 4096, 1 << 18
    lp = _lowpass(256)
    bq = _bq(orc)
    x = _synth(torch, ch, n, 3)
    x[17] = x[4000]                                    # two channels with identical IF
    torch.cuda.synchronize()
    y = torch.empty((ch, n), dtype=torch.float32, device="cuda")
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
    chain.process(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize()
    assert torch.equal(y[17], y[4000])                 # channel independence
    rng = np.random.default_rng(1)
    for c in [0, 4095] + [int(v) for v in rng.integers(1, 4095, 3)]:
        for lo in (0, int(rng.integers(8192, n - 4096))):
            err = _window_check(orc, x[c], y[c], lo, 4096, orclib.AM, lp, lp, None, None, bq, preroll=8192, phase_period=4)
            assert err < TOL, (c, lo, err)
    # state carry: the same stream as two calls (n1 not a multiple of anything nice)
    chain.reset()
    n1 = 100003
    y2 = torch.empty_like(y)
    xa, xb = x[:, :n1].contiguous(), x[:, n1:].contiguous()
    ya = torch.empty((ch, n1), dtype=torch.float32, device="cuda")
    yb = torch.empty((ch, n - n1), dtype=torch.float32, device="cuda")
    chain.process(xa.data_ptr(), ya.data_ptr(), n1)
    chain.process(xb.data_ptr(), yb.data_ptr(), n - n1)
    torch.cuda.synchronize()
    y2 = torch.cat([ya, yb], dim=1)
    d = (y2 - y).double()
    assert float(d.pow(2).sum().sqrt() / y.double().pow(2).sum().sqrt()) < 2e-6


def test_c4_shard_of_8192_ssb_channels(tctx, orc):
    """configs[3], one GPU's shard: 8192 SSB channels x 2^14 samples, 100-tap pair; permutation property."""
    torch, ctx = tctx
    ch, n = 8192, 1 << 14
    hi, hq = _hilbert(100)
    oi, oq = _q15_nco4()
    bq = _bq(orc)
    x = _synth(torch, ch, n, 4)
    y = torch.empty((ch, n), dtype=torch.float32, device="cuda")
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, hi, hq, mixer=msdr.MIXER_NCO, mode=orclib.LSB, osc_i=oi, osc_q=oq, biquad_coeffs=bq)
    chain.process(x.data_ptr(), y.data_ptr(), n)
    perm = torch.randperm(ch, device="cuda")
    xp = x[perm].contiguous()
    yp = torch.empty_like(y)
    chain.reset()
    chain.process(xp.data_ptr(), yp.data_ptr(), n)
    torch.cuda.synchronize()
    assert torch.equal(yp, y[perm])
    rng = np.random.default_rng(2)
    for c in [0, ch - 1] + [int(v) for v in rng.integers(1, ch - 1, 4)]:
        want = orc.chain_f32(x[c].cpu().numpy(), orclib.LSB, hi, hq, oi, oq, bq)
        assert rel_rms(y[c].cpu().numpy(), want) < TOL, c


def test_c5_mixed_modes_512_taps_1m_blocks(tctx, orc):
    """configs[4], one GPU's share in small channel count: per-channel AM/LSB, 512 taps, 2^20-sample blocks."""
    torch, ctx = tctx
    ch, n = 64, 1 << 20
    lp = _lowpass(512)
    hi, hq = _hilbert(512)
    bq = _bq(orc)
    modes = np.array([orclib.AM if (c * 2654435761 >> 7) & 1 else orclib.LSB for c in range(ch)], np.int32)
    tapsets = np.array([0 if m == orclib.AM else 1 for m in modes], np.int32)
    x = _synth(torch, ch, n, 5)
    y = torch.empty((ch, n), dtype=torch.float32, device="cuda")
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, [lp, hi], [lp, hq], mixer=msdr.MIXER_FS4, modes=modes, tapsets=tapsets, biquad_coeffs=bq)
    chain.process(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize()
    info = chain.info()
    seg = -(-n // info["time_segments"])
    seg = -(-seg // info["tile"]) * info["tile"]
    rng = np.random.default_rng(3)
    for c in [0, ch - 1] + [int(v) for v in rng.integers(1, ch - 1, 3)]:
        ti, tq = ([lp, hi][tapsets[c]], [lp, hq][tapsets[c]])
        los = [0, n - 4096] + ([seg - 300] if info["time_segments"] > 1 else [])
        for lo in los:
            err = _window_check(orc, x[c], y[c], lo, 4096, modes[c], ti, tq, None, None, bq, preroll=8192, phase_period=4)
            assert err < TOL, (c, lo, err)


def test_q15_chain_full_block_batch_bit_exact_windows(tctx, orc, golden):
    """The as-written q15 chain on 4096 channels x 2^16 samples: FIR/demod stage re-derived exactly on random
    windows (integer arithmetic: any window with N-1 samples of pre-roll must match bit for bit)."""
    torch, ctx = tctx
    ch, n = 4096, 1 << 16
    taps = golden["fir/taps_am102"]
    x = _synth(torch, ch, n, 6)
    y = torch.empty((ch, n), dtype=torch.int16, device="cuda")
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps, taps, mode=orclib.AM)
    chain.process(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize()
    rng = np.random.default_rng(4)
    for c in [0, ch - 1] + [int(v) for v in rng.integers(1, ch - 1, 6)]:
        lo = int(rng.integers(1, n // 128 - 40)) * 128               # block aligned: the Fs/4 phase restarts at 0
        xs = x[c, lo - 128:lo + 32 * 128].cpu().numpy()
        want = orc.chain_q15(xs, orclib.AM, taps, taps)[128:]        # 128 >= N-1 samples of pre-roll
        assert np.array_equal(y[c, lo:lo + 32 * 128].cpu().numpy(), want), (c, lo)
    full = orc.chain_q15(x[5].cpu().numpy(), orclib.AM, taps, taps)
    assert np.array_equal(y[5].cpu().numpy(), full)
