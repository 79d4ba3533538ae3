"""GPU parity of chain_amsy_kernel (minimal-sdr_amd/csrc/msdr_chain_amsy.hiph): the fp32 envelope chain behind the exact Fs/4 mixer for
LINEAR-PHASE taps -- what the reference's own designer produces for its AM filter (calc_FIR_coeffs, Minimal-SDR.ino:782-872: symmetric about an
integer index; bound to both FIR instances by init_FIR, :917-924).  The kernel folds the sample window about the filter's centre and runs half
the matrix products; the results must be the sequential fp32 oracle's within the chain's tolerance (1e-5 relative RMS per channel), whatever
the tap count, the centre's parity, the block length, the cascade."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr, rel_rms  # noqa: F401
from test_gpu_chain import run_chain, _f32_biquads

pytestmark = pytest.mark.gpu
TOL = 1e-5
COS4, SIN4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)


def designer(n, fc=2800.0):
    """FIR_AM_coeffs as calc_demod_filter() fills it (Minimal-SDR.ino:221-223), converted as arm_q15_to_float does"""
    return (msdr.calc_fir_coeffs(n, fc, 70.0, 0, 0.0, 24000.0)[:n].astype(np.float32) / 32768.0).astype(np.float32)


def centred(n, centre, seed):
    """random taps, symmetric about pCoeffs index `centre` (an integer), zero where the mirror image falls outside"""
    rng = np.random.default_rng(seed)
    h = np.zeros(n, np.float32)
    half = min(centre, n - 1 - centre)
    w = (rng.standard_normal(half + 1) / (1 + np.arange(half + 1))).astype(np.float32)
    for i in range(half + 1):
        h[centre - i] = w[i]
        h[centre + i] = w[i]
    return h


def check(ctx, orc, taps, stages, blocks=(None, 3073, 1000, 129), n=9001, seed=1, name="chain_amsy_kernel"):
    rng = np.random.default_rng(seed)
    bq = _f32_biquads(orc, stages)
    x = rng.integers(-32768, 32768, (3, n)).astype(np.int16)
    x[2, :64] = -32768                                       # the extreme sample: (x >> 5) = -1024, sums reach -2048
    x[2, 64:128] = 32767
    tapsets = [0, len(taps) - 1, 0]
    want = [orc.chain_f32(x[c], orclib.AM, taps[tapsets[c]], taps[tapsets[c]], SIN4, COS4, bq) for c in range(3)]
    for block in blocks:
        chain = msdr.Chain(ctx, msdr.ARITH_F32, 3, taps, taps, mixer=msdr.MIXER_FS4, mode=orclib.AM, tapsets=tapsets, biquad_coeffs=bq)
        got = run_chain(ctx, chain, x, np.float32, block)
        assert chain.info()["kernel"] == name, chain.info()["kernel"]
        for c in range(3):
            assert rel_rms(got[c], want[c]) < TOL, (block, c, rel_rms(got[c], want[c]))


@pytest.mark.parametrize("stages", [0, 1, 2])
@pytest.mark.parametrize("n", [102, 256, 512])
def test_designer_taps(ctx, orc, n, stages):
    """the reference's AM filter at its own length (102) and at BASELINE.json's 256 / 512 taps, two bandwidths of the menu as two tap sets"""
    check(ctx, orc, [designer(n, 2800.0), designer(n, 1400.0)], stages, seed=n + stages)


@pytest.mark.parametrize("n,centre", [(2, 0), (3, 1), (20, 9), (33, 16), (62, 31), (64, 32), (100, 49), (129, 64), (191, 95), (193, 96), (195, 97),
                                      (256, 100), (257, 128), (320, 159), (321, 160), (323, 161), (400, 200), (512, 255), (512, 300), (513, 256)])
def test_every_centre_class(ctx, orc, n, centre):
    """every step count and both parities of the centre delay D0 = n - 1 - centre (the up- and the down-walking reads start D0 + 32 elements apart:
    every residue of that distance mod 8, both copies of the window), filters whose mirror image is cut off by the array's end"""
    check(ctx, orc, [centred(n, centre, 7 * n + centre)], 1, blocks=(None, 1531), n=5000, seed=n)


def test_not_linear_phase_falls_back(ctx, orc):
    """half-sample centre (even length, the numpy design of the other tests), one asymmetric tap, the designer's unmatched c[0] (+-1 at 102 taps
    for about half of the menu's bandwidths, e.g. 1500 Hz): the other kernels run"""
    k = np.arange(256) - 127.5
    lp = (np.sinc(2 * 2800 / 24000 * k) * np.kaiser(256, 7.0)).astype(np.float32)
    rng = np.random.default_rng(3)
    x = rng.integers(-12000, 12001, (2, 4000)).astype(np.int16)
    bq = _f32_biquads(orc, 1)
    bad = designer(256)
    bad[7] += np.float32(1e-6)
    odd_one = designer(102, 1500.0)
    assert odd_one[0] != 0
    for taps in (lp, bad, odd_one):
        chain = msdr.Chain(ctx, msdr.ARITH_F32, 2, taps, taps, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
        got = run_chain(ctx, chain, x, np.float32)
        assert "amsy" not in chain.info()["kernel"]
        for c in range(2):
            assert rel_rms(got[c], orc.chain_f32(x[c], orclib.AM, taps, taps, SIN4, COS4, bq)) < TOL


def test_mixed_modes_and_time_segments(ctx, orc):
    """SSB channels of the same chain stay on the wave-stream kernel (two launches); a long block is cut into time segments with a settled cascade"""
    rng = np.random.default_rng(41)
    lp = designer(256)
    bq = _f32_biquads(orc, 2)
    modes = [orclib.AM, orclib.LSB, orclib.AM, orclib.USB]
    x = rng.integers(-12000, 12001, (4, 6000)).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 4, lp, lp, mixer=msdr.MIXER_FS4, modes=modes, biquad_coeffs=bq)
    got = run_chain(ctx, chain, x, np.float32, 2500)
    assert chain.info()["kernel"] == "chain_mfw_kernel + chain_amsy_kernel"
    for c in range(4):
        assert rel_rms(got[c], orc.chain_f32(x[c], modes[c], lp, lp, SIN4, COS4, bq)) < TOL, c
    n = 1 << 19
    x = rng.integers(-12000, 12001, (1, n)).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 1, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
    got = run_chain(ctx, chain, x, np.float32)
    info = chain.info()
    assert info["kernel"] == "chain_amsy_kernel" and info["time_segments"] > 1
    want = orc.chain_f32(x[0], orclib.AM, lp, lp, SIN4, COS4, bq)
    assert rel_rms(got[0], want) < TOL
    seg = -(-n // info["time_segments"])
    seg = -(-seg // info["tile"]) * info["tile"]
    for sgi in range(1, info["time_segments"]):
        lo = sgi * seg
        if lo + 64 <= n:
            assert rel_rms(got[0, lo:lo + 64], want[lo:lo + 64]) < TOL, sgi


def test_same_results_as_the_other_kernels(ctx, orc, monkeypatch):
    """MSDR_AMSY=0 keeps the Toeplitz kernels: both evaluations of one chain agree far inside the tolerance"""
    rng = np.random.default_rng(5)
    lp = designer(256)
    bq = _f32_biquads(orc, 2)
    x = rng.integers(-20000, 20001, (2, 20000)).astype(np.int16)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MSDR_AMSY", flag)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, 2, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
        outs[flag] = (run_chain(ctx, chain, x, np.float32, 7000), chain.info()["kernel"])
    assert outs["1"][1] == "chain_amsy_kernel" and "amsy" not in outs["0"][1]
    for c in range(2):
        assert rel_rms(outs["1"][0][c], outs["0"][0][c]) < 2e-6
