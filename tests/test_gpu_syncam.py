"""GPU parity of the SYNCAM PLL demodulator through the C ABI: bit-exact against the oracle, as a
stage and inside the fused Q15 chain (MSDR_CHAIN_SYNCAM_PLL)."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr  # noqa: F401

pytestmark = pytest.mark.gpu
B = 128
CORR = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0


@pytest.fixture(scope="module")
def orc():
    return orclib.Oracle()


def _iq(rng, ch, n):
    t = np.arange(n)
    i, q = np.empty((ch, n), np.int16), np.empty((ch, n), np.int16)
    for c in range(ch):
        off = rng.uniform(-300, 300)
        car = 2 * np.pi * off * t / 24000 + rng.uniform(0, 6.28)
        env = rng.uniform(500, 20000) * (1 + 0.6 * np.sin(2 * np.pi * rng.uniform(100, 2000) * t / 24000))
        i[c] = (env * np.cos(car) + rng.integers(-40, 41, n)).clip(-32768, 32767)
        q[c] = (env * np.sin(car) + rng.integers(-40, 41, n)).clip(-32768, 32767)
    return i, q


@pytest.mark.parametrize("ch,n,calls", [(1, 4096, 1), (70, 1000, 3), (64, 128, 5)])
def test_syncam_stage_matches_oracle(ctx, orc, ch, n, calls):
    rng = np.random.default_rng(ch + n)
    i, q = _iq(rng, ch, n * calls)
    i[0, 50:60], q[0, 50:60] = 32767, -32768
    pll = msdr.Syncam(ctx, ch)
    got = np.empty((ch, n * calls), np.int16)
    for k in range(calls):
        di, dq = ctx.to_device(np.ascontiguousarray(i[:, k * n:(k + 1) * n])), ctx.to_device(np.ascontiguousarray(q[:, k * n:(k + 1) * n]))
        pll.process(di, dq, di, n)                        # in place on I
        got[:, k * n:(k + 1) * n] = di.download()
    for c in range(ch):
        s = orc.syncam_new()
        want = orc.syncam_q15(s, i[c], q[c])
        assert np.array_equal(got[c], want), c
        if c in (0, ch - 1):
            assert np.array_equal(pll.state(c), np.array([s.fil_out, s.omega2, s.phzerror], np.float32))


def test_syncam_in_the_q15_chain(ctx, orc):
    """Fused Q15 chain with per-channel modes: SYNCAM channels go FIR pair -> PLL -> biquad nodes, the others are untouched
    by the PLL stage; without the flag SYNCAM demodulates like AM (the Teensy 3.2 build)."""
    rng = np.random.default_rng(12)
    ch, nblk = 6, 20
    n = nblk * B
    t = np.arange(n)
    x = np.empty((ch, n), np.int16)
    for c in range(ch):
        x[c] = (9000 * (1 + 0.5 * np.sin(2 * np.pi * 300 * t / 24000)) * np.cos(2 * np.pi * (6000 + 20 * c) * t / 24000 + c)
                + rng.integers(-100, 101, n)).astype(np.int16)
    taps = msdr.calc_fir_coeffs(102, 2800)[:102]
    lp = msdr.biquad_design(msdr.BQ_LOWPASS, np.float32(6000 * 0.9 * CORR), 0.54)
    modes = np.array([orclib.SYNCAM, orclib.AM, orclib.SYNCAM, orclib.LSB, orclib.SYNCAM, orclib.USB], np.int32)
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps, taps, modes=modes, biquad_nodes=[[lp]], flags=msdr.CHAIN_SYNCAM_PLL)
    got = np.empty((ch, n), np.int16)
    for b0 in range(0, n, 5 * B):                        # several calls: PLL, FIR and biquad state carried
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, b0:b0 + 5 * B])), ctx.array((ch, 5 * B), np.int16)
        chain.process(dx, dy, 5 * B)
        got[:, b0:b0 + 5 * B] = dy.download()
    assert chain.info()["kernel"] == "chain_q15mf_kernel"     # the matrix-core kernel hands the PLL its I and Q
    for c in range(ch):
        if modes[c] == orclib.SYNCAM:
            _, i_f, q_f = orc.chain_q15(x[c], orclib.AM, taps, taps, want_iq=True)
            audio = orc.syncam_q15(orc.syncam_new(), i_f, q_f)
            want = orc.biquad_teensy_update(orc.biquad_teensy_new([lp]), audio)
        else:
            want = orc.chain_q15(x[c], modes[c], taps, taps, biquads=[orc.biquad_teensy_new([lp])])
        assert np.array_equal(got[c], want), c
    plain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps, taps, modes=modes, biquad_nodes=[[lp]])
    dx, dy = ctx.to_device(x), ctx.array((ch, n), np.int16)
    plain.process(dx, dy, n)
    assert np.array_equal(dy.download()[0], orc.chain_q15(x[0], orclib.AM, taps, taps, biquads=[orc.biquad_teensy_new([lp])]))
