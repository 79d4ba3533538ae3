"""GPU parity of the LMS automatic notch / noise reduction through the C ABI: bit-exact against the oracle,
as a stage and inside the fused Q15 chain (msdr_chain_set_anr)."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr  # noqa: F401
from test_oracle_anr import signal

pytestmark = pytest.mark.gpu
B = 128
CORR = orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0


@pytest.fixture(scope="module")
def orc():
    return orclib.Oracle()


@pytest.mark.parametrize("ch,n,calls", [(1, 3000, 1), (70, 640, 3), (64, 128, 4)])
def test_anr_stage_matches_oracle(ctx, orc, ch, n, calls):
    rng = np.random.default_rng(ch + n)
    x = np.stack([signal(rng, n * calls, tone=rng.uniform(300, 3000), level=rng.uniform(200, 12000)) for _ in range(ch)])
    x[0, 200:210], x[0, 210:220] = 32767, -32768
    modes = np.array([(c % 3) for c in range(ch)], np.int32)          # off / notch / noise reduction per channel
    modes[0] = 1
    dm = ctx.to_device(modes)
    anr = msdr.Anr(ctx, ch)
    got = np.empty_like(x)
    for k in range(calls):
        d = ctx.to_device(np.ascontiguousarray(x[:, k * n:(k + 1) * n]))
        anr.process(d, n, d_anr_on=dm)
        got[:, k * n:(k + 1) * n] = d.download()
    for c in range(ch):
        a = orc.anr_new()
        want = orc.anr_q15(a, modes[c], x[c])
        assert np.array_equal(got[c], want), (c, modes[c])
        if c in (0, ch - 1) and modes[c]:
            st = anr.state(c)
            assert st[0] == np.float32(a.lidx) and st[1] == np.float32(a.ngamma) and st[2:3].view(np.int32)[0] == a.in_idx
            assert np.array_equal(st[4:68], np.array(a.w[:64], np.float32))
            ring = np.array(a.d[:], np.float32)
            recent = [(a.in_idx + 1 + k) & 511 for k in range(80)]                # the entries still to be read
            assert all(st[68 + (k & 127)] == ring[k] for k in recent)


def test_anr_in_the_q15_chain(ctx, orc):
    rng = np.random.default_rng(21)
    ch, n = 4, 12 * B
    t = np.arange(n)
    x = np.stack([(9000 * (1 + 0.5 * np.sin(2 * np.pi * 400 * t / 24000)) * np.cos(2 * np.pi * 6000 * t / 24000 + c)
                   + 1500 * np.cos(2 * np.pi * 7000 * t / 24000) + rng.integers(-100, 101, n)).astype(np.int16) for c in range(ch)])
    taps = msdr.calc_fir_coeffs(102, 2800)[:102]
    lp = msdr.biquad_design(msdr.BQ_LOWPASS, np.float32(6000 * 0.9 * CORR), 0.54)
    anr_on = np.array([1, 0, 2, 1], np.int32)
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps, taps, mode=orclib.AM, biquad_nodes=[[lp]])
    chain.set_anr(anr_on)
    got = np.empty((ch, n), np.int16)
    for b0 in range(0, n, 4 * B):
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, b0:b0 + 4 * B])), ctx.array((ch, 4 * B), np.int16)
        chain.process(dx, dy, 4 * B)
        got[:, b0:b0 + 4 * B] = dy.download()
    for c in range(ch):
        audio = orc.chain_q15(x[c], orclib.AM, taps, taps)                 # demodulated, before any biquad
        audio = orc.anr_q15(orc.anr_new(), anr_on[c], audio)
        want = orc.biquad_teensy_update(orc.biquad_teensy_new([lp]), audio)
        assert np.array_equal(got[c], want), c
    # (an fp32 chain takes the filter too since round 2 -- its float flavour: tests/test_gpu_chain_post.py)
    msdr.Chain(ctx, msdr.ARITH_F32, 1, taps.astype(np.float32), taps.astype(np.float32)).set_anr(None, 1)
