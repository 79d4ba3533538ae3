"""init_FIR() (Minimal-SDR.ino:901-930) on the GPU: msdr_chain_init_fir zeroes the FIR state and nothing else -- the biquad /
cascade state, the SYNCAM PLL statics and the LMS filter persist across it, as they do in the reference; msdr_chain_reset is the
full stream restart."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr, rel_rms  # noqa: F401
from test_gpu_chain import _f32_biquads, _hilbert_pair, _ref_nodes, run_chain

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("mode", [orclib.LSB, orclib.AM])
@pytest.mark.parametrize("stages", [1, 2, 4])
def test_init_fir_f32_keeps_cascade_state(ctx, orc, mode, stages):
    rng = np.random.default_rng(100 + stages)
    hi, hq = _hilbert_pair(100)
    bq = _f32_biquads(orc, stages)
    cos4, sin4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)
    x = rng.integers(-8000, 8001, (3, 3 * 1300)).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 3, hi, hq, mixer=msdr.MIXER_FS4, mode=mode, biquad_coeffs=bq)
    st = [{} for _ in range(3)]
    for k in range(3):
        if k:
            chain.init_fir()
            if k == 2:
                chain.init_fir()                 # twice before the next sample: still the same stream
            for s in st:
                s["hist_i"][:] = 0
                s["hist_q"][:] = 0
        seg = x[:, 1300 * k:1300 * (k + 1)]
        got = run_chain(ctx, chain, seg, np.float32)
        for c in range(3):
            want = orc.chain_f32(seg[c], mode, hi, hq, sin4, cos4, bq, state=st[c])
            assert rel_rms(got[c], want) < TOL, (k, c, rel_rms(got[c], want))
    # init_fir followed by a retune before any sample, then the full restart
    chain.init_fir()
    chain.set_mode(1, orclib.USB, 0)
    for s in st:
        s["hist_i"][:] = 0
        s["hist_q"][:] = 0
    seg = x[:, :1300]
    got = run_chain(ctx, chain, seg, np.float32)
    for c, m in enumerate([mode, orclib.USB, mode]):
        want = orc.chain_f32(seg[c], m, hi, hq, sin4, cos4, bq, state=st[c])
        assert rel_rms(got[c], want) < TOL, (c, rel_rms(got[c], want))
    chain.reset()
    got = run_chain(ctx, chain, seg, np.float32)
    for c, m in enumerate([mode, orclib.USB, mode]):
        assert rel_rms(got[c], orc.chain_f32(seg[c], m, hi, hq, sin4, cos4, bq)) < TOL


@pytest.mark.parametrize("mode", [orclib.AM, orclib.LSB, orclib.SYNCAM])
def test_init_fir_q15_keeps_biquad_nodes_pll_and_lms(ctx, orc, golden, mode):
    rng = np.random.default_rng(7)
    ti = golden["fir/taps_am102"] if mode != orclib.LSB else golden["taps/FIR_SSB_I_coeffs"]
    tq = golden["fir/taps_am102"] if mode != orclib.LSB else golden["taps/FIR_SSB_Q_coeffs"]
    lp, nt = _ref_nodes(orc)
    ch, nb = 64, 9
    x = rng.integers(-12000, 12001, (ch, 3 * nb * 128)).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, ti, tq, mode=mode, biquad_nodes=[[lp], [nt]],
                       flags=msdr.CHAIN_SYNCAM_PLL if mode == orclib.SYNCAM else 0)
    chain.set_anr(None, 1)
    whole = np.empty_like(x)
    for k in range(3):
        if k:
            chain.init_fir()
        whole[:, k * nb * 128:(k + 1) * nb * 128] = run_chain(ctx, chain, x[:, k * nb * 128:(k + 1) * nb * 128], np.int16)
    for c in (0, 17, 63):
        nodes = [orc.biquad_teensy_new([lp]), orc.biquad_teensy_new([nt])]
        anr, pll = orc.anr_new(), orc.syncam_new()
        for k in range(3):
            seg = x[c, k * nb * 128:(k + 1) * nb * 128]
            # demodulation() with freshly initialised FIR instances; everything behind the FIR pair carries on
            if mode == orclib.SYNCAM:
                _, i_f, q_f = orc.chain_q15(seg, orclib.AM, ti, tq, want_iq=True)
                audio = orc.syncam_q15(pll, i_f, q_f)
            else:
                audio = orc.chain_q15(seg, mode, ti, tq)
            audio = orc.anr_q15(anr, 1, audio)
            for b in range(nb):
                blk = audio[b * 128:(b + 1) * 128].copy()
                for node in nodes:
                    blk = orc.biquad_teensy_update(node, blk)
                assert np.array_equal(whole[c, (k * nb + b) * 128:(k * nb + b + 1) * 128], blk), (c, k, b)
