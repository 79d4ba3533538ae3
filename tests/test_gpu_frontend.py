"""GPU parity of the front end (DC block, AudioAmplifier, AGC) through the C ABI: bit-exact
against the oracle, including the state records."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr  # noqa: F401

pytestmark = pytest.mark.gpu
B = 128


@pytest.fixture(scope="module")
def orc():
    return orclib.Oracle()


def _adc(rng, ch, nblk, level):
    n = nblk * B
    t = np.arange(n)
    x = np.empty((ch, n), np.uint16)
    for c in range(ch):
        lv = level * (0.2 + 1.8 * rng.random())
        x[c] = (32768 + rng.integers(-3000, 3000) + lv * (0.6 + 0.4 * np.sin(2 * np.pi * t / (3000 + 97 * c)))
                * np.cos(2 * np.pi * 6000 * t / 24000 + c) + rng.integers(-60, 61, n)).clip(0, 65535).astype(np.uint16)
    return x


def _check_state(fe, f_orc, c):
    st = fe.state(c)
    assert st[0] == f_orc.dc.hpf_y1 and st[1] == f_orc.dc.hpf_x1
    assert st[2] == f_orc.agc.multiplier and st[3] == f_orc.agc.agc_idx
    assert st[4] == np.float32(f_orc.agc.AGC_val).view(np.int32) and st[5] == f_orc.agc.AGC_on
    buf = st[6:19].view(np.int16)[:25]
    assert list(buf) == list(f_orc.agc.agc_buffer)


@pytest.mark.parametrize("ch,call_blocks", [(1, 40), (64, 7), (70, 1), (130, 13), (128, 3), (192, 80), (64, 1), (64, 2), (16, 5), (48, 2), (80, 3)])   # multiples of 16 channels: the slab pipeline
@pytest.mark.parametrize("level", [400, 9000, 31000])
def test_frontend_matches_oracle(ctx, orc, ch, call_blocks, level):
    """Whole front end over many calls (state carried in HBM), channel counts around the 64-lane workgroup, levels that
    make the AGC climb, fall and hit the amplifier's saturation."""
    rng = np.random.default_rng(ch * 1000 + level)
    nblk = 80
    x = _adc(rng, ch, nblk, level)
    x[0, 500:540] = 65535
    x[0, 540:600] = 0
    fe = msdr.Frontend(ctx, ch)
    fe.prime(x[:, 0])
    got = np.empty((ch, nblk * B), np.int16)
    for b0 in range(0, nblk, call_blocks):
        m = min(call_blocks, nblk - b0) * B
        seg = np.ascontiguousarray(x[:, b0 * B:b0 * B + m])
        dx, dy = ctx.to_device(seg), ctx.array((ch, m), np.int16)
        fe.update(dx, dy, m)
        got[:, b0 * B:b0 * B + m] = dy.download()
    for c in range(ch):
        f = orc.frontend_new(first_conversion=int(x[c, 0]))
        want = orc.frontend_run(f, x[c])
        assert np.array_equal(got[c], want), c
        if c in (0, ch - 1):
            _check_state(fe, f, c)


@pytest.mark.parametrize("per_group", [16, 32, 64])
@pytest.mark.parametrize("call_blocks", [1, 2, 9])
def test_frontend_slab_pipeline_channels_per_workgroup(ctx, orc, per_group, call_blocks, monkeypatch):
    """frontend_pipe4_kernel<CH>: the host picks 16, 32 or 64 channels per workgroup from the batch size (read at create time);
    every instantiation against the oracle, state records included, over calls of one, two and nine blocks."""
    monkeypatch.setenv("MSDR_FRONTEND_PIPE_CH", str(per_group))
    ch, nblk = 128, 27
    rng = np.random.default_rng(per_group * 10 + call_blocks)
    x = _adc(rng, ch, nblk, 14000)
    x[5, 300:340] = 65535
    x[5, 340:420] = 0
    fe = msdr.Frontend(ctx, ch)
    fe.prime(x[:, 0])
    got = np.empty((ch, nblk * B), np.int16)
    for b0 in range(0, nblk, call_blocks):
        m = min(call_blocks, nblk - b0) * B
        dx, dy = ctx.to_device(np.ascontiguousarray(x[:, b0 * B:b0 * B + m])), ctx.array((ch, m), np.int16)
        fe.update(dx, dy, m)
        got[:, b0 * B:b0 * B + m] = dy.download()
    for c in range(ch):
        f = orc.frontend_new(first_conversion=int(x[c, 0]))
        want = orc.frontend_run(f, x[c])
        assert np.array_equal(got[c], want), c
        if c % 17 == 0:
            _check_state(fe, f, c)


def test_frontend_in_place_on_the_slab_pipeline(ctx, orc):
    """in == out (the ADC words are overwritten by the conditioned samples, as the reference's node does with its block): the slab
    pipeline loads slab t + 2 -- and the sample in front of it -- before slab t - 1 is stored"""
    rng = np.random.default_rng(5)
    ch, nblk = 64, 11
    x = _adc(rng, ch, nblk, 9000)
    fe = msdr.Frontend(ctx, ch)
    fe.prime(x[:, 0])
    d = ctx.to_device(x.view(np.int16))
    fe.update(d, d, nblk * B)
    got = d.download()
    for c in (0, 17, 63):
        f = orc.frontend_new(first_conversion=int(x[c, 0]))
        assert np.array_equal(got[c], orc.frontend_run(f, x[c])), c


def test_frontend_stage_selection_and_controls(ctx, orc):
    rng = np.random.default_rng(5)
    ch, nblk = 9, 30
    x = _adc(rng, ch, nblk, 5000)
    n = nblk * B
    # DC block alone
    fe = msdr.Frontend(ctx, ch)
    fe.prime(np.uint16(1234))
    dx, dy = ctx.to_device(x), ctx.array((ch, n), np.int16)
    fe.update(dx, dy, n, msdr.FE_DCBLOCK)
    dc = dy.download()
    for c in range(ch):
        st = orclib.DcBlock(0, 1234 << 14)
        assert np.array_equal(dc[c], orc.dcblock(st, x[c]))
    # fixed gain, AGC off (amp_adc.gain(n) with AGC_on = 0), in place on int16 data
    fe2 = msdr.Frontend(ctx, ch)
    fe2.set_agc(False)
    fe2.gain(3.7)
    d = ctx.to_device(dc)
    fe2.update(d, d, n, msdr.FE_AMP | msdr.FE_AGC)
    amp = d.download()
    mult = orc.amp_multiplier(3.7)
    assert msdr.amp_multiplier(3.7) == mult
    for c in range(ch):
        assert np.array_equal(amp[c], orc.amp_update(mult, dc[c]))
    assert fe2.state(0)[2] == mult and fe2.state(0)[3] == 25          # AGC() returned at once: buffer index untouched
    # zero gain: AudioAmplifier transmits nothing -> zeros, AGC not run
    fe3 = msdr.Frontend(ctx, ch)
    fe3.gain(0.0)
    d = ctx.to_device(dc)
    fe3.update(d, d, n, msdr.FE_AMP | msdr.FE_AGC)
    assert not d.download().any() and fe3.state(0)[3] == 25
    # block cadence is part of the contract
    with pytest.raises(msdr.MsdrError):
        fe3.update(d, d, 100)


def test_amp_q15_stage(ctx, orc):
    rng = np.random.default_rng(8)
    x = rng.integers(-32768, 32768, (5, 3 * B)).astype(np.int16)
    for gain in (0.5, 1.0, 2.5, -1.0, 40.0, 0.0):
        mult = msdr.amp_multiplier(gain)
        d = ctx.to_device(x)
        sent = msdr.amp_q15(ctx, mult, d, 5, 3 * B)
        want = orc.amp_update(mult, x.reshape(-1))
        if want is None:
            assert not sent
        else:
            assert sent and np.array_equal(d.download().reshape(-1), want), gain


def test_dac_format_stage(ctx, orc):
    """output_dac.cpp:139-151: (sample + 32768) >> 4, mid-scale when no block arrived; odd sizes take the scalar tail."""
    rng = np.random.default_rng(9)
    for ch, n in ((3, 128), (5, 131), (1, 7)):
        x = rng.integers(-32768, 32768, (ch, n)).astype(np.int16)
        x.flat[:4] = [-32768, 32767, 0, -1]
        d, o = ctx.to_device(x), ctx.array((ch, n), np.int16)
        msdr.dac_format_q15(ctx, d, o, ch, n)
        want = ((x.astype(np.int32) + 32768) >> 4).astype(np.int16)
        assert np.array_equal(o.download(), want)
        ref = np.empty(x.size, np.int16)
        orc.lib.orc_dac_format(x.ctypes.data_as(orclib.C.c_void_p), ref.ctypes.data_as(orclib.C.c_void_p), orclib.C.c_uint32(x.size))
        assert np.array_equal(ref.reshape(x.shape), want)
        msdr.dac_format_q15(ctx, None, o, ch, n)
        assert (o.download() == 2048).all()
