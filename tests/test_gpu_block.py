"""GPU parity of the fp32 chain at the reference's BLOCK cadence: one AUDIO_BLOCK_SAMPLES = 128 block per call
(Minimal-SDR.ino:518-530, :574-575), served by chain_mfb_kernel (channel-batched tiles, msdr_chain_mfb.hiph).

Every test streams >= 64 consecutive ticks through the C ABI and compares the whole stream with the oracle's sequential fp32 chain
(<= 1e-5 relative RMS per channel, the north-star's tolerance); FIR history and cascade state travel from tick to tick through the
library's ping-pong buffers, written by the block kernel itself."""
import numpy as np
import pytest

import orclib
from gpuhelp import ctx, msdr, rel_rms  # noqa: F401
from test_gpu_chain import _f32_biquads, _hilbert_pair, _truth_f64, run_chain

pytestmark = pytest.mark.gpu
TOL = 1e-5
COS4, SIN4 = np.array([1, 0, -1, 0], np.float32), np.array([0, 1, 0, -1], np.float32)


def _lowpass(n_taps, fc=2800.0):
    k = np.arange(n_taps)
    h = np.sinc(2 * fc / 24000.0 * (k - (n_taps - 1) / 2.0)) * np.kaiser(n_taps, 7.0)
    return (h / h.sum()).astype(np.float32)


def _if(rng, ch, n):
    x = rng.integers(-8000, 8001, (ch, n)).astype(np.int16)
    t = np.arange(n)
    x[0] = np.round(6000 * (0.5 + 0.4 * np.sin(2 * np.pi * 300 * t / 24000)) * np.cos(2 * np.pi * 6000 * t / 24000)).astype(np.int16)
    return x


def _is_block(chain):
    return chain.info()["kernel"].startswith("chain_mfb_kernel")


@pytest.mark.parametrize("stages", [0, 1, 2])
@pytest.mark.parametrize("ch", [37, 64])
def test_block_am_c3_shape_vs_oracle(ctx, orc, stages, ch):
    """c3 in small: AM channels behind the exact Fs/4 mixer, 256-tap low-pass pair, 0..2 biquad sections; 70 ticks of 128.
    37 channels: the last tile has idle channel slots."""
    rng = np.random.default_rng(500 + stages)
    n = 70 * 128
    x = _if(rng, ch, n)
    lp = _lowpass(256)
    bq = _f32_biquads(orc, stages) if stages else None
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
    got = run_chain(ctx, chain, x, np.float32, 128)
    assert _is_block(chain), chain.info()["kernel"]
    for c in range(ch):
        want = orc.chain_f32(x[c], orclib.AM, lp, lp, SIN4, COS4, bq)
        assert rel_rms(got[c], want) < TOL, (c, rel_rms(got[c], want))
        assert rel_rms(got[c, -128:], want[-128:]) < TOL, c        # the LAST tick on its own: nothing drifts over the ticks


@pytest.mark.parametrize("mode", [orclib.LSB, orclib.USB])
@pytest.mark.parametrize("stages", [0, 1, 2])
def test_block_ssb_c4_shape_vs_oracle(ctx, orc, mode, stages):
    """c4 in small: SSB channels, freq_conv tables at fs/4, 100-tap Hilbert pair (numerator and all-pole response folded into the taps)."""
    rng = np.random.default_rng(510 + stages)
    ch, n = 21, 66 * 128
    x = _if(rng, ch, n)
    hi, hq = _hilbert_pair(100)
    k = np.arange(128)       # freq_conv's q15 tables at fs/4, converted as arm_q15_to_float does: exactly period 4 (bench.py workload())
    oi = (np.round(32767 * np.sin(2 * np.pi * k / 4)).astype(np.int16) / 32768.0).astype(np.float32)
    oq = (np.round(32767 * np.cos(2 * np.pi * k / 4)).astype(np.int16) / 32768.0).astype(np.float32)
    bq = _f32_biquads(orc, stages) if stages else None
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, hi, hq, mixer=msdr.MIXER_NCO, mode=mode, osc_i=oi, osc_q=oq, biquad_coeffs=bq)
    got = run_chain(ctx, chain, x, np.float32, 128)
    assert _is_block(chain), chain.info()["kernel"]
    for c in range(ch):
        want = orc.chain_f32(x[c], mode, hi, hq, oi, oq, bq)
        err = rel_rms(got[c], want)
        if err >= TOL:     # a tone on the suppressed sideband: judge both against float64 (tests/test_gpu_chain.py:test_chain_f32_nco_vs_oracle)
            truth = _truth_f64(x[c], mode, hi, hq, oi, oq, bq)
            assert rel_rms(got[c], truth) <= max(TOL, 1.5 * rel_rms(want, truth)), (c, err)
            continue
        assert err < TOL, (c, err)


@pytest.mark.parametrize("block", [32, 64, 256, 512])
def test_block_other_block_lengths(ctx, orc, block):
    """Blocks of 32 .. 512 samples take the same kernel (1 .. 16 tile rows per channel)."""
    rng = np.random.default_rng(520 + block)
    ch, n = 19, 64 * block
    x = _if(rng, ch, n)
    lp, (hi, hq) = _lowpass(128), _hilbert_pair(128)
    modes = np.array([orclib.AM if c % 3 else orclib.LSB for c in range(ch)], np.int32)
    tapsets = np.array([0 if m == orclib.AM else 1 for m in modes], np.int32)
    bq = _f32_biquads(orc, 2)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, [lp, hi], [lp, hq], mixer=msdr.MIXER_FS4, modes=modes, tapsets=tapsets, biquad_coeffs=bq)
    got = run_chain(ctx, chain, x, np.float32, block)
    assert _is_block(chain), chain.info()["kernel"]
    for c in range(ch):
        want = orc.chain_f32(x[c], modes[c], [lp, hi][tapsets[c]], [lp, hq][tapsets[c]], SIN4, COS4, bq)
        assert rel_rms(got[c], want) < TOL, (block, c, rel_rms(got[c], want))


def test_block_mixed_modes_c5_shape(ctx, orc):
    """c5 in small at block cadence: per-channel AM / LSB / USB / CW with per-mode 512-tap sets: two launches (SSB tables, envelope tables)."""
    rng = np.random.default_rng(530)
    ch, n = 45, 64 * 128
    x = _if(rng, ch, n)
    lp, (hi, hq) = _lowpass(512), _hilbert_pair(512)
    kinds = [orclib.AM, orclib.LSB, orclib.USB, orclib.CW]
    modes = np.array([kinds[(c * 2654435761 >> 5) & 3] for c in range(ch)], np.int32)
    tapsets = np.array([0 if m in (orclib.AM, orclib.CW) else 1 for m in modes], np.int32)
    bq = _f32_biquads(orc, 2)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, [lp, hi], [lp, hq], mixer=msdr.MIXER_FS4, modes=modes, tapsets=tapsets, biquad_coeffs=bq)
    got = run_chain(ctx, chain, x, np.float32, 128)
    assert _is_block(chain), chain.info()["kernel"]
    for c in range(ch):
        want = orc.chain_f32(x[c], modes[c], [lp, hi][tapsets[c]], [lp, hq][tapsets[c]], SIN4, COS4, bq)
        assert rel_rms(got[c], want) < TOL, (c, modes[c], rel_rms(got[c], want))


def test_block_and_stream_calls_alternate_on_one_chain(ctx, orc):
    """Call lengths change from call to call: 128-sample ticks (block kernel), long calls (wave-stream kernel), ragged calls (cold
    tiles) -- history and cascade state are one format, every hand-over is exact to the tolerance."""
    rng = np.random.default_rng(540)
    ch = 11
    plan = [128] * 5 + [4096] + [128] * 3 + [100, 28] + [128] * 4 + [3000, 200] + [64] * 6 + [128] * 40
    n = sum(plan)
    x = _if(rng, ch, n)
    for mode, (ti, tq) in ((orclib.AM, (_lowpass(256),) * 2), (orclib.LSB, _hilbert_pair(100))):
        bq = _f32_biquads(orc, 2)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, ti, tq, mixer=msdr.MIXER_FS4, mode=mode, biquad_coeffs=bq)
        got = np.empty((ch, n), np.float32)
        o, kernels = 0, set()
        for m in plan:
            dx, dy = ctx.to_device(x[:, o:o + m]), ctx.array((ch, m), np.float32)
            chain.process(dx, dy, m)
            got[:, o:o + m] = dy.download()
            kernels.add(chain.info()["kernel"].split("<")[0])
            o += m
        assert "chain_mfb_kernel" in kernels and len(kernels) >= 2, kernels
        for c in range(ch):
            want = orc.chain_f32(x[c], mode, ti, tq, SIN4, COS4, bq)
            assert rel_rms(got[c], want) < TOL, (mode, c, rel_rms(got[c], want))
            o = 0
            for m in plan:                                           # every call on its own
                if m >= 64:
                    assert rel_rms(got[c, o:o + m], want[o:o + m]) < 2 * TOL, (mode, c, o, m)
                o += m


def test_block_kernel_agrees_with_wave_stream_kernel(ctx, orc, monkeypatch):
    """The same stream through the wave-stream kernel (MSDR_NO_BLOCK=1 at create time) and through the block kernel: the two evaluate
    the same split-fp16 products in the same order per row; they differ by the cascade's rounding only."""
    rng = np.random.default_rng(550)
    ch, n = 24, 64 * 128
    x = _if(rng, ch, n)
    lp = _lowpass(256)
    bq = _f32_biquads(orc, 2)
    a = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
    monkeypatch.setenv("MSDR_NO_BLOCK", "1")
    b = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
    monkeypatch.delenv("MSDR_NO_BLOCK")
    ga, gb = run_chain(ctx, a, x, np.float32, 128), run_chain(ctx, b, x, np.float32, 128)
    assert _is_block(a) and not _is_block(b), (a.info()["kernel"], b.info()["kernel"])
    for c in range(ch):
        assert rel_rms(ga[c], gb[c]) < 2e-6, (c, rel_rms(ga[c], gb[c]))


@pytest.mark.parametrize("mode", [orclib.AM, orclib.USB])
def test_block_tile_fill_does_not_change_a_bit(ctx, orc, monkeypatch, mode):
    """Small batches place fewer channels per tile than a tile holds (so that every SIMD gets a wave); where a channel sits in a tile
    changes no arithmetic: every fill gives the same bits, the fill the host picks by itself included."""
    rng = np.random.default_rng(555)
    ch, n = 53, 24 * 128
    x = _if(rng, ch, n)
    ti, tq = ((_lowpass(256),) * 2) if mode == orclib.AM else _hilbert_pair(256)
    bq = _f32_biquads(orc, 2)
    outs = []
    for fill in (None, "1", "2", "4", "8"):
        if fill is None:
            monkeypatch.delenv("MSDR_MB_FILL", raising=False)
        else:
            monkeypatch.setenv("MSDR_MB_FILL", fill)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, ti, tq, mixer=msdr.MIXER_FS4, mode=mode, biquad_coeffs=bq)
        outs.append(run_chain(ctx, chain, x, np.float32, 128))
        assert _is_block(chain), chain.info()["kernel"]
    monkeypatch.delenv("MSDR_MB_FILL", raising=False)
    for o in outs[1:]:
        assert np.array_equal(o.view(np.uint32), outs[0].view(np.uint32))
    for c in range(0, ch, 7):
        want = orc.chain_f32(x[c], mode, ti, tq, SIN4, COS4, bq)
        assert rel_rms(outs[0][c], want) < TOL


def test_block_int16_audio_out(ctx, orc):
    """MSDR_CHAIN_OUT_I16 at block cadence: the play queue's sample type written by the block kernel's store phase."""
    rng = np.random.default_rng(560)
    ch, n = 16, 64 * 128
    x = _if(rng, ch, n)
    lp = _lowpass(102)
    bq = _f32_biquads(orc, 2)
    f = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
    q = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq, flags=msdr.CHAIN_OUT_I16)
    gf, gq = run_chain(ctx, f, x, np.float32, 128), run_chain(ctx, q, x, np.int16, 128)
    assert _is_block(f) and _is_block(q)
    want = np.clip(np.trunc((gf * np.float32(32768.0)).astype(np.float64)), -32768, 32767).astype(np.int16)      # arm_float_to_q15
    assert np.array_equal(gq, want)


def test_block_live_updates_between_ticks(ctx, orc):
    """What the reference changes while the stream runs, between two 128-sample ticks: the bandwidth menu rewrites the AM taps
    (UI.cpp:332-345), tune() switches a channel's mode, the cascade is re-programmed -- state kept across every change."""
    rng = np.random.default_rng(570)
    ch, ticks = 13, 96
    n = ticks * 128
    x = _if(rng, ch, n)
    lp_a, lp_b = _lowpass(102, 2800.0), _lowpass(102, 1500.0)
    hi, hq = _hilbert_pair(102)
    bq_a, bq_b = _f32_biquads(orc, 2), _f32_biquads(orc, 2).copy()
    bq_b[0] = _f32_biquads(orc, 4)[2]                       # another low-pass in section 0
    modes = np.array([orclib.AM if c % 2 else orclib.LSB for c in range(ch)], np.int32)
    tapsets = np.array([0 if m == orclib.AM else 1 for m in modes], np.int32)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, [lp_a, hi], [lp_a, hq], mixer=msdr.MIXER_FS4, modes=modes, tapsets=tapsets, biquad_coeffs=bq_a)
    events = {20: "taps", 40: "mode", 60: "cascade", 80: "taps_back"}
    got = np.empty((ch, n), np.float32)
    used = set()
    for t in range(ticks):
        ev = events.get(t)
        if ev == "taps":
            chain.set_taps(0, lp_b, lp_b)
        elif ev == "taps_back":
            chain.set_taps(0, lp_a, lp_a)
        elif ev == "mode":
            chain.set_mode(2, orclib.USB, 1)                 # an LSB channel turns USB
            chain.set_mode(3, orclib.LSB, 1)                 # an AM channel turns LSB (another table part)
        elif ev == "cascade":
            chain.set_biquad_coeffs(bq_b)
        dx, dy = ctx.to_device(x[:, t * 128:(t + 1) * 128]), ctx.array((ch, 128), np.float32)
        chain.process(dx, dy, 128)
        got[:, t * 128:(t + 1) * 128] = dy.download()
        used.add(chain.info()["kernel"].split("<")[0])
    assert "chain_mfb_kernel" in used, used
    for c in range(ch):
        st, want = {}, np.empty(n, np.float32)
        mode, ts = int(modes[c]), int(tapsets[c])
        sets_i, sets_q, bq = [lp_a, hi], [lp_a, hq], bq_a
        bounds = sorted(events) + [ticks]
        lo = 0
        for b in bounds:
            want[lo * 128:b * 128] = orc.chain_f32(x[c, lo * 128:b * 128], mode, sets_i[ts], sets_q[ts], SIN4, COS4, bq, state=st)
            ev = events.get(b)
            if ev == "taps":
                sets_i, sets_q = [lp_b, hi], [lp_b, hq]
            elif ev == "taps_back":
                sets_i, sets_q = [lp_a, hi], [lp_a, hq]
            elif ev == "mode":
                if c == 2:
                    mode, ts = orclib.USB, 1
                if c == 3:
                    mode, ts = orclib.LSB, 1
            elif ev == "cascade":
                bq = bq_b                                      # (CMSIS semantics: the cascade carries on from its pState under the new coefficients)
            lo = b
        lo = 0
        for b in bounds:                                       # every stretch between two changes on its own
            if True:
                assert rel_rms(got[c, lo * 128:b * 128], want[lo * 128:b * 128]) < 2 * TOL, (c, lo, rel_rms(got[c, lo * 128:b * 128], want[lo * 128:b * 128]))
            lo = b
        assert rel_rms(got[c], want) < 2 * TOL, (c, rel_rms(got[c], want))


def test_block_large_batch_many_workgroups(ctx, orc):
    """4096 channels x 128 (the headline shape at the reference's cadence): 512 tiles, two waves per workgroup; spot-checked channels."""
    rng = np.random.default_rng(580)
    ch, ticks = 4096, 8
    x = rng.integers(-8000, 8001, (ch, ticks * 128)).astype(np.int16)
    lp = _lowpass(256)
    bq = _f32_biquads(orc, 2)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
    got = run_chain(ctx, chain, x, np.float32, 128)
    info = chain.info()
    assert _is_block(chain) and info["grid"] >= 128, info
    for c in list(range(0, ch, 97)) + [ch - 1]:
        want = orc.chain_f32(x[c], orclib.AM, lp, lp, SIN4, COS4, bq)
        assert rel_rms(got[c], want) < TOL, (c, rel_rms(got[c], want))


# ------------------------------------------------------------------------------------------------------------------------------------
# the reference AS WRITTEN at its own cadence: chain_q15mb_kernel (msdr_chain_q15mb.hiph) + the Teensy biquad nodes, bit-exact
# ------------------------------------------------------------------------------------------------------------------------------------
from test_gpu_chain import _ref_nodes  # noqa: E402


def _is_qblock(chain):
    return chain.info()["kernel"].startswith("chain_q15mb_kernel")


@pytest.mark.parametrize("mixer", [0, 1])
@pytest.mark.parametrize("sqrt_kind", [0, 1])
def test_block_q15_reference_graph_64_ticks_bit_exact(ctx, orc, golden, mixer, sqrt_kind):
    """demodulation() + biquad1_dac + biquad2_dac, one 128-sample block per call for 64 calls, per-channel mode and tap set (the
    reference's own tap tables), retunes between ticks: every sample equal to the oracle's (FIR part pinned to the compiled reference)."""
    rng = np.random.default_rng(600 + 2 * mixer + sqrt_kind)
    ch, ticks = 70, 64
    pad = lambda t: np.concatenate([np.zeros(102 - t.size, np.int16), t])
    sets_i = [golden["fir/taps_am102"], pad(golden["taps/FIR_SSB_I_coeffs"]), pad(golden["taps/FIR_CW_I_coeffs"])]
    sets_q = [golden["fir/taps_am102"], pad(golden["taps/FIR_SSB_Q_coeffs"]), pad(golden["taps/FIR_CW_Q_coeffs"])]
    lp, nt = _ref_nodes(orc)
    oi = oq = None
    if mixer:
        k = np.arange(128)
        oi = np.round(32767 * np.sin(2 * np.pi * k / 4)).astype(np.int16)
        oq = np.round(32767 * np.cos(2 * np.pi * k / 4)).astype(np.int16)
    modes = rng.integers(1, 5, ch).astype(np.int32)
    tapsets = rng.integers(0, 3, ch).astype(np.int32)
    x = rng.integers(-32768, 32768, (ch, ticks * 128)).astype(np.int16)
    x[3] = -32768                                               # the mixer's wrapping negate, the envelope's wrapping sum
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, sets_i, sets_q, mixer=mixer, modes=modes, tapsets=tapsets, osc_i=oi, osc_q=oq,
                       sqrt_kind=sqrt_kind, biquad_nodes=[[lp], [nt]])
    watch = [0, 1, 3, 17, 63, 64, 69]
    states = {c: {} for c in watch}
    got = np.empty((ch, ticks * 128), np.int16)
    marks = {c: [0] for c in watch}                            # tick indices where channel c's mode / tap set changed
    hist = {c: [(int(modes[c]), int(tapsets[c]))] for c in watch}
    for t in range(ticks):
        if t in (16, 33, 50):
            for c in watch[(t % 2)::2] + [5, 40]:
                modes[c], tapsets[c] = int(rng.integers(1, 5)), int(rng.integers(0, 3))
                chain.set_mode(c, int(modes[c]), int(tapsets[c]))
                if c in marks:
                    marks[c].append(t); hist[c].append((int(modes[c]), int(tapsets[c])))
        dx, dy = ctx.to_device(x[:, t * 128:(t + 1) * 128]), ctx.array((ch, 128), np.int16)
        chain.process(dx, dy, 128)
        got[:, t * 128:(t + 1) * 128] = dy.download()
        assert _is_qblock(chain), chain.info()["kernel"]
    for c in watch:
        bounds = marks[c] + [ticks]
        for (lo, hi), (m, ts) in zip(zip(bounds[:-1], bounds[1:]), hist[c]):
            want = orc.chain_q15(x[c, lo * 128:hi * 128], m, sets_i[ts], sets_q[ts], mixer=mixer, osc_i=oi, osc_q=oq, sqrt_kind=sqrt_kind,
                                 biquads=[orc.biquad_teensy_new([lp]), orc.biquad_teensy_new([nt])], state=states[c])
            assert np.array_equal(got[c, lo * 128:hi * 128], want), (c, lo, hi, m, ts)


@pytest.mark.parametrize("ch", [16, 48, 4096])
def test_block_q15_nodes_sub_slab_pipeline_bit_exact(ctx, orc, golden, monkeypatch, ch):
    """biquad1_dac -> biquad2_dac at block cadence on channel counts the sub-slab kernel takes (multiples of 16: biquad_teensy_blk_kernel, the
    two recursions pipelined over 16-sample sub-slabs inside the one 128-sample block): 40 ticks against the oracle, state carried from tick to
    tick, full-scale and -32768 inputs; and the same stream through the slab kernels (MSDR_BIQUAD_BLK=0): identical."""
    rng = np.random.default_rng(640 + ch)
    ticks = 40
    taps = golden["fir/taps_am102"]
    lp, nt = _ref_nodes(orc)
    x = rng.integers(-32768, 32768, (ch, ticks * 128)).astype(np.int16)
    x[3] = -32768
    x[5] = 32767
    outs = []
    monkeypatch.setenv("MSDR_Q15_NO_FUSE", "1")                 # (the node kernels on their own: where the chain kernel takes the nodes in, this test would compare it with itself)
    for force in (None, "0"):
        if force is None:
            monkeypatch.delenv("MSDR_BIQUAD_BLK", raising=False)
        else:
            monkeypatch.setenv("MSDR_BIQUAD_BLK", force)
        chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps, taps, mode=orclib.AM, biquad_nodes=[[lp], [nt]])
        outs.append(run_chain(ctx, chain, x, np.int16, 128))
        assert _is_qblock(chain), chain.info()["kernel"]
    monkeypatch.delenv("MSDR_BIQUAD_BLK", raising=False)
    monkeypatch.delenv("MSDR_Q15_NO_FUSE", raising=False)
    assert np.array_equal(outs[0], outs[1])
    for c in sorted(set([0, 3, 5, 15, ch - 1, ch // 2])):
        want = orc.chain_q15(x[c], orclib.AM, taps, taps, biquads=[orc.biquad_teensy_new([lp]), orc.biquad_teensy_new([nt])])
        assert np.array_equal(outs[0][c], want), c


@pytest.mark.parametrize("nw", ["3", "4", "8"])
@pytest.mark.parametrize("mixed", [False, True])
def test_block_q15_nodes_inside_the_chain_kernel_bit_exact(ctx, orc, golden, monkeypatch, nw, mixed):
    """The two biquad nodes as chain_q15mb_kernel's second phase (one tile per wave on three or more waves: what the host picks for small
    batches; forced here on a small chain by MSDR_MB_NW): 30 ticks with retunes and a node rewrite in between against the oracle, state
    carried from tick to tick; ragged channel counts (idle slots, idle waves); SSB and envelope channels mixed = two launches, each with its
    own node phase; and the same stream with the node kernel behind the chain kernel (MSDR_Q15_NO_FUSE=1): identical."""
    rng = np.random.default_rng(660 + int(nw) + 10 * mixed)
    ch, ticks = 83, 30
    pad = lambda t: np.concatenate([np.zeros(102 - t.size, np.int16), t])
    sets_i = [golden["fir/taps_am102"], pad(golden["taps/FIR_SSB_I_coeffs"])]
    sets_q = [golden["fir/taps_am102"], pad(golden["taps/FIR_SSB_Q_coeffs"])]
    lp, nt = _ref_nodes(orc)
    modes0 = (rng.integers(1, 5, ch) if mixed else np.full(ch, orclib.AM)).astype(np.int32)
    tapsets0 = (rng.integers(0, 2, ch) if mixed else np.zeros(ch)).astype(np.int32)
    x = rng.integers(-32768, 32768, (ch, ticks * 128)).astype(np.int16)
    x[3] = -32768
    outs = []
    monkeypatch.setenv("MSDR_MB_NW", nw)
    for fuse in (True, False):
        if fuse:
            monkeypatch.delenv("MSDR_Q15_NO_FUSE", raising=False)
        else:
            monkeypatch.setenv("MSDR_Q15_NO_FUSE", "1")
        modes, tapsets = modes0.copy(), tapsets0.copy()
        chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, sets_i, sets_q, modes=modes, tapsets=tapsets, biquad_nodes=[[lp], [nt]])
        got = np.empty((ch, ticks * 128), np.int16)
        r2 = np.random.default_rng(7)
        for t in range(ticks):
            if t in (9, 20):
                for c in (0, 5, 40, 82):
                    modes[c], tapsets[c] = (int(r2.integers(1, 5)), int(r2.integers(0, 2))) if mixed else (orclib.AM, 0)
                    chain.set_mode(c, int(modes[c]), int(tapsets[c]))
            dx, dy = ctx.to_device(x[:, t * 128:(t + 1) * 128]), ctx.array((ch, 128), np.int16)
            chain.process(dx, dy, 128)
            got[:, t * 128:(t + 1) * 128] = dy.download()
            want_name = "chain_q15mb_kernel (block tiles) + both biquad nodes" if fuse else "chain_q15mb_kernel (channel-batched block tiles)"
            assert chain.info()["kernel"] == want_name, chain.info()["kernel"]
        outs.append(got)
    monkeypatch.delenv("MSDR_Q15_NO_FUSE", raising=False)
    monkeypatch.delenv("MSDR_MB_NW", raising=False)
    assert np.array_equal(outs[0], outs[1])
    if not mixed:
        for c in (0, 3, 7, 8, 41, 82):
            want = orc.chain_q15(x[c], orclib.AM, sets_i[0], sets_q[0], biquads=[orc.biquad_teensy_new([lp]), orc.biquad_teensy_new([nt])])
            assert np.array_equal(outs[0][c], want), c


def test_block_q15_matches_reference_golden_and_stream_kernel(ctx, golden, monkeypatch):
    """The reference-generated golden chain vectors at block cadence, and the same through the streaming kernel (MSDR_NO_BLOCK=1)."""
    sigs = ["am", "tones", "noise", "full"]
    x = np.stack([golden["chain/x_" + s] for s in sigs])
    for mn, mode, ti, tq in (("AM", orclib.AM, "fir/taps_am102", "fir/taps_am102"), ("LSB", orclib.LSB, "taps/FIR_SSB_I_coeffs", "taps/FIR_SSB_Q_coeffs"),
                             ("USB", orclib.USB, "taps/FIR_SSB_I_coeffs", "taps/FIR_SSB_Q_coeffs"), ("CW", orclib.CW, "taps/FIR_CW_I_coeffs", "taps/FIR_CW_Q_coeffs")):
        a = msdr.Chain(ctx, msdr.ARITH_Q15, len(sigs), golden[ti], golden[tq], mode=mode)
        monkeypatch.setenv("MSDR_NO_BLOCK", "1")
        b = msdr.Chain(ctx, msdr.ARITH_Q15, len(sigs), golden[ti], golden[tq], mode=mode)
        monkeypatch.delenv("MSDR_NO_BLOCK")
        ga, gb = run_chain(ctx, a, x, np.int16, 128), run_chain(ctx, b, x, np.int16, 128)
        assert _is_qblock(a) and not _is_qblock(b), (a.info()["kernel"], b.info()["kernel"])
        for c, s in enumerate(sigs):
            assert np.array_equal(ga[c], golden["chain/%s_%s_audio" % (s, mn)]), (s, mn)
        assert np.array_equal(ga, gb)


@pytest.mark.parametrize("block", [32, 64, 256, 512])
def test_block_q15_other_block_lengths_and_long_taps(ctx, orc, block):
    """256-tap designer low-pass (c3's filter as the reference designs it) and a 256-tap pair, blocks of 32 .. 512, calls of other
    lengths in between (the streaming kernel takes those: one history format)."""
    rng = np.random.default_rng(620 + block)
    ch = 19
    am = orc.calc_fir_coeffs(256, 2800.0)[:256].copy()
    hi = np.round(_hilbert_pair(256)[0].astype(np.float64) * 32767).astype(np.int16)
    hq = np.round(_hilbert_pair(256)[1].astype(np.float64) * 32767).astype(np.int16)
    modes = np.array([orclib.AM if c % 3 else orclib.USB for c in range(ch)], np.int32)
    tapsets = np.array([0 if m == orclib.AM else 1 for m in modes], np.int32)
    plan = [block] * 20 + [1024 + 128] + [block] * 20 + [128 * 3] + [block] * 24
    plan = [m for m in plan if m % 128 == 0 or m == block]
    n = sum(plan)
    pad = (-n) % 128
    if pad:
        plan.append(pad); n += pad
    x = rng.integers(-20000, 20001, (ch, n)).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, [am, hi], [am, hq], modes=modes, tapsets=tapsets)
    got = np.empty((ch, n), np.int16)
    o, kernels = 0, set()
    for m in plan:
        dx, dy = ctx.to_device(x[:, o:o + m]), ctx.array((ch, m), np.int16)
        chain.process(dx, dy, m)
        got[:, o:o + m] = dy.download()
        kernels.add(chain.info()["kernel"].split(" ")[0].split("<")[0])
        o += m
    assert "chain_q15mb_kernel" in kernels, kernels
    for c in range(ch):
        want = orc.chain_q15(x[c], int(modes[c]), [am, hi][tapsets[c]], [am, hq][tapsets[c]])
        assert np.array_equal(got[c], want), (block, c)


def test_block_q15_live_tap_and_node_updates(ctx, orc):
    """The bandwidth menu rewrites FIR_AM_coeffs in place (UI.cpp:332-345) and tune() re-programs biquad2_dac (Minimal-SDR.ino:356) between
    two 128-sample ticks: FIR state and node history kept, bit-exact."""
    import ctypes as C
    rng = np.random.default_rng(640)
    ch, ticks = 33, 72
    lp, nt = _ref_nodes(orc)
    nt2 = orc.biquad_design(orclib.BQ_NOTCH, np.float32(2950.0 * orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0), 15.0)
    taps = [orc.calc_fir_coeffs(102, float(bw))[:102].copy() for bw in (2800, 1500, 4000)]
    x = rng.integers(-20000, 20001, (ch, ticks * 128)).astype(np.int16)
    chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, taps[0], taps[0], mode=orclib.AM, biquad_nodes=[[lp], [nt]])
    cur = taps[0]
    states = {c: {} for c in range(0, ch, 4)}
    nodes = {c: [orc.biquad_teensy_new([lp]), orc.biquad_teensy_new([nt])] for c in states}
    lo = 0
    got = np.empty((ch, ticks * 128), np.int16)
    events = {18: ("taps", taps[1]), 36: ("node", nt2), 54: ("taps", taps[2])}
    for t in range(ticks + 1):
        ev = events.get(t)
        if ev or t == ticks:
            for c in states:                                   # the stretch up to this change
                want = orc.chain_q15(x[c, lo * 128:t * 128], orclib.AM, cur, cur, biquads=nodes[c], state=states[c])
                assert np.array_equal(got[c, lo * 128:t * 128], want), (c, lo, t)
            lo = t
            if t == ticks:
                break
            if ev[0] == "taps":
                chain.set_taps(0, ev[1], ev[1]); cur = ev[1]
            else:
                chain.set_node_coefficients(1, 0, ev[1])
                for c in states:
                    recs = states[c].get("bq", nodes[c])
                    orc.lib.orc_biquad_teensy_set_coefficients(C.byref(recs[1]), C.c_uint32(0), orclib._ptr(np.ascontiguousarray(ev[1], np.int32)))
                    states[c]["bq"] = recs
        dx, dy = ctx.to_device(x[:, t * 128:(t + 1) * 128]), ctx.array((ch, 128), np.int16)
        chain.process(dx, dy, 128)
        got[:, t * 128:(t + 1) * 128] = dy.download()
        assert _is_qblock(chain), chain.info()["kernel"]


# ------------------------------------------------------------------------------------------------------------------------------------
# msdr_chain_graph_*: an even number of consecutive block-cadence calls as one HIP graph
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arith", ["f32", "q15"])
@pytest.mark.parametrize("ticks_per_graph", [2, 8])
def test_chain_graph_replays_equal_direct_calls(ctx, orc, golden, arith, ticks_per_graph):
    """64+ ticks as replays of one graph (the caller refills the graph's input buffers between replays), with direct calls and a live
    update in between: the audio equals the oracle's over the whole stream -- bit-exact (Q15) / <= 1e-5 (fp32)."""
    rng = np.random.default_rng(700 + ticks_per_graph)
    ch, T = 24, ticks_per_graph
    replays = 64 // T
    lp_a, lp_b = _lowpass(102, 2800.0), _lowpass(102, 1700.0)
    if arith == "q15":
        lpn, ntn = _ref_nodes(orc)
        ta, tb = orc.calc_fir_coeffs(102, 2800.0)[:102].copy(), orc.calc_fir_coeffs(102, 1700.0)[:102].copy()
        chain = msdr.Chain(ctx, msdr.ARITH_Q15, ch, ta, ta, mode=orclib.AM, biquad_nodes=[[lpn], [ntn]])
        out_t = np.int16
    else:
        bq = _f32_biquads(orc, 2)
        chain = msdr.Chain(ctx, msdr.ARITH_F32, ch, lp_a, lp_a, mixer=msdr.MIXER_FS4, mode=orclib.AM, biquad_coeffs=bq)
        out_t = np.float32
    n_total = (replays * T + 4 + T + (replays - 1) * T) * 128    # replays | 3 + 1 direct calls | one replay | a tap change | replays
    x = rng.integers(-12000, 12001, (ch, n_total)).astype(np.int16)
    dxs, dys = [ctx.array((ch, 128), np.int16) for _ in range(T)], [ctx.array((ch, 128), out_t) for _ in range(T)]
    got = np.empty((ch, n_total), out_t)
    o = 0

    def run_graph(g, times):
        nonlocal o
        for _ in range(times):
            for k in range(T):
                dxs[k].upload(x[:, o + 128 * k:o + 128 * (k + 1)])
            g.launch()
            for k in range(T):
                got[:, o + 128 * k:o + 128 * (k + 1)] = dys[k].download()
            o += 128 * T

    def direct(times):
        nonlocal o
        for _ in range(times):
            dx, dy = ctx.to_device(x[:, o:o + 128]), ctx.array((ch, 128), out_t)
            chain.process(dx, dy, 128)
            got[:, o:o + 128] = dy.download()
            o += 128

    g = chain.graph(dxs, dys, 128)
    run_graph(g, replays)
    direct(3)                                                   # an odd number of direct calls: the buffers the graph points at are the other pair now
    with pytest.raises(msdr.MsdrError):
        g.launch()
    direct(1)                                                   # even again: the old graph is valid once more
    run_graph(g, 1)
    change_at = o
    if arith == "q15":
        chain.set_taps(0, tb, tb)
    else:
        chain.set_taps(0, lp_b, lp_b)
    with pytest.raises(msdr.MsdrError):                        # the tables moved: make the graph again
        g.launch()
    g.close()
    g = chain.graph(dxs, dys, 128)
    run_graph(g, replays - 1)
    g.close()
    assert o <= x.shape[1]
    for c in range(0, ch, 5):
        st = {}
        if arith == "q15":
            nodes = [orc.biquad_teensy_new([lpn]), orc.biquad_teensy_new([ntn])]
            w1 = orc.chain_q15(x[c, :change_at], orclib.AM, ta, ta, biquads=nodes, state=st)
            w2 = orc.chain_q15(x[c, change_at:o], orclib.AM, tb, tb, biquads=nodes, state=st)
            assert np.array_equal(got[c, :o], np.concatenate([w1, w2])), c
        else:
            w1 = orc.chain_f32(x[c, :change_at], orclib.AM, lp_a, lp_a, SIN4, COS4, bq, state=st)
            w2 = orc.chain_f32(x[c, change_at:o], orclib.AM, lp_b, lp_b, SIN4, COS4, bq, state=st)
            assert rel_rms(got[c, :change_at], w1) < TOL and rel_rms(got[c, change_at:o], w2) < TOL, c


def test_chain_graph_refuses_what_it_cannot_hold(ctx, orc):
    lp = _lowpass(102)
    chain = msdr.Chain(ctx, msdr.ARITH_F32, 8, lp, lp, mixer=msdr.MIXER_FS4, mode=orclib.AM)
    dx, dy = [ctx.array((8, 128), np.int16) for _ in range(3)], [ctx.array((8, 128), np.float32) for _ in range(3)]
    with pytest.raises(msdr.MsdrError):
        chain.graph(dx, dy, 128)                               # an odd number of calls
    with pytest.raises(msdr.MsdrError):
        chain.graph(dx[:2], dy[:2], 100)                       # not a block-cadence length
    big_x, big_y = [ctx.array((8, 4096), np.int16) for _ in range(2)], [ctx.array((8, 4096), np.float32) for _ in range(2)]
    with pytest.raises(msdr.MsdrError):
        chain.graph(big_x, big_y, 4096)                        # the streaming kernel's calls are not recorded
    g = chain.graph(dx[:2], dy[:2], 128)
    g.launch()
    g.close()
    # a chain left untouched by the refusals: a direct call still works from the state the one replay left
    chain.process(dx[0], dy[0], 128)
