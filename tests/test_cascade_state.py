"""Host side of the live cascade update (msdr_biquad_df1_f32_set_coeffs, msdr_chain_set_biquad_coeffs): the bridge between the
block-parallel kernels' state record and arm_biquad_cascade_df1_f32's pState (csrc/msdr_cascade_state.h), checked against the
oracle's CMSIS-order cascade (oracle/msdr_oracle.c: orc_biquad_df1_f32_run) -- no GPU needed.

What is asserted is the property the live update needs: a stream filtered with coefficients A, whose coefficients are then
rewritten to B with the CMSIS state kept, continues identically when the state is taken through
lib record (A) -> pState -> lib record (B) and the block-parallel form carries on from there."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import orclib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimal-sdr_amd", "python"))
import msdr  # noqa: E402


def _random_cascade(rng, stages, kind="mixed"):
    c = np.zeros((stages, 5))
    for s in range(stages):
        r = rng.uniform(0.3, 0.97)
        th = rng.uniform(0.05, 3.0)
        a1, a2 = 2 * r * np.cos(th), -r * r                  # CMSIS: y = ... + a1 y1 + a2 y2
        if kind == "notch" or (kind == "mixed" and rng.random() < 0.3):
            tz = rng.uniform(0.05, 3.0)
            b = np.array([1.0, -2 * np.cos(tz), 1.0])
        else:
            b = rng.standard_normal(3)
        c[s] = [*(b * rng.uniform(0.2, 1.0)), a1, a2]
    return c.astype(np.float32)


def _df1_f64(c, x, state=None):
    """CMSIS order in float64; state = [x1, x2, y1, y2] per stage.  Returns y and the final state."""
    S = c.shape[0]
    st = np.zeros((S, 4)) if state is None else np.array(state, np.float64).reshape(S, 4)
    y = np.array(x, np.float64)
    for s in range(S):
        b0, b1, b2, a1, a2 = [float(v) for v in c[s]]
        x1, x2, y1, y2 = st[s]
        out = np.empty_like(y)
        for n, xn in enumerate(y):
            acc = b0 * xn + b1 * x1 + b2 * x2 + a1 * y1 + a2 * y2
            x2, x1, y2, y1 = x1, xn, y1, acc
            out[n] = acc
        st[s] = [x1, x2, y1, y2]
        y = out
    return y, st.reshape(-1)


def _lib_f64(c, x, rec=None):
    """Numerators first, all-pole sections afterwards, float64; rec = the 16-entry record.  Returns y and the final record."""
    S = c.shape[0]
    rec = np.zeros(16) if rec is None else np.array(rec, np.float64)
    num = np.array([1.0])
    for s in range(S):
        num = np.convolve(num, c[s, :3].astype(np.float64))
    d = np.concatenate([rec[:2 * S][::-1], np.asarray(x, np.float64)])        # d[-2S] .. d[-1], then the block
    v = np.array([np.dot(num, d[2 * S + n - np.arange(2 * S + 1)]) for n in range(len(x))])
    out = v
    for s in range(S):
        a1, a2 = float(c[s, 3]), float(c[s, 4])
        w1, w2 = rec[8 + 2 * s], rec[9 + 2 * s]
        o = np.empty_like(out)
        for n, u in enumerate(out):
            w = u + a1 * w1 + a2 * w2
            w2, w1 = w1, w
            o[n] = w
        rec[8 + 2 * s], rec[9 + 2 * s] = w1, w2
        out = o
    tail = d[-2 * S:][::-1] if S else np.zeros(0)
    rec[:8] = 0.0
    rec[:2 * S] = tail
    return out, rec


@pytest.mark.parametrize("stages", [1, 2, 3, 4])
def test_lib_record_and_cmsis_state_describe_the_same_stream(stages):
    rng = np.random.default_rng(100 + stages)
    for trial in range(12):
        cA = _random_cascade(rng, stages)
        x = rng.standard_normal(400)
        _, ps = _df1_f64(cA, x)
        _, rec = _lib_f64(cA, x)
        got = msdr.biquad_state_to_cmsis(cA, rec.astype(np.float32))
        scale = np.abs(ps).max()
        assert np.abs(got - ps).max() < 3e-5 * scale, (trial, got, ps)          # (fp32 in / out; the solve itself is long double)
        # ... and back: the record rebuilt from pState continues the stream like the CMSIS form
        rec2 = msdr.biquad_state_from_cmsis(cA, ps.astype(np.float32), d_hist=rec[:2 * stages])
        x2 = rng.standard_normal(200)
        yl, _ = _lib_f64(cA, x2, rec2)
        yd, _ = _df1_f64(cA, x2, ps)
        assert np.sqrt(((yl - yd) ** 2).sum() / (yd ** 2).sum()) < 2e-6, trial


@pytest.mark.parametrize("stages", [1, 2, 3, 4])
@pytest.mark.parametrize("kind", ["mixed", "notch"])
def test_coefficient_change_with_cmsis_state_kept(stages, kind):
    """A -> B under a running stream: lib record (A) -> pState -> lib record (B) against arm_biquad_cascade_df1_f32 with its pState
    untouched and pCoeffs rewritten (float64 models of both forms; the oracle's fp32 cascade is compared below)."""
    rng = np.random.default_rng(7 * stages + (kind == "notch"))
    for trial in range(10):
        cA, cB = _random_cascade(rng, stages, kind), _random_cascade(rng, stages, kind)
        x = rng.standard_normal(500)
        _, psA = _df1_f64(cA, x)
        _, recA = _lib_f64(cA, x)
        ps = msdr.biquad_state_to_cmsis(cA, recA.astype(np.float32))
        recB = msdr.biquad_state_from_cmsis(cB, ps, d_hist=recA[:2 * stages])
        x2 = rng.standard_normal(300)
        yl, _ = _lib_f64(cB, x2, recB)
        yd, _ = _df1_f64(cB, x2, psA)
        err = np.sqrt(((yl - yd) ** 2).sum() / (yd ** 2).sum())
        assert err < 2e-5, (trial, err)


def test_cmsis_state_without_deeper_history_is_enough():
    """Coming from the CMSIS-order kernel only x[n-1], x[n-2] of the input are known: the record built with zeros for the older inputs
    still continues the stream exactly (the section states absorb the difference)."""
    rng = np.random.default_rng(5)
    for stages in (2, 3, 4):
        c = _random_cascade(rng, stages)
        x = rng.standard_normal(300)
        _, ps = _df1_f64(c, x)
        rec = msdr.biquad_state_from_cmsis(c, ps.astype(np.float32))
        assert np.all(rec[2:8] == 0.0)
        x2 = rng.standard_normal(200)
        yl, _ = _lib_f64(c, x2, rec)
        yd, _ = _df1_f64(c, x2, ps)
        assert np.sqrt(((yl - yd) ** 2).sum() / (yd ** 2).sum()) < 2e-6


def test_against_the_oracle_cascade_with_rewritten_pcoeffs(orc):
    """The oracle's orc_biquad_df1_f32_run (fp32, CMSIS order) with pCoeffs rewritten in place between two calls -- what the reference's
    callers may do -- against the float64 lib form started from the bridged state."""
    rng = np.random.default_rng(77)
    lp = orc.biquad_design(orclib.BQ_LOWPASS, 2400 * orclib.AUDIO_SAMPLE_RATE_EXACT / 24000.0, 0.54)
    to_f = lambda q: np.array([q[0], q[1], q[2], -q[3], -q[4]], np.float64) / 2.0 ** 30     # Teensy sign -> CMSIS "added" feedback
    for f_old, f_new in [(3000.0, 2500.0), (1000.0, 5000.0)]:
        nA = orc.biquad_design(orclib.BQ_NOTCH, f_old, 15.0)
        nB = orc.biquad_design(orclib.BQ_NOTCH, f_new, 15.0)
        cA = np.stack([to_f(lp), to_f(nA)]).astype(np.float32)
        cB = np.stack([to_f(lp), to_f(nB)]).astype(np.float32)
        x = rng.standard_normal(600).astype(np.float32)
        x2 = rng.standard_normal(400).astype(np.float32)
        coeffs = cA.reshape(-1).copy()
        st = np.zeros(8, np.float32)
        S = orclib.BiquadDf1()
        orc.lib.orc_biquad_df1_init_f32(C.byref(S), C.c_uint8(2), orclib._ptr(coeffs), orclib._ptr(st))
        y = np.empty_like(x)
        orc.lib.orc_biquad_df1_f32_run(C.byref(S), orclib._ptr(x), orclib._ptr(y), C.c_uint32(x.size))
        _, recA = _lib_f64(cA, x)
        ps = msdr.biquad_state_to_cmsis(cA, recA.astype(np.float32))
        assert np.abs(ps - st).max() < 1e-4 * np.abs(st).max()
        coeffs[:] = cB.reshape(-1)                              # pCoeffs rewritten in place, pState kept
        y2 = np.empty_like(x2)
        orc.lib.orc_biquad_df1_f32_run(C.byref(S), orclib._ptr(x2), orclib._ptr(y2), C.c_uint32(x2.size))
        recB = msdr.biquad_state_from_cmsis(cB, ps, d_hist=recA[:4])
        yl, _ = _lib_f64(cB, x2, recB)
        err = np.sqrt(((yl - y2) ** 2).sum() / (y2.astype(np.float64) ** 2).sum())
        assert err < 1e-5, (f_old, f_new, err)


def test_a_cascade_whose_state_has_no_cmsis_form_is_refused():
    """Stage 2's numerator cancels stage 1's poles: y_1 cannot be recovered from w_1 = B_2 y_1 -- ARGUMENT_ERROR, never a guess."""
    r, th = 0.9, 0.7
    a1, a2 = 2 * r * np.cos(th), -r * r
    c = np.array([[1.0, 0.3, 0.2, a1, a2], [1.0, -a1, -a2, 0.5, -0.2]], np.float32)
    with pytest.raises(msdr.MsdrError) as e:
        msdr.biquad_state_to_cmsis(c, np.ones(16, np.float32))
    assert e.value.status == msdr.STATUS_ARGUMENT_ERROR
