"""Row f1 (front end) oracle against an independent second model written from the reference text
(input_adc.cpp:198-212, mixer.cpp:34-47/134-159, Minimal-SDR.ino:446-515).  These reference files cannot be built
here (un-vendored Teensyduino core, ARM-only intrinsics), so the row is "parity unpinned"; this test at least makes
two independently written restatements agree, including on the quirks (packed int min/max start values, abs() of the
whole word, the dropped 26th buffer store)."""
import numpy as np
import pytest

import orclib


@pytest.fixture(scope="module")
def orc():
    return orclib.Oracle()


COEF = 1048300 << 10


def _dcblock_model(x1, y1, adc):
    out = []
    for v in adc:
        tmp = int(v) << 14
        acc = (y1 - x1 + tmp + 2 ** 31) % 2 ** 32 - 2 ** 31
        y1 = (((acc * COEF) >> 30) + 2 ** 31) % 2 ** 32 - 2 ** 31          # bits [61:30] of the 64-bit product
        x1 = tmp
        out.append(max(-32768, min(32767, y1 >> 14)))
    return x1, y1, np.array(out, np.int16)


def _s16(v):
    v &= 0xFFFF
    return v - 65536 if v >= 32768 else v


def _sel(ca, cb, sa, sb):
    lo = sa if _s16(ca) - _s16(cb) >= 0 else sb
    hi = sa if _s16(ca >> 16) - _s16(cb >> 16) >= 0 else sb
    return (lo & 0xFFFF) | (hi & 0xFFFF0000)


def _i32(v):
    v &= 0xFFFFFFFF
    return v - 2 ** 32 if v >= 2 ** 31 else v


class AgcModel:
    def __init__(self):
        self.buf = [0] * 25
        self.idx = 25
        self.val = np.float32(0.25)
        self.mult = int(np.float32(0.25) * np.float32(65536.0))

    def block(self, blk):
        minv, maxv = 32767, _i32(-32767) & 0xFFFFFFFF
        for i in range(64):
            data = (int(blk[2 * i]) & 0xFFFF) | ((int(blk[2 * i + 1]) & 0xFFFF) << 16)
            maxv = _sel(maxv, data, maxv, data)
            minv = _sel(data, minv, minv, data)
        sh = lambda w: (_i32(w) >> 16) & 0xFFFFFFFF
        maxv = _sel(maxv, sh(maxv), maxv, sh(maxv))
        minv = _sel(sh(minv), minv, minv, sh(minv))
        minv, maxv = abs(_i32(minv)) & 0xFFFFFFFF, abs(_i32(maxv)) & 0xFFFFFFFF
        absmax = _sel(maxv, minv, maxv, minv) & 0xFFFF
        self.idx -= 1
        if self.idx >= 0:
            self.buf[self.idx] = _s16(absmax)
        else:
            self.idx = 25
        m = sum(self.buf)
        d = int(m / 25)                                   # C division truncates toward zero
        f32 = np.float32
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            f = f32(16000) / f32(d)
            v = self.val
            new = None
            if float(f) > 1.3:
                fagc = f32(v + f32(f32(v * f) / f32(1500)))
                if fagc < f32(40.0):
                    new = fagc
            elif float(v) > 0.1:
                for lim, div in ((0.6, 50), (0.7, 200), (0.8, 2000), (0.9, 4000)):
                    if float(f) < lim:
                        new = f32(v - f32(f32(v * f) / f32(div)))
                        break
        if new is not None:
            self.val = new
            n = min(f32(32767.0), max(f32(-32767.0), new))
            self.mult = int(f32(n * f32(65536.0)))


def _amp_model(mult, data):
    if mult == 0:
        return None
    if mult == 65536:
        return data.copy()
    return np.array([max(-32768, min(32767, (mult * int(s)) >> 16)) for s in data], np.int16)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_frontend_oracle_vs_second_model(orc, seed):
    rng = np.random.default_rng(seed)
    nblk = 90
    n = nblk * 128
    t = np.arange(n)
    level = [3000, 30000, 200][seed]                    # AGC pulls up / pushes down / saturates its gain
    adc = (32768 + level * (0.6 + 0.4 * np.sin(2 * np.pi * t / 5000)) * np.cos(2 * np.pi * 6000 * t / 24000)
           + rng.integers(-50, 51, n)).clip(0, 65535).astype(np.uint16)
    if seed == 1:
        adc[700:760] = 65535                            # rail: saturating DC block and amplifier
        adc[760:800] = 0
    f = orc.frontend_new(first_conversion=int(adc[0]))
    got = orc.frontend_run(f, adc)
    x1, y1 = int(adc[0]) << 14, 0
    agc = AgcModel()
    want = np.empty(n, np.int16)
    for b in range(nblk):
        x1, y1, blk = _dcblock_model(x1, y1, adc[128 * b:128 * (b + 1)])
        blk = _amp_model(agc.mult, blk)
        want[128 * b:128 * (b + 1)] = blk
        agc.block(blk)
    assert np.array_equal(got, want)
    assert f.agc.multiplier == agc.mult and np.float32(f.agc.AGC_val) == agc.val and f.agc.agc_idx == agc.idx
    assert f.dc.hpf_x1 == x1 and f.dc.hpf_y1 == y1
    assert list(f.agc.agc_buffer) == agc.buf


def test_agc_quirks(orc):
    """Start values of the packed words: odd samples can never raise the maximum above what -1 allows nor the minimum
    below 0's floor; a silent block gives d = 0 -> f = +inf -> gain proposal inf, rejected by fagc < AGC_Max."""
    a = orclib.Agc()
    orc.lib.orc_agc_init(orclib.C.byref(a))
    orc.agc_block(a, np.zeros(128, np.int16))
    assert np.float32(a.AGC_val) == np.float32(0.25) and a.agc_idx == 24
    blk = np.zeros(128, np.int16)
    blk[1::2] = -5                                       # only odd samples negative: the packed minimum's high half starts at 0
    orc.agc_block(a, blk)
    m = AgcModel(); m.block(np.zeros(128, np.int16)); m.block(blk)
    assert list(a.agc_buffer) == m.buf and a.multiplier == m.mult
    for k in range(30):                                  # across the dropped 26th store
        blk = np.full(128, 100 * (k + 1), np.int16)
        orc.agc_block(a, blk); m.block(blk)
        assert list(a.agc_buffer) == m.buf and a.agc_idx == m.idx and a.multiplier == m.mult, k


def test_amp_and_dcblock_units(orc):
    assert orc.amp_multiplier(1.0) == 65536 and orc.amp_multiplier(0.25) == 16384 and orc.amp_multiplier(1e9) == 32767 * 65536
    d = np.array([-32768, -1, 0, 1, 32767, 12345], np.int16)
    assert orc.amp_update(0, d) is None
    assert np.array_equal(orc.amp_update(65536, d), d)
    for mult in (16384, 3 * 65536, -65536, 70000, 1):
        assert np.array_equal(orc.amp_update(mult, d), _amp_model(mult, d)), mult
    st = orclib.DcBlock(0, 5 << 14)
    adc = np.array([5, 5, 6, 65535, 0, 40000] * 30, np.uint16)
    got = orc.dcblock(st, adc)
    x1, y1, want = _dcblock_model(5 << 14, 0, adc)
    assert np.array_equal(got, want) and st.hpf_x1 == x1 and st.hpf_y1 == y1
