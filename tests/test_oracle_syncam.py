"""Row f2 (SYNCAM PLL, Minimal-SDR.ino:631-688): the C oracle against an independent numpy restatement (float32 scalars,
library calls in double then rounded once), and the host-side constants of the product library against the oracle's."""
import os
import sys

import numpy as np
import pytest

import orclib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimal-sdr_amd", "python"))


@pytest.fixture(scope="module")
def orc():
    return orclib.Oracle()


def pll_model(i, q, st=None):
    f32 = np.float32
    fs = 24000
    omega_n, zeta = f32(400.0), f32(0.45)
    pi = 3.1415926535897932384626433832795
    omega_min, omega_max = f32(2.0 * pi * -4000.0 / fs), f32(2.0 * pi * 4000.0 / fs)
    g1 = f32(1.0 - np.exp(-2.0 * float(omega_n) * float(zeta) / fs))
    e_arg = f32(f32(-omega_n * zeta) / f32(fs))
    c_arg = f32(f32(omega_n / f32(fs)) * f32(np.sqrt(f32(1.0 - float(f32(zeta * zeta))))))
    g2 = f32(-float(g1) + 2.0 * (1 - np.exp(float(e_arg)) * float(f32(np.cos(c_arg)))))
    fil, om, ph = st or (f32(0), f32(0), f32(0))
    out = np.empty(len(i), np.int16)
    for k in range(len(i)):
        s, c = f32(np.sin(float(ph))), f32(np.cos(float(ph)))
        fi, fq = f32(i[k]), f32(q[k])
        c0 = f32(f32(c * fi) + f32(s * fq))
        c1 = f32(f32(-f32(s * fi)) + f32(c * fq))
        out[k] = np.int16(((int(c0) + 32768) % 65536) - 32768)
        det = f32(np.arctan2(float(c1), float(c0)))
        delo = fil
        om = f32(om + f32(g2 * det))
        om = omega_min if om < omega_min else omega_max if om > omega_max else om
        fil = f32(f32(g1 * det) + om)
        ph = f32(ph + delo)
        while float(ph) >= 2 * pi:
            ph = f32(float(ph) - 2.0 * pi)
        while float(ph) < 0.0:
            ph = f32(float(ph) + 2.0 * pi)
    return out, (fil, om, ph), (omega_min, omega_max, g1, g2)


def test_syncam_oracle_vs_numpy_model(orc):
    rng = np.random.default_rng(3)
    n = 3000
    t = np.arange(n)
    car = 2 * np.pi * 37.0 * t / 24000 + 0.7                     # a carrier 37 Hz off: the PLL has to pull in
    env = 9000 * (1 + 0.5 * np.sin(2 * np.pi * 400 * t / 24000))
    i = (env * np.cos(car) + rng.integers(-30, 31, n)).astype(np.int16)
    q = (env * np.sin(car) + rng.integers(-30, 31, n)).astype(np.int16)
    i[100:110] = 32767; q[100:110] = -32768                      # |corr0| > 32767: the halfword store wraps
    s = orc.syncam_new()
    got = np.concatenate([orc.syncam_q15(s, i[:1000], q[:1000]), orc.syncam_q15(s, i[1000:], q[1000:])])
    want, st, consts = pll_model(i, q)
    assert np.array_equal(got, want)
    assert (np.float32(s.fil_out), np.float32(s.omega2), np.float32(s.phzerror)) == st
    assert np.array_equal(orc.syncam_constants(), np.array(consts, np.float32))


def test_product_constants_match_oracle(orc):
    import msdr
    assert np.array_equal(msdr.syncam_constants(), orc.syncam_constants())
