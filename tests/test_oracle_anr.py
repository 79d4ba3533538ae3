"""Row f3 (LMS automatic notch / noise reduction, Minimal-SDR.ino:702-770): the C oracle against an independent numpy
restatement (float32 scalars, double where an unsuffixed literal enters, sums in tap order)."""
import numpy as np
import pytest

import orclib

f32 = np.float32


@pytest.fixture(scope="module")
def orc():
    return orclib.Oracle()


class AnrModel:
    def __init__(self):
        self.lidx, self.ngamma, self.idx = f32(120.0), f32(0.001), 0
        self.d, self.w = np.zeros(512, f32), np.zeros(64, f32)

    def run(self, on, data):
        two_mu, gamma, den = f32(0.001), f32(0.1), f32(6.25e-10)
        out = np.array(data, np.int16)
        for i in range(out.size):
            self.d[self.idx] = f32(out[i])
            y, sigma = f32(0), f32(0)
            taps = [(self.idx + j + 16) & 511 for j in range(64)]
            for j, k in enumerate(taps):
                y = f32(y + f32(self.w[j] * self.d[k]))
                sigma = f32(sigma + f32(self.d[k] * self.d[k]))
            inv = f32(1.0 / (float(sigma) + 1e-10))
            err = f32(self.d[self.idx] - y)
            v = int(err) if on == 1 else int(y)
            out[i] = ((v + 32768) % 65536) - 32768
            nel = f32(float(err) * (1.0 - float(f32(f32(two_mu * sigma) * inv))))
            nel = -nel if float(nel) < 0.0 else nel
            nev = f32(float(self.d[self.idx]) - (1.0 - float(f32(two_mu * self.ngamma))) * float(y)
                      - float(f32(f32(f32(two_mu * err) * sigma) * inv)))
            nev = -nev if float(nev) < 0.0 else nev
            if nev < nel:
                self.lidx = f32(self.lidx + f32(1.0))
                if self.lidx > f32(200.0):
                    self.lidx = f32(200.0)
                else:
                    self.lidx = f32(self.lidx - f32(3.0))
                    if self.lidx < f32(0.0):
                        self.lidx = f32(0.0)
            l2 = f32(self.lidx * self.lidx)
            self.ngamma = f32(f32(f32(gamma * l2) * l2) * den)
            c0 = f32(1.0 - float(f32(two_mu * self.ngamma)))
            c1 = f32(f32(two_mu * err) * inv)
            for j, k in enumerate(taps):
                self.w[j] = f32(f32(c0 * self.w[j]) + f32(c1 * self.d[k]))
            self.idx = (self.idx + 511) & 511
        return out


def signal(rng, n, tone=1000.0, level=6000, noise=800):
    t = np.arange(n)
    return (level * np.sin(2 * np.pi * tone * t / 24000) + 2000 * np.sin(2 * np.pi * 310 * t / 24000 + 1)
            + rng.integers(-noise, noise + 1, n)).astype(np.int16)


@pytest.mark.parametrize("on", [1, 2])
def test_anr_oracle_vs_numpy_model(orc, on):
    rng = np.random.default_rng(on)
    x = signal(rng, 900)
    x[300:310] = 32767                                # big steps: |error| beyond int16, the store wraps
    x[310:320] = -32768
    a = orc.anr_new()
    got = np.concatenate([orc.anr_q15(a, on, x[:400]), orc.anr_q15(a, on, x[400:])])
    m = AnrModel()
    want = m.run(on, x)
    assert np.array_equal(got, want)
    assert f32(a.lidx) == m.lidx and f32(a.ngamma) == m.ngamma and a.in_idx == m.idx
    assert np.array_equal(np.array(a.w[:64], f32), m.w) and np.array_equal(np.array(a.d[:], f32), m.d)


def test_anr_off_is_untouched(orc):
    x = signal(np.random.default_rng(0), 256)
    a = orc.anr_new()
    assert np.array_equal(orc.anr_q15(a, 0, x), x) and a.in_idx == 0


def test_anr_notch_removes_a_steady_tone(orc):
    """Sanity of the restatement as a filter: the notch output loses most of a steady carrier."""
    n = 24000
    x = (8000 * np.sin(2 * np.pi * 1000 * np.arange(n) / 24000)).astype(np.int16)
    y = orc.anr_q15(orc.anr_new(), 1, x)
    assert np.abs(y[-4000:]).astype(float).std() < 0.2 * np.abs(x[-4000:]).astype(float).std()
