"""Index-level model of chain_amsy_kernel (minimal-sdr_amd/csrc/msdr_chain_amsy.hiph): the envelope chain behind the exact Fs/4 mixer for
LINEAR-PHASE taps (what calc_FIR_coeffs always designs, Minimal-SDR.ino:782-872: symmetric about an integer delay D0), with the symmetry used
to halve the matrix products.  Not a test of the product: a numpy restatement of the kernel's index maps (window arrays, the two copies, operand
elements per lane, tap matrices, recombination) checked against the direct sums, used to derive the host-side table builder.

    python tests/debug/amsy_model.py
"""
import numpy as np


def kslot(s, lg, jj):
    """window position k of K-slot jj (0..7) of lane group lg (0..3) in k-step s: dword pairs kappa = lg + 4 i"""
    return 32 * s + 2 * (lg + 4 * (jj >> 1)) + (jj & 1)


def find_center(hd):
    """integer D0 with hd[d] == hd[2 D0 - d] (zero outside), or None"""
    nz = np.nonzero(hd)[0]
    if len(nz) == 0:
        return None
    s = nz[0] + nz[-1]
    if s & 1:
        return None
    D0 = s // 2
    for d in range(len(hd)):
        o = 2 * D0 - d
        v = hd[o] if 0 <= o < len(hd) else 0.0
        if hd[d] != v:
            return None
    return D0


def sy_steps(D0):
    return (D0 + 32 + 63) // 64          # K = D0 + 32 window positions (R), pairs K / 2, 32 per step


def sy_hh(D0, ns):
    return ((max(D0, 32 * ns - 28, 32) + 31) // 32) * 32


def build_tables(h):
    """h = pCoeffs (CMSIS order).  Returns D0, ns, M[4][ns][16 rows][32 slots-in-k-order] (R1, R2, I1, I2)"""
    N = len(h)
    hd = np.array([h[N - 1 - d] for d in range(N)], dtype=np.float64)
    D0 = find_center(hd)
    if D0 is None:
        return None
    gR = lambda d: hd[d] * (-1.0 if (d & 2) else 1.0) if (0 <= d < N and not (d & 1)) else 0.0
    gI = lambda d: hd[d] * (1.0 if (d & 2) else -1.0) if (0 <= d < N and (d & 1)) else 0.0
    ns = sy_steps(D0)
    out = np.zeros((4, 32 * ns, 16))
    for typ in range(2):
        J = D0 + 1 if typ == 0 else D0
        a = np.array([gR(2 * j) if typ == 0 else gI(2 * j + 1) for j in range(J)])
        eps = 1.0 if (D0 % 2 == 0) == (typ == 0) else -1.0          # eps_R = (-1)^D0, eps_I = -eps_R
        assert np.array_equal(a, eps * a[::-1]), "sub-filter symmetry"
        K = J + 31
        T = np.zeros((32, K))
        for p in range(32):
            for k in range(K):
                j = p + J - 1 - k
                if 0 <= j < J:
                    T[p, k] = a[j]
        for r in range(16):
            for k in range((K + 1) // 2):
                U = T[r, k] + T[31 - r, k]
                V = T[r, k] - T[31 - r, k]
                if (K & 1) and k == K // 2:
                    U *= 0.5
                    V *= 0.5
                # M1 pairs with f + b, M2 with f - b
                m1, m2 = (U, V) if eps > 0 else (V, U)
                out[2 * typ, k, r] = m1
                out[2 * typ + 1, k, r] = m2
    return D0, ns, out


def model_tile(x, t0, h):
    """x: int samples (long enough, index t >= 0), t0: tile start (multiple of 1024 and >= H).  Returns A[m], B[m] (the two real FIRs) for the tile"""
    D0, ns, M = build_tables(h)
    Hh = sy_hh(D0, ns)
    H = 2 * Hh
    ne = Hh + 512 + 64
    xe = np.zeros(ne)
    xo = np.zeros(ne)
    for e in range(Hh + 512):
        xe[e] = x[t0 - H + 2 * e]
        xo[e] = x[t0 - H + 2 * e + 1]
    FR = Hh - D0
    A = np.zeros(1024)
    B = np.zeros(1024)
    for b in range(16):
        # conv: (array, fwd base, bwd base, matrices, rho, is_I)
        convs = [(xe, FR, Hh + 31, 0, 0, 0), (xo, FR, Hh + 31, 0, 1, 0), (xo, FR, Hh + 30, 2, 0, 1), (xe, FR + 1, Hh + 31, 2, 1, 1)]
        for X, fb, bb, mi, rho, isI in convs:
            acc1 = np.zeros(16)
            acc2 = np.zeros(16)
            for s in range(ns):
                for lg in range(4):
                    for jj in range(8):
                        k = kslot(s, lg, jj)
                        f = X[fb + 32 * b + k]
                        bw = X[bb + 32 * b - k]
                        acc1 += M[mi, k] * (f + bw)
                        acc2 += M[mi + 1, k] * (f - bw)
            for r in range(16):
                y1 = 0.5 * (acc1[r] + acc2[r])
                y2 = 0.5 * (acc1[r] - acc2[r])
                m1 = 64 * b + 2 * r + rho
                m2 = 64 * b + 2 * (31 - r) + rho
                if isI:
                    B[m1], B[m2] = y1, y2
                else:
                    A[m1], A[m2] = y1, y2
    return A, B, (D0, ns, Hh)


def direct(x, t0, h):
    N = len(h)
    hd = np.array([h[N - 1 - d] for d in range(N)], dtype=np.float64)
    A = np.zeros(1024)
    B = np.zeros(1024)
    for m in range(1024):
        for d in range(N):
            v = hd[d] * x[t0 + m - d]
            if d & 1:
                B[m] += v * (1.0 if (d & 2) else -1.0)
            else:
                A[m] += v * (-1.0 if (d & 2) else 1.0)
    return A, B


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    x = rng.integers(-32768, 32768, size=8192).astype(np.float64)
    for N, c in ((256, 128), (102, 51), (100, 50), (64, 31), (512, 256), (33, 16), (20, 9), (256, 100)):
        h = np.zeros(N)
        half = min(c, N - 1 - c)
        w = rng.integers(-2000, 2000, size=half + 1).astype(np.float64)
        for i in range(half + 1):
            h[c - i] = w[i]
            h[c + i] = w[i]
        A, B, info = model_tile(x, 2048, h)
        Ad, Bd = direct(x, 2048, h)
        e = np.hypot(A, B)
        ed = np.hypot(Ad, Bd)
        print(N, c, info, "max |A| err", np.abs(np.abs(A) - np.abs(Ad)).max(), "max |B| err", np.abs(np.abs(B) - np.abs(Bd)).max(), "env err", np.abs(e - ed).max(),
              "signed A ok", np.allclose(A, Ad), "signed B ok", np.allclose(B, Bd))
