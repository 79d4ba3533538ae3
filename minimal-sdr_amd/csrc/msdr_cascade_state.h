// msdr_cascade_state.h -- host-side bridge between the two ways a df1 biquad cascade's state is kept (no device code).
//
// CMSIS (arm_biquad_cascade_df1_f32, prototype src/CMSIS_5/arm_math.h:1333-1351; pState layout arm_math.h:1233) keeps, per stage s,
// x_s[n-1], x_s[n-2], y_s[n-1], y_s[n-2] with x_1 = the cascade's input d and x_(s+1) = y_s.  A caller may rewrite pCoeffs while the
// stream runs and the filter carries on from that state (the reference does exactly this to its Teensy biquad on every retune,
// Minimal-SDR.ino:356 -> filter_biquad.cpp:84-100, history kept :95-97).
//
// The block-parallel kernels (msdr_biquad.hiph, the cascade inside the chain kernels) evaluate "numerators first, all-pole sections
// afterwards": v = C(z) d with C = prod B_s, then w_1 = v / A_1, w_2 = w_1 / A_2 ... y = w_S.  Their state record is
//     lib[0..7]  = d[n-1-k]            (numerator history, 2 S entries used)
//     lib[8+2s]  = w_(s+1)[n-1],  lib[9+2s] = w_(s+1)[n-2]
// and w_s = (B_(s+1) ... B_S) y_s: the same stream in another basis, and the basis depends on the coefficients.  A live coefficient
// change with CMSIS semantics therefore goes  lib state (old coefficients) -> CMSIS state -> lib state (new coefficients):
//
//   lib -> CMSIS.  Unknowns y_s[-1 .. -L_s], L_s = 2 + 2 (S - s), for s = 1 .. S-1 (y_0 = d and y_S = w_S are known).  Equations:
//       sum_k G_s[k] y_s[-j-k] = w_s[-j]  (j = 1, 2;  G_s = B_(s+1) ... B_S)   and the stage's own recurrence
//       y_s[-j] = a1 y_s[-j-1] + a2 y_s[-j-2] + b0 y_(s-1)[-j] + b1 y_(s-1)[-j-1] + b2 y_(s-1)[-j-2]   (j = 1 .. L_s - 2):
//       (S - 1)(S + 2) equations for as many unknowns, singular only where a numerator G_s shares a root with A_s.
//   CMSIS -> lib.  With zero input from now on, both forms produce a signal that obeys the order-2S recurrence of A_1 ... A_S from
//       sample 2 S on, so they agree for ever once they agree on samples 0 .. 2S-1.  The numerator history d stays what it is (the
//       true inputs); the 2 S section states w are solved from  R w = z - z_d  (z: the CMSIS form's zero-input response from its
//       state, z_d: the lib form's from the history alone, R: its response to unit section states).
//
// Everything in long double on the host; the matrices depend on the coefficients only, so a batch of channels costs one
// factorisation and a matrix-vector product per channel.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace msdr {
namespace cstate {

typedef long double real;
constexpr int kMaxS = 4;

// dense solve with full pivoting; rank-deficient systems get their free unknowns set to zero.  Returns false when the system is
// inconsistent (residual above tol x the right-hand side's scale).  A is n x n row-major and is destroyed; B is n x m (m right-hand sides).
inline bool solve_full_pivot(std::vector<real> &A, std::vector<real> &B, int n, int m, real rank_tol, int *rank_out = nullptr)
{
    std::vector<int> colperm(n);
    for (int i = 0; i < n; i++) colperm[i] = i;
    real amax = 0;
    for (real v : A) amax = std::max(amax, std::fabs(v));
    int rank = 0;
    for (int k = 0; k < n; k++) {
        int pr = -1, pc = -1;
        real best = 0;
        for (int i = k; i < n; i++)
            for (int j = k; j < n; j++)
                if (std::fabs(A[(size_t)i * n + j]) > best) { best = std::fabs(A[(size_t)i * n + j]); pr = i; pc = j; }
        if (pr < 0 || best <= rank_tol * amax) break;
        if (pr != k) {
            for (int j = 0; j < n; j++) std::swap(A[(size_t)pr * n + j], A[(size_t)k * n + j]);
            for (int j = 0; j < m; j++) std::swap(B[(size_t)pr * m + j], B[(size_t)k * m + j]);
        }
        if (pc != k) {
            for (int i = 0; i < n; i++) std::swap(A[(size_t)i * n + pc], A[(size_t)i * n + k]);
            std::swap(colperm[pc], colperm[k]);
        }
        const real piv = A[(size_t)k * n + k];
        for (int i = k + 1; i < n; i++) {
            const real f = A[(size_t)i * n + k] / piv;
            if (f == 0) continue;
            for (int j = k; j < n; j++) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
            for (int j = 0; j < m; j++) B[(size_t)i * m + j] -= f * B[(size_t)k * m + j];
        }
        rank++;
    }
    if (rank_out) *rank_out = rank;
    // rows below the rank must have (numerically) zero right-hand sides
    real bmax = 0;
    for (real v : B) bmax = std::max(bmax, std::fabs(v));
    for (int i = rank; i < n; i++)
        for (int j = 0; j < m; j++)
            if (std::fabs(B[(size_t)i * m + j]) > 1e-9L * std::max(bmax, (real)1e-300L)) return false;
    std::vector<real> X((size_t)n * m, 0);
    for (int k = rank - 1; k >= 0; k--)
        for (int j = 0; j < m; j++) {
            real a = B[(size_t)k * m + j];
            for (int c = k + 1; c < rank; c++) a -= A[(size_t)k * n + c] * X[(size_t)c * m + j];
            X[(size_t)k * m + j] = a / A[(size_t)k * n + k];
        }
    std::vector<real> out((size_t)n * m, 0);
    for (int k = 0; k < n; k++)
        for (int j = 0; j < m; j++) out[(size_t)colperm[k] * m + j] = X[(size_t)k * m + j];
    B.swap(out);
    return true;
}

struct Cascade {
    int S = 0;
    real b[kMaxS][3], a1[kMaxS], a2[kMaxS];
    std::vector<real> cnum;                 // C = B_1 ... B_S, 2 S + 1 entries
    void set(const float *coeffs, int stages)
    {
        S = stages;
        for (int s = 0; s < S; s++) {
            for (int k = 0; k < 3; k++) b[s][k] = (real)coeffs[5 * s + k];
            a1[s] = (real)coeffs[5 * s + 3]; a2[s] = (real)coeffs[5 * s + 4];
        }
        cnum = num_from(0);
    }
    // B_(from+1) ... B_S as a polynomial in z^-1 (from = 0-based index of the first stage taken)
    std::vector<real> num_from(int from) const
    {
        std::vector<real> c(1, (real)1);
        for (int s = from; s < S; s++) {
            std::vector<real> nx(c.size() + 2, (real)0);
            for (size_t i = 0; i < c.size(); i++)
                for (int k = 0; k < 3; k++) nx[i + k] += c[i] * b[s][k];
            c.swap(nx);
        }
        return c;
    }
    // zero-input response of the CMSIS form over n samples.  Y[2 s], Y[2 s + 1] = y_s[-1], y_s[-2], s = 0 .. S (y_0 = d)
    void zir_cmsis(const real *Y, real *z, int n) const
    {
        real st[kMaxS + 1][2];
        for (int s = 0; s <= S; s++) { st[s][0] = Y[2 * s]; st[s][1] = Y[2 * s + 1]; }
        for (int t = 0; t < n; t++) {
            real x = 0;                                             // the cascade's input
            for (int s = 0; s < S; s++) {
                const real y = b[s][0] * x + b[s][1] * st[s][0] + b[s][2] * st[s][1] + a1[s] * st[s + 1][0] + a2[s] * st[s + 1][1];
                st[s][1] = st[s][0]; st[s][0] = x;
                x = y;
            }
            st[S][1] = st[S][0]; st[S][0] = x;
            z[t] = x;
        }
    }
    // zero-input response of the lib form: D[k] = d[-1-k] (k < 2 S), w[2 s], w[2 s + 1] = w_(s+1)[-1], w_(s+1)[-2]
    void zir_lib(const real *D, const real *w, real *z, int n) const
    {
        real st[kMaxS][2];
        for (int s = 0; s < S; s++) { st[s][0] = w[2 * s]; st[s][1] = w[2 * s + 1]; }
        for (int t = 0; t < n; t++) {
            real u = 0;
            for (int k = t + 1; k <= 2 * S; k++) u += cnum[k] * D[k - t - 1];      // d[t - k] = D[k - t - 1]
            for (int s = 0; s < S; s++) {
                const real y = u + a1[s] * st[s][0] + a2[s] * st[s][1];
                st[s][1] = st[s][0]; st[s][0] = y;
                u = y;
            }
            z[t] = u;
        }
    }
};

// The two linear maps for one pair (old coefficients, new coefficients), applied to any number of channels.
struct Bridge {
    Cascade oldc, newc;
    int S = 0;
    bool ok_to = false, ok_from = false;
    // lib (4 S: D[0 .. 2S-1], w[0 .. 2S-1]) -> CMSIS (2 S + 2) under the OLD coefficients
    std::vector<real> to_cmsis;             // (2 S + 2) x (4 S)
    // CMSIS (2 S + 2) and history D (2 S) -> w (2 S) under the NEW coefficients
    std::vector<real> from_cmsis;           // (2 S) x (4 S + 2): columns = Y then D

    void build(const float *old_coeffs, const float *new_coeffs, int stages)
    {
        S = stages;
        oldc.set(old_coeffs, S); newc.set(new_coeffs, S);
        ok_to = build_to(); ok_from = build_from();
    }

    bool build_to()
    {
        const int nin = 4 * S, nout = 2 * S + 2;
        to_cmsis.assign((size_t)nout * nin, (real)0);
        if (S == 0) return true;
        // unknown layout: stage s (1-based, 1 .. S-1) owns y_s[-1 .. -L_s]
        int off[kMaxS + 1] = {0}, L[kMaxS + 1] = {0}, U = 0;
        for (int s = 1; s <= S - 1; s++) { L[s] = 2 + 2 * (S - s); off[s] = U; U += L[s]; }
        std::vector<real> A((size_t)U * U, (real)0), B((size_t)U * std::max(nin, 1), (real)0);
        int row = 0;
        for (int s = 1; s <= S - 1; s++) {
            const std::vector<real> G = oldc.num_from(s);                    // B_(s+1) ... B_S: 2 (S - s) + 1 entries
            for (int j = 1; j <= 2; j++, row++) {
                for (int k = 0; k < (int)G.size(); k++) A[(size_t)row * U + off[s] + (j + k - 1)] += G[k];      // y_s[-(j + k)]
                B[(size_t)row * nin + 2 * S + 2 * (s - 1) + (j - 1)] = 1;                                         // w_s[-j]
            }
            for (int j = 1; j <= L[s] - 2; j++, row++) {
                A[(size_t)row * U + off[s] + (j - 1)] += 1;
                A[(size_t)row * U + off[s] + j] -= oldc.a1[s - 1];
                A[(size_t)row * U + off[s] + (j + 1)] -= oldc.a2[s - 1];
                for (int k = 0; k < 3; k++) {
                    const int depth = j + k;                                     // y_(s-1)[-depth]
                    if (s == 1) B[(size_t)row * nin + (depth - 1)] += oldc.b[0][k];                              // d[-depth] = D[depth - 1]
                    else A[(size_t)row * U + off[s - 1] + (depth - 1)] -= oldc.b[s - 1][k];
                }
            }
        }
        if (U > 0) {
            int rank = 0;
            if (!solve_full_pivot(A, B, U, nin, 1e-13L, &rank) || rank < U) return false;
        }
        // Y[0..1] = d[-1], d[-2];  Y[2 s ..] = y_s[-1], y_s[-2];  Y[2 S ..] = w_S[-1], w_S[-2]
        to_cmsis[(size_t)0 * nin + 0] = 1; to_cmsis[(size_t)1 * nin + 1] = 1;
        for (int s = 1; s <= S - 1; s++)
            for (int j = 0; j < 2; j++)
                for (int c = 0; c < nin; c++) to_cmsis[(size_t)(2 * s + j) * nin + c] = B[(size_t)(off[s] + j) * nin + c];
        to_cmsis[(size_t)(2 * S) * nin + 2 * S + 2 * (S - 1)] = 1;
        to_cmsis[(size_t)(2 * S + 1) * nin + 2 * S + 2 * (S - 1) + 1] = 1;
        return true;
    }

    bool build_from()
    {
        const int n = 2 * S, nin = 2 * S + 2 + 2 * S;
        from_cmsis.assign((size_t)std::max(n, 1) * nin, (real)0);
        if (S == 0) return true;
        std::vector<real> R((size_t)n * n), rhs((size_t)n * nin, (real)0), z(n), e(std::max(nin, n), (real)0), zero(nin, (real)0);
        for (int i = 0; i < n; i++) {                         // unit section states
            std::fill(e.begin(), e.end(), (real)0); e[i] = 1;
            newc.zir_lib(zero.data(), e.data(), z.data(), n);
            for (int t = 0; t < n; t++) R[(size_t)t * n + i] = z[t];
        }
        for (int c = 0; c < 2 * S + 2; c++) {                 // unit CMSIS states
            std::fill(e.begin(), e.end(), (real)0); e[c] = 1;
            newc.zir_cmsis(e.data(), z.data(), n);
            for (int t = 0; t < n; t++) rhs[(size_t)t * nin + c] = z[t];
        }
        for (int c = 0; c < 2 * S; c++) {                     // unit history entries: minus their own response
            std::fill(e.begin(), e.end(), (real)0); e[c] = 1;
            newc.zir_lib(e.data(), zero.data(), z.data(), n);
            for (int t = 0; t < n; t++) rhs[(size_t)t * nin + 2 * S + 2 + c] = -z[t];
        }
        // (the history's first two entries ARE the CMSIS form's x_1 state: a caller passes the same values in both places)
        if (!solve_full_pivot(R, rhs, n, nin, 1e-13L)) return false;
        from_cmsis = rhs;
        return true;
    }

    // lib16: the 16-float state record; D: the TRUE numerator history d[-1-k] (k < 8; for a chain mode that folds the numerator into
    // its FIR this is not what the record holds).  Y: 2 S + 2 values.
    void lib_to_cmsis(const double *D, const float *lib16, double *Y) const
    {
        const int nin = 4 * S;
        real in[4 * kMaxS];
        for (int k = 0; k < 2 * S; k++) in[k] = (real)D[k];
        for (int k = 0; k < 2 * S; k++) in[2 * S + k] = (real)lib16[8 + k];
        for (int r = 0; r < 2 * S + 2; r++) {
            real a = 0;
            for (int c = 0; c < nin; c++) a += to_cmsis[(size_t)r * nin + c] * in[c];
            Y[r] = (double)a;
        }
        if (S == 0) { Y[0] = D[0]; Y[1] = D[1]; }
    }
    // w16: receives lib[8 .. 8 + 2S) (the section states under the new coefficients)
    void cmsis_to_lib(const double *Y, const double *D, float *w_out) const
    {
        const int nin = 4 * S + 2;
        real in[4 * kMaxS + 2];
        for (int k = 0; k < 2 * S + 2; k++) in[k] = (real)Y[k];
        for (int k = 0; k < 2 * S; k++) in[2 * S + 2 + k] = (real)D[k];
        for (int r = 0; r < 2 * S; r++) {
            real a = 0;
            for (int c = 0; c < nin; c++) a += from_cmsis[(size_t)r * nin + c] * in[c];
            w_out[r] = (float)a;
        }
    }
    // CMSIS pState (4 per stage: x[n-1], x[n-2], y[n-1], y[n-2]) <-> Y
    static void y_to_pstate(const double *Y, int S, float *pState)
    {
        for (int s = 0; s < S; s++) {
            pState[4 * s + 0] = (float)Y[2 * s]; pState[4 * s + 1] = (float)Y[2 * s + 1];
            pState[4 * s + 2] = (float)Y[2 * s + 2]; pState[4 * s + 3] = (float)Y[2 * s + 3];
        }
    }
    static void pstate_to_y(const float *pState, int S, double *Y)
    {
        if (S == 0) { Y[0] = Y[1] = 0.0; return; }
        Y[0] = pState[0]; Y[1] = pState[1];
        for (int s = 0; s < S; s++) { Y[2 * s + 2] = pState[4 * s + 2]; Y[2 * s + 3] = pState[4 * s + 3]; }
    }
};

}  // namespace cstate
}  // namespace msdr
