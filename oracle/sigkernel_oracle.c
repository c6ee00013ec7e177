/*
 * sigkernel_oracle.c -- CPU restatement of the reference's signature-PDE solver.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker (or as the timed CPU baseline), never as
 * a fallback for the HIP kernels.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here
 * against the .npz fixtures under tests/golden/, which were produced by importing the real
 * reference (Cython build of /root/reference/sigkernel/cython_backend.pyx plus
 * /root/reference/sigkernel/sigkernel.py) with tests/golden/make_golden.py.
 * The forward solver is bit-identical to the Cython build (max diff 0.0).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off keeps the arithmetic FMA-free like the reference's plain
 * `-O2` x86-64 Cython build, so results are comparable bit for bit.
 *
 * Citations are relative to /root/reference/.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifdef _OPENMP
#include <math.h>
#include <omp.h>
#endif

/* One stencil update, in the operand order of the C that Cython generates for
 * sigkernel/cython_backend.pyx:116 (default scheme) and :114 (_naive_solver):
 *   (k10 + k01)*(1. + 0.5*g + (1./12)*g**2) - k00*(1. - (1./12)*g**2)
 * k10 = K[i+1,j], k01 = K[i,j+1], k00 = K[i,j]. */
static inline double sk_cell(double k10, double k01, double k00, double g, int naive)
{
    if (naive)
        return (k10 + k01) * (1. + 0.5 * g) - k00;
    return (k10 + k01) * ((1. + 0.5 * g) + (1. / 12.) * (g * g)) - k00 * (1. - (1. / 12.) * (g * g));
}

/* sigkernel_cython (cython_backend.pyx:7-33) and the non-symmetric branch of
 * sigkernel_Gram_cython (:98-117; identical loop with P = A*B).
 * inc : [P, MM, NN] fine increments (already dyadically refined by the caller,
 *       sigkernel.py:218 / :364).
 * K   : [P, MM+1, NN+1] full solution grid, boundary K[.,0,:] = K[.,:,0] = 1. */
void sk_oracle_solve_fine(const double *inc, int64_t P, int MM, int NN, int naive, double *K)
{
    const int64_t gs = (int64_t)(MM + 1) * (NN + 1);
    for (int64_t l = 0; l < P; ++l) {
        double *Kl = K + l * gs;
        const double *gl = inc + l * (int64_t)MM * NN;
        memset(Kl, 0, sizeof(double) * gs);
        for (int i = 0; i <= MM; ++i) Kl[(int64_t)i * (NN + 1)] = 1.;
        for (int j = 0; j <= NN; ++j) Kl[j] = 1.;
        for (int i = 0; i < MM; ++i)
            for (int j = 0; j < NN; ++j) {
                double g = gl[(int64_t)i * NN + j];
                Kl[(int64_t)(i + 1) * (NN + 1) + j + 1] =
                    sk_cell(Kl[(int64_t)(i + 1) * (NN + 1) + j], Kl[(int64_t)i * (NN + 1) + j + 1],
                            Kl[(int64_t)i * (NN + 1) + j], g, naive);
            }
    }
}

/* The sym=True branch of sigkernel_Gram_cython (cython_backend.pyx:74-97):
 * only pairs l <= m are solved and the TRANSPOSED grid is mirrored into (m,l).
 * inc: [A, A, MM, NN], K: [A, A, MM+1, NN+1]; assumes MM == NN like the
 * reference silently does. */
void sk_oracle_gram_sym_fine(const double *inc, int A, int MM, int NN, int naive, double *K)
{
    const int64_t gs = (int64_t)(MM + 1) * (NN + 1);
    memset(K, 0, sizeof(double) * gs * A * A);
    for (int l = 0; l < A; ++l)
        for (int m = l; m < A; ++m) {
            double *Klm = K + ((int64_t)l * A + m) * gs;
            double *Kml = K + ((int64_t)m * A + l) * gs;
            const double *g = inc + ((int64_t)l * A + m) * (int64_t)MM * NN;
            for (int i = 0; i <= MM; ++i) { Klm[(int64_t)i * (NN + 1)] = 1.; Kml[(int64_t)i * (NN + 1)] = 1.; }
            for (int j = 0; j <= NN; ++j) { Klm[j] = 1.; Kml[j] = 1.; }
            for (int i = 0; i < MM; ++i)
                for (int j = 0; j < NN; ++j) {
                    double v = sk_cell(Klm[(int64_t)(i + 1) * (NN + 1) + j], Klm[(int64_t)i * (NN + 1) + j + 1],
                                       Klm[(int64_t)i * (NN + 1) + j], g[(int64_t)i * NN + j], naive);
                    Klm[(int64_t)(i + 1) * (NN + 1) + j + 1] = v;
                    Kml[(int64_t)(j + 1) * (NN + 1) + i + 1] = v;
                }
        }
}

/* Fine increment from the COARSE matrix: the reference materialises
 * tile(tile(inc_c, r)/r, r)/r (sigkernel.py:218, :364); here the same value is
 * produced by index arithmetic, inc_c[i>>d][j>>d] divided by r twice. */
static inline double sk_fine_inc(const double *inc_c, int Nc, int d, double r, int i, int j)
{
    return (inc_c[(int64_t)(i >> d) * Nc + (j >> d)] / r) / r;
}

/* Solve P independent problems from coarse increments.
 * inc_c [P, Mc, Nc]; out_final [P] = K[MM, NN]; grid (nullable) [P, MM+1, NN+1].
 * nthreads > 1 parallelises over problems with OpenMP (the reference is
 * single-threaded: cython_backend.pyx:75,100 have prange commented out).
 * Same arithmetic as sk_oracle_solve_fine; only two grid rows are kept when
 * grid == NULL. Returns 0, or 1 on bad arguments / allocation failure. */
int sk_oracle_solve_coarse(const double *inc_c, int64_t P, int Mc, int Nc, int dyadic, int naive,
                           double *out_final, double *grid, int nthreads)
{
    if (P < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 20) return 1;
    const int MM = Mc << dyadic, NN = Nc << dyadic;
    const double r = (double)(1 << dyadic);
    int fail = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        double *rows = (double *)malloc(sizeof(double) * 2 * (size_t)(NN + 1));
        if (!rows) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(dynamic, 16)
        for (int64_t l = 0; l < P; ++l) {
            if (!rows) continue;
            const double *gl = inc_c + l * (int64_t)Mc * Nc;
            double *G = grid ? grid + l * (int64_t)(MM + 1) * (NN + 1) : NULL;
            double *prev = rows, *cur = rows + (NN + 1);
            for (int j = 0; j <= NN; ++j) prev[j] = 1.;
            if (G) memcpy(G, prev, sizeof(double) * (NN + 1));
            for (int i = 0; i < MM; ++i) {
                cur[0] = 1.;
                for (int j = 0; j < NN; ++j)
                    cur[j + 1] = sk_cell(cur[j], prev[j + 1], prev[j], sk_fine_inc(gl, Nc, dyadic, r, i, j), naive);
                if (G) memcpy(G + (int64_t)(i + 1) * (NN + 1), cur, sizeof(double) * (NN + 1));
                double *t = prev; prev = cur; cur = t;
            }
            if (out_final) out_final[l] = prev[NN];
        }
        free(rows);
    }
    return fail;
}

/* The whole Gram pipeline per pair inside ONE parallel region -- the all-cores CPU baseline of bench.py (not a parity
 * path): static kernel of the pair (static_kernels.py:26-33 linear <x, y>; :58-73 rbf exp(-(|x|^2 + |y|^2 - 2<x, y>) / sigma)),
 * its 4-corner increments (sigkernel.py:362-363) and the PDE solve (cython_backend.pyx:101-117), with every scratch buffer
 * allocated and first touched by the thread that uses it.  sk_oracle_solve_coarse alone leaves the static kernel and the
 * increments serial, which is why a baseline built from it scales 5x on 128 threads.
 * X [A, M, D], Y [B, N, D] row-major; kind 0 linear, 1 rbf (param = sigma); out [A, B].  Returns 0, 1 on bad arguments. */
int sk_oracle_gram_pipeline(const double *X, int64_t A, int M, const double *Y, int64_t B, int N, int D, int kind, double param,
                            int dyadic, int naive, double *out, int nthreads)
{
    if (A < 0 || B < 0 || M < 2 || N < 2 || D < 1 || dyadic < 0 || dyadic > 20 || (kind != 0 && kind != 1)) return 1;
    const int Mc = M - 1, Nc = N - 1, MM = Mc << dyadic, NN = Nc << dyadic;
    const double r = (double)(1 << dyadic);
    int fail = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        double *G = (double *)malloc(sizeof(double) * ((size_t)M * N + (size_t)Mc * Nc + 2 * (size_t)(NN + 1) + M + N));
        if (!G) {
#pragma omp atomic write
            fail = 1;
        }
        double *inc = G ? G + (size_t)M * N : NULL, *rows = G ? inc + (size_t)Mc * Nc : NULL;
        double *xs = G ? rows + 2 * (size_t)(NN + 1) : NULL, *ys = G ? xs + M : NULL;
#pragma omp for schedule(dynamic, 8)
        for (int64_t l = 0; l < A * B; ++l) {
            if (!G) continue;
            const double *x = X + (l / B) * (int64_t)M * D, *y = Y + (l % B) * (int64_t)N * D;
            if (kind == 1) {
                for (int p = 0; p < M; ++p) { double s = 0.; for (int k = 0; k < D; ++k) s += x[p * D + k] * x[p * D + k]; xs[p] = s; }
                for (int q = 0; q < N; ++q) { double s = 0.; for (int k = 0; k < D; ++k) s += y[q * D + k] * y[q * D + k]; ys[q] = s; }
            }
            for (int p = 0; p < M; ++p)
                for (int q = 0; q < N; ++q) {
                    double s = 0.;
                    for (int k = 0; k < D; ++k) s += x[p * D + k] * y[q * D + k];
                    G[(size_t)p * N + q] = kind == 0 ? s : exp(-((-2. * s + xs[p]) + ys[q]) / param);
                }
            for (int p = 0; p < Mc; ++p)
                for (int q = 0; q < Nc; ++q)
                    inc[(size_t)p * Nc + q] = ((G[(size_t)(p + 1) * N + q + 1] + G[(size_t)p * N + q]) - G[(size_t)(p + 1) * N + q]) -
                                              G[(size_t)p * N + q + 1];
            double *prev = rows, *cur = rows + (NN + 1);
            for (int j = 0; j <= NN; ++j) prev[j] = 1.;
            for (int i = 0; i < MM; ++i) {
                cur[0] = 1.;
                for (int j = 0; j < NN; ++j)
                    cur[j + 1] = sk_cell(cur[j], prev[j + 1], prev[j], sk_fine_inc(inc, Nc, dyadic, r, i, j), naive);
                double *t = prev; prev = cur; cur = t;
            }
            out[l] = prev[NN];
        }
        free(G);
    }
    return fail;
}

/* Adjoint ("variation of parameters") weights of prep_backward
 * (sigkernel.py:419-502) and _SigKernel.backward (:257-343), in closed form
 * (SURVEY section 3.2):
 *   K      = forward solution on inc                       (sigkernel.py:366-395)
 *   Krev   = solution on the doubly flipped inc            (:438-467)
 *   Kt[i,j]= Krev[MM-i, NN-j]                              (:469)
 *   KK[i,j]= K[i,j] * Kt[i+1,j+1],  i<MM, j<NN             (:470)
 *   W[p,q] = 4^-d * sum_{(i,j) in coarse cell (p,q)} KK[i,j]
 * W is d k_sig / d inc_c: the reference contracts KK with the finite-difference
 * derivative of the *fine* increments (:483-495), and every fine increment in
 * coarse cell (p,q) is inc_c[p,q]/4^d.
 * inc_c [P,Mc,Nc]; out_final [P] (nullable); W [P,Mc,Nc]. */
int sk_oracle_adjoint_coarse(const double *inc_c, int64_t P, int Mc, int Nc, int dyadic, int naive,
                             double *out_final, double *W, int nthreads)
{
    if (P < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 20) return 1;
    const int MM = Mc << dyadic, NN = Nc << dyadic;
    const double r = (double)(1 << dyadic);
    const int64_t gs = (int64_t)(MM + 1) * (NN + 1);
    int fail = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        double *K = (double *)malloc(sizeof(double) * 2 * (size_t)gs);
        if (!K) {
#pragma omp atomic write
            fail = 1;
        }
        double *Kr = K ? K + gs : NULL;
#pragma omp for schedule(dynamic, 4)
        for (int64_t l = 0; l < P; ++l) {
            if (!K) continue;
            const double *gl = inc_c + l * (int64_t)Mc * Nc;
            double *Wl = W + l * (int64_t)Mc * Nc;
            for (int i = 0; i <= MM; ++i) { K[(int64_t)i * (NN + 1)] = 1.; Kr[(int64_t)i * (NN + 1)] = 1.; }
            for (int j = 0; j <= NN; ++j) { K[j] = 1.; Kr[j] = 1.; }
            for (int i = 0; i < MM; ++i)
                for (int j = 0; j < NN; ++j) {
                    int64_t o = (int64_t)i * (NN + 1) + j;
                    K[o + NN + 2] = sk_cell(K[o + NN + 1], K[o + 1], K[o], sk_fine_inc(gl, Nc, dyadic, r, i, j), naive);
                    Kr[o + NN + 2] = sk_cell(Kr[o + NN + 1], Kr[o + 1], Kr[o],
                                             sk_fine_inc(gl, Nc, dyadic, r, MM - 1 - i, NN - 1 - j), naive);
                }
            if (out_final) out_final[l] = K[gs - 1];
            memset(Wl, 0, sizeof(double) * (size_t)Mc * Nc);
            for (int i = 0; i < MM; ++i)
                for (int j = 0; j < NN; ++j) {
                    double kk = K[(int64_t)i * (NN + 1) + j] * Kr[(int64_t)(MM - 1 - i) * (NN + 1) + (NN - 1 - j)];
                    Wl[(int64_t)(i >> dyadic) * Nc + (j >> dyadic)] += kk;
                }
            for (int64_t c = 0; c < (int64_t)Mc * Nc; ++c) Wl[c] = (Wl[c] / r) / r;
        }
        free(K);
    }
    return fail;
}

/* inc_c = 4-corner double difference of the static Gram (sigkernel.py:217, :363):
 * G[p+1,q+1] + G[p,q] - G[p+1,q] - G[p,q+1], evaluated left to right like torch does. */
void sk_oracle_increments(const double *G, int64_t P, int M, int N, double *inc_c)
{
    for (int64_t l = 0; l < P; ++l)
        for (int p = 0; p < M - 1; ++p)
            for (int q = 0; q < N - 1; ++q) {
                const double *g = G + l * (int64_t)M * N;
                inc_c[l * (int64_t)(M - 1) * (N - 1) + (int64_t)p * (N - 1) + q] =
                    ((g[(int64_t)(p + 1) * N + q + 1] + g[(int64_t)p * N + q]) - g[(int64_t)(p + 1) * N + q]) - g[(int64_t)p * N + q + 1];
            }
}

/* Transpose of sk_oracle_increments: dL/dG from dL/dinc_c = W.
 * dG[m,n] = W[m-1,n-1] + W[m,n] - W[m-1,n] - W[m,n-1] (out-of-range W = 0). */
void sk_oracle_increments_adjoint(const double *W, int64_t P, int M, int N, double *dG)
{
    const int Mc = M - 1, Nc = N - 1;
    for (int64_t l = 0; l < P; ++l)
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                const double *w = W + l * (int64_t)Mc * Nc;
#define SK_W(a, b) (((a) >= 0 && (a) < Mc && (b) >= 0 && (b) < Nc) ? w[(int64_t)(a) * Nc + (b)] : 0.)
                dG[l * (int64_t)M * N + (int64_t)m * N + n] = SK_W(m - 1, n - 1) + SK_W(m, n) - SK_W(m - 1, n) - SK_W(m, n - 1);
#undef SK_W
            }
}

/* Directional-derivative solver: K, K_gamma, K_gamma_gamma in one sweep.
 * Stencil of sigkernel_derivatives_Gram_cuda (cuda_backend.py:206-220), identical to
 * sigkernel_derivatives_Gram_mps (mps_backend.py:118-131), in their operand order; boundary
 * K = 1, K_gamma = K_gamma_gamma = 0 on the first row and column (sigkernel.py:553-558).
 * The reference's Cython twin (cython_backend.pyx:122-184) uses a different stencil and its
 * call site is broken (sigkernel.py:588), so the CUDA/MPS formulas define the behaviour.
 * inc_c, incd_c, incdd_c [P, Mc, Nc] coarse increments of k, of its first and of its second
 * finite-difference derivative along gamma (sigkernel.py:526-541); all three are refined by
 * index and divided by 4^d like sigkernel.py:543-545.  out_* [P] (each nullable) receive the
 * values at (MM, NN); grids (nullable) [3, P, MM+1, NN+1]. */
int sk_oracle_solve_deriv_coarse(const double *inc_c, const double *incd_c, const double *incdd_c, int64_t P, int Mc,
                                 int Nc, int dyadic, double *out_k, double *out_kd, double *out_kdd, double *grids,
                                 int nthreads)
{
    if (P < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 20) return 1;
    const int MM = Mc << dyadic, NN = Nc << dyadic;
    const double r = (double)(1 << dyadic);
    const int64_t gs = (int64_t)(MM + 1) * (NN + 1);
    int fail = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        double *rows = (double *)malloc(sizeof(double) * 6 * (size_t)(NN + 1));
        if (!rows) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(dynamic, 16)
        for (int64_t l = 0; l < P; ++l) {
            if (!rows) continue;
            const double *g0 = inc_c + l * (int64_t)Mc * Nc, *g1 = incd_c + l * (int64_t)Mc * Nc,
                         *g2 = incdd_c + l * (int64_t)Mc * Nc;
            double *pk = rows, *pd = rows + (NN + 1), *pdd = rows + 2 * (NN + 1);
            double *ck = rows + 3 * (NN + 1), *cd = rows + 4 * (NN + 1), *cdd = rows + 5 * (NN + 1);
            for (int j = 0; j <= NN; ++j) { pk[j] = 1.; pd[j] = 0.; pdd[j] = 0.; }
            for (int s = 0; grids && s < 3; ++s)
                memcpy(grids + (s * P + l) * gs, s == 0 ? pk : s == 1 ? pd : pdd, sizeof(double) * (NN + 1));
            for (int i = 0; i < MM; ++i) {
                ck[0] = 1.; cd[0] = 0.; cdd[0] = 0.;
                for (int j = 0; j < NN; ++j) {
                    const double inc = sk_fine_inc(g0, Nc, dyadic, r, i, j);
                    const double incd = sk_fine_inc(g1, Nc, dyadic, r, i, j);
                    const double incdd = sk_fine_inc(g2, Nc, dyadic, r, i, j);
                    const double k01 = pk[j + 1], k10 = ck[j], k00 = pk[j];
                    const double k01d = pd[j + 1], k10d = cd[j], k00d = pd[j];
                    const double k01dd = pdd[j + 1], k10dd = cdd[j], k00dd = pdd[j];
                    const double k11 = (k01 + k10) * ((1. + 0.5 * inc) + (1. / 12) * (inc * inc)) -
                                       k00 * (1. - (1. / 12) * (inc * inc));
                    const double f1 = k00 * incd + k00d * inc;
                    const double f2 = k01 * incd + k01d * inc;
                    const double f3 = k10 * incd + k10d * inc;
                    const double f4 = k11 * incd + (((k01d + k10d) - k00d) + f1) * inc;
                    const double k11d = ((k01d + k10d) - k00d) + 0.25 * (((f1 + f2) + f3) + f4);
                    const double h1 = (k00 * incdd + (2. * k00d) * incd) + k00dd * inc;
                    const double h2 = (k01 * incdd + (2. * k01d) * incd) + k01dd * inc;
                    const double h3 = (k10 * incdd + (2. * k10d) * incd) + k10dd * inc;
                    const double h4 = (k11 * incdd + (2. * k11d) * incd) + (((k01dd + k10dd) - k00dd) + h1) * inc;
                    const double k11dd = ((k01dd + k10dd) - k00dd) + 0.25 * (((h1 + h2) + h3) + h4);
                    ck[j + 1] = k11; cd[j + 1] = k11d; cdd[j + 1] = k11dd;
                }
                if (grids) {
                    memcpy(grids + (0 * P + l) * gs + (int64_t)(i + 1) * (NN + 1), ck, sizeof(double) * (NN + 1));
                    memcpy(grids + (1 * P + l) * gs + (int64_t)(i + 1) * (NN + 1), cd, sizeof(double) * (NN + 1));
                    memcpy(grids + (2 * P + l) * gs + (int64_t)(i + 1) * (NN + 1), cdd, sizeof(double) * (NN + 1));
                }
                double *t;
                t = pk; pk = ck; ck = t;
                t = pd; pd = cd; cd = t;
                t = pdd; pdd = cdd; cdd = t;
            }
            if (out_k) out_k[l] = pk[NN];
            if (out_kd) out_kd[l] = pd[NN];
            if (out_kdd) out_kdd[l] = pdd[NN];
        }
        free(rows);
    }
    return fail;
}

int sk_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
