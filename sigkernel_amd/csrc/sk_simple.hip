// sk_simple.hip -- the simple, bit-reproducible solver kernels.
//
// One 64-lane wavefront per pair sweeps the anti-diagonals of the fine grid; the three live
// diagonals sit in LDS and the lanes stride along the diagonal.  This translation unit is
// compiled with -ffp-contract=off and evaluates every cell in the operand order of the
// reference's generated C (sigkernel/cython_backend.pyx:114-116), so its results are
// bit-identical to the reference's CPU solver.  It serves SK_FLAG_EXACT / SK_FLAG_SIMPLE
// requests, shapes the tiled kernels do not cover, full-grid output, and the robust adjoint
// (stored grids) that backs up the fast adjoint when its self-check trips.
//
// Reference behaviour replaced: sigkernel_cuda / sigkernel_Gram_cuda (cuda_backend.py:6-49,
// :121-160) -- without their global-memory solution buffer, 1024-thread limit or
// out-of-bounds extra row/column.
#include "sk_pair_sweep.h"

namespace sk {

namespace {

template <typename T>
__global__ __launch_bounds__(WAVE) void k_fwd_simple(const T *__restrict__ inc_c, int64_t ld, int64_t P, int Mc, int Nc, int d,
                                                     int naive, T *__restrict__ out_final, T *__restrict__ out_grid,
                                                     double *__restrict__ out_edges) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int MM = Mc << d, NN = Nc << d;
    const int64_t gs = (int64_t)(MM + 1) * (NN + 1);
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
        const double v = sweep_pair<T, T, false>(inc_c + p * (int64_t)Mc * ld, ld, Mc, Nc, d, naive, lds,
                                                 out_grid ? out_grid + p * gs : nullptr,
                                                 out_edges ? out_edges + p * (int64_t)(MM + NN + 2) : nullptr);
        if (threadIdx.x == 0 && out_final) out_final[p] = (T)v;
        __syncthreads();
    }
}


template <typename T>
__global__ __launch_bounds__(WAVE) void k_adj_simple(const T *__restrict__ inc_c, int64_t ld, int64_t P, int Mc, int Nc, int d,
                                                     int naive, T *__restrict__ out_final, T *__restrict__ W,
                                                     int64_t ldw, double *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t gs = (int64_t)((Mc << d) + 1) * ((Nc << d) + 1);
    double *Kf = ws + (int64_t)blockIdx.x * 2 * gs;
    double *Kr = Kf + gs;
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x)
        adj_pair<T>(inc_c + p * (int64_t)Mc * ld, ld, Mc, Nc, d, naive, lds, Kf, Kr, out_final ? out_final + p : nullptr,
                    W + p * (int64_t)Mc * ldw, ldw);
}

// Device-side rescue of the fast adjoint: every block scans 64 self-check residuals at a time and re-solves, with both
// grids stored, exactly the pairs whose residual exceeds `tol` (a NaN residual is left alone when the pair's increments are poisoned
// themselves, and re-solved when they are finite: an overflow of the recompute).  Launched unconditionally after the fast
// kernel: when nothing is flagged (the normal case) it reads P doubles and exits, so the host never has to look at the
// residuals -- no device-to-host synchronisation in a backward pass.
template <typename T>
__global__ __launch_bounds__(WAVE) void k_adj_rescue(const T *__restrict__ inc_c, int64_t ld, int64_t P, int Mc, int Nc, int d,
                                                     int naive, const double *__restrict__ err, double tol,
                                                     T *__restrict__ out_final, T *__restrict__ W, int64_t ldw,
                                                     double *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t gs = (int64_t)((Mc << d) + 1) * ((Nc << d) + 1);
    double *Kf = ws + (int64_t)blockIdx.x * 2 * gs;
    double *Kr = Kf + gs;
    for (int64_t p0 = (int64_t)blockIdx.x * WAVE; p0 < P; p0 += (int64_t)gridDim.x * WAVE) {
        const int64_t p = p0 + threadIdx.x;
        // a NaN residual usually means the inputs of the pair are already poisoned (NaN / inf coordinates): a stored-grid re-solve
        // would spend milliseconds to produce NaN again -- such a pair keeps the fast kernel's NaN in W.  But a recompute that
        // overflowed on FINITE increments (inf - inf) also ends in NaN: those pairs are told apart by one pass over the pair's
        // increments and re-solved like any other failed self-check
        const double e = p < P ? err[p] : 0.0;
        const bool bad = p < P && e > tol;
        unsigned long long m = __ballot(bad), mn = __ballot(p < P && e != e);
        while (mn) {
            const int b = __ffsll((long long)mn) - 1;
            mn &= mn - 1;
            const T *iq = inc_c + (p0 + b) * (int64_t)Mc * ld;
            bool fin = true;
            for (int c = threadIdx.x; c < Mc * Nc; c += WAVE) fin &= isfinite((double)iq[(int64_t)(c / Nc) * ld + c % Nc]);
            if (__all(fin)) m |= 1ull << b;
        }
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int64_t q = p0 + b;
            adj_pair<T>(inc_c + q * (int64_t)Mc * ld, ld, Mc, Nc, d, naive, lds, Kf, Kr, out_final ? out_final + q : nullptr,
                        W + q * (int64_t)Mc * ldw, ldw);
        }
    }
}

// Directional-derivative sweep: K, K_gamma, K_gamma_gamma on the same anti-diagonals (nine live diagonals in
// LDS), every cell in the operand order of sigkernel_derivatives_Gram_cuda (cuda_backend.py:206-220), so the
// results are bit-identical to oracle/sigkernel_oracle.c:sk_oracle_solve_deriv_coarse.
template <typename T>
__global__ __launch_bounds__(WAVE) void k_deriv_simple(const T *__restrict__ inc0, const T *__restrict__ inc1,
                                                       const T *__restrict__ inc2, int64_t ld, int64_t P, int Mc, int Nc,
                                                       int d, T *__restrict__ out_k, T *__restrict__ out_kd,
                                                       T *__restrict__ out_kdd) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int MM = Mc << d, NN = Nc << d;
    const double rs = 1.0 / (double)(1 << d);
    const int W1 = MM + 1;
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
        const int64_t base = p * (int64_t)Mc * ld;
        // q[s][n]: diagonal n (0 = two back, 1 = previous, 2 = current) of state s
        double *q[3][3];
        for (int s = 0; s < 3; ++s)
            for (int n = 0; n < 3; ++n) q[s][n] = lds + (s * 3 + n) * W1;
        for (int sd = 2; sd <= MM + NN; ++sd) {
            const int ilo = max(1, sd - NN), ihi = min(MM, sd - 1);
            for (int i = ilo + lane; i <= ihi; i += WAVE) {
                const int j = sd - i;
                const bool e10 = j == 1, e01 = i == 1, e00 = i == 1 || j == 1;
                const double k10 = e10 ? 1. : q[0][1][i], k01 = e01 ? 1. : q[0][1][i - 1], k00 = e00 ? 1. : q[0][0][i - 1];
                const double k10d = e10 ? 0. : q[1][1][i], k01d = e01 ? 0. : q[1][1][i - 1], k00d = e00 ? 0. : q[1][0][i - 1];
                const double k10dd = e10 ? 0. : q[2][1][i], k01dd = e01 ? 0. : q[2][1][i - 1],
                             k00dd = e00 ? 0. : q[2][0][i - 1];
                const int64_t o = base + (int64_t)((i - 1) >> d) * ld + ((j - 1) >> d);
                const double inc = ((double)inc0[o] * rs) * rs, incd = ((double)inc1[o] * rs) * rs,
                             incdd = ((double)inc2[o] * rs) * rs;
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
                q[0][2][i] = k11; q[1][2][i] = k11d; q[2][2][i] = k11dd;
                if (i == MM && j == NN) {
                    if (out_k) out_k[p] = (T)k11;
                    if (out_kd) out_kd[p] = (T)k11d;
                    if (out_kdd) out_kdd[p] = (T)k11dd;
                }
            }
            __syncthreads();
            for (int s = 0; s < 3; ++s) {
                double *t = q[s][0]; q[s][0] = q[s][1]; q[s][1] = q[s][2]; q[s][2] = t;
            }
        }
    }
}

int pick_blocks(int64_t P) {
    const int64_t cap = 256 * 16;  // 16 single-wave workgroups per CU keep every SIMD busy
    return (int)(P < cap ? P : cap);
}

}  // namespace

size_t simple_lds_bytes(const Geom &g) { return sizeof(double) * 3 * (size_t)(g.MM + 1); }

size_t adj_simple_workspace_bytes(const Geom &g) {
    const int64_t gs = (int64_t)(g.MM + 1) * (g.NN + 1);
    int64_t blocks = g.P < 1024 ? g.P : 1024;
    return (size_t)blocks * 2 * gs * sizeof(double);
}

template <typename T>
int launch_fwd_simple(const T *inc_c, const Geom &g, T *out_final, T *out_grid, double *out_edges, hipStream_t s) {
    const size_t lds = simple_lds_bytes(g);
    if (lds > 160 * 1024) return SK_ERR_UNSUPPORTED;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_fwd_simple<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SK_LAUNCH(k_fwd_simple<T>, dim3(pick_blocks(g.P)), dim3(WAVE), lds, s, inc_c, g.ld, g.P, g.Mc, g.Nc,
                       g.dyadic, g.naive, out_final, out_grid, out_edges);
    return check_launch();
}

template <typename T>
int launch_adj_simple(const T *inc_c, const Geom &g, T *out_final, T *W, int64_t ldw, void *ws, size_t ws_bytes,
                      hipStream_t s) {
    const size_t lds = simple_lds_bytes(g);
    if (lds > 160 * 1024) return SK_ERR_UNSUPPORTED;
    const int64_t gs = (int64_t)(g.MM + 1) * (g.NN + 1);
    const size_t per_block = (size_t)2 * gs * sizeof(double);
    if (!ws || ws_bytes < per_block) return SK_ERR_WORKSPACE;
    int64_t blocks = (int64_t)(ws_bytes / per_block);
    if (blocks > g.P) blocks = g.P;
    if (blocks > 1024) blocks = 1024;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_adj_simple<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SK_LAUNCH(k_adj_simple<T>, dim3((int)blocks), dim3(WAVE), lds, s, inc_c, g.ld, g.P, g.Mc, g.Nc, g.dyadic,
                       g.naive, out_final, W, ldw, (double *)ws);
    return check_launch();
}

// Re-solve, with stored grids, the pairs whose fast-adjoint residual err[p] exceeds tol.  ws: any number (>= 1) of
// per-block scratch slots of 2 (MM+1)(NN+1) doubles; the blocks grid-stride over the residuals.
template <typename T>
int launch_adj_rescue(const T *inc_c, const Geom &g, const double *err, double tol, T *out_final, T *W, int64_t ldw, void *ws,
                      size_t ws_bytes, hipStream_t s) {
    const size_t lds = simple_lds_bytes(g);
    if (lds > 160 * 1024) return SK_ERR_UNSUPPORTED;
    const int64_t gs = (int64_t)(g.MM + 1) * (g.NN + 1);
    const size_t per_block = (size_t)2 * gs * sizeof(double);
    if (!ws || ws_bytes < per_block) return SK_ERR_WORKSPACE;
    int64_t blocks = (int64_t)(ws_bytes / per_block);
    const int64_t need = (g.P + WAVE - 1) / WAVE;
    if (blocks > need) blocks = need;
    if (blocks > 1024) blocks = 1024;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_adj_rescue<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SK_LAUNCH(k_adj_rescue<T>, dim3((int)blocks), dim3(WAVE), lds, s, inc_c, g.ld, g.P, g.Mc, g.Nc, g.dyadic, g.naive,
                       err, tol, out_final, W, ldw, (double *)ws);
    return check_launch();
}
template int launch_adj_rescue<double>(const double *, const Geom &, const double *, double, double *, double *, int64_t, void *,
                                       size_t, hipStream_t);
template int launch_adj_rescue<float>(const float *, const Geom &, const double *, double, float *, float *, int64_t, void *, size_t,
                                      hipStream_t);

template int launch_fwd_simple<double>(const double *, const Geom &, double *, double *, double *, hipStream_t);
template int launch_fwd_simple<float>(const float *, const Geom &, float *, float *, double *, hipStream_t);
template int launch_adj_simple<double>(const double *, const Geom &, double *, double *, int64_t, void *, size_t,
                                       hipStream_t);
template int launch_adj_simple<float>(const float *, const Geom &, float *, float *, int64_t, void *, size_t, hipStream_t);

template <typename T>
int launch_deriv_simple(const T *inc, const T *inc_d, const T *inc_dd, const Geom &g, T *out_k, T *out_kd, T *out_kdd,
                        hipStream_t s) {
    const size_t lds = 3 * simple_lds_bytes(g);
    if (lds > 160 * 1024) return SK_ERR_UNSUPPORTED;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_deriv_simple<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SK_LAUNCH(k_deriv_simple<T>, dim3(pick_blocks(g.P)), dim3(WAVE), lds, s, inc, inc_d, inc_dd, g.ld, g.P, g.Mc,
                       g.Nc, g.dyadic, out_k, out_kd, out_kdd);
    return check_launch();
}

template int launch_deriv_simple<double>(const double *, const double *, const double *, const Geom &, double *, double *,
                                         double *, hipStream_t);
template int launch_deriv_simple<float>(const float *, const float *, const float *, const Geom &, float *, float *, float *,
                                        hipStream_t);

}  // namespace sk
