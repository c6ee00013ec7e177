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
#include "sk_internal.h"

namespace sk {

namespace {

constexpr int WAVE = 64;

// (k10 + k01)*(1. + 0.5*g + (1./12)*g**2) - k00*(1. - (1./12)*g**2), cython_backend.pyx:116;
// _naive_solver: (k10 + k01)*(1. + 0.5*g) - k00, cython_backend.pyx:114.
__device__ __forceinline__ double cell_exact(double k10, double k01, double k00, double g, int naive) {
    if (naive) return (k10 + k01) * (1. + 0.5 * g) - k00;
    return (k10 + k01) * ((1. + 0.5 * g) + (1. / 12.) * (g * g)) - k00 * (1. - (1. / 12.) * (g * g));
}

// Sweep one pair.  FLIP selects the doubly flipped increments (the reverse PDE of
// sigkernel.py:438).  `grid` (nullable) receives the full (MM+1)x(NN+1) node grid in the
// sweep's own coordinates; `edges` (nullable) the terminal row and column.
template <typename T, typename TG, bool FLIP>
__device__ double sweep_pair(const T *__restrict__ inc, int64_t ld, int Mc, int Nc, int d, int naive, double *lds,
                             TG *__restrict__ grid, double *__restrict__ edges) {
    const int lane = threadIdx.x;
    const int MM = Mc << d, NN = Nc << d;
    const double rs = 1.0 / (double)(1 << d);  // power of two: multiplying == the reference's division
    double *d0 = lds, *d1 = lds + (MM + 1), *d2 = lds + 2 * (MM + 1);
    const int64_t gw = NN + 1;

    if (grid) {
        for (int j = lane; j <= NN; j += WAVE) grid[j] = (TG)1.;
        for (int i = lane; i <= MM; i += WAVE) grid[(int64_t)i * gw] = (TG)1.;
    }
    if (edges) {
        if (lane == 0) { edges[0] = 1.; edges[NN + 1] = 1.; }
    }
    double last = 1.;
    for (int s = 2; s <= MM + NN; ++s) {
        const int ilo = max(1, s - NN), ihi = min(MM, s - 1);
        for (int i = ilo + lane; i <= ihi; i += WAVE) {
            const int j = s - i;
            const double k10 = (j == 1) ? 1. : d1[i];
            const double k01 = (i == 1) ? 1. : d1[i - 1];
            const double k00 = (i == 1 || j == 1) ? 1. : d0[i - 1];
            int ci = (i - 1) >> d, cj = (j - 1) >> d;
            if (FLIP) { ci = Mc - 1 - ci; cj = Nc - 1 - cj; }
            const double g = ((double)inc[(int64_t)ci * ld + cj] * rs) * rs;
            const double v = cell_exact(k10, k01, k00, g, naive);
            d2[i] = v;
            if (grid) grid[(int64_t)i * gw + j] = (TG)v;
            if (edges) {
                if (i == MM) edges[j] = v;
                if (j == NN) edges[NN + 1 + i] = v;
            }
            if (i == MM && j == NN) last = v;
        }
        __syncthreads();
        double *t = d0; d0 = d1; d1 = d2; d2 = t;
    }
    // broadcast K[MM][NN] (computed by exactly one lane at the last step)
    const int owner = (MM - max(1, MM)) % WAVE;  // lane of i == MM at s == MM+NN: ilo == MM there
    return __shfl(last, owner, WAVE);
}

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

// Robust adjoint: both solution grids are written to a per-block scratch slot, then every
// coarse cell sums its r*r products in the oracle's order (i-major), so W is bit-identical
// to oracle/sigkernel_oracle.c:sk_oracle_adjoint_coarse.
template <typename T>
__global__ __launch_bounds__(WAVE) void k_adj_simple(const T *__restrict__ inc_c, int64_t ld, int64_t P, int Mc, int Nc, int d,
                                                     int naive, T *__restrict__ out_final, T *__restrict__ W,
                                                     int64_t ldw, double *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int MM = Mc << d, NN = Nc << d, r = 1 << d;
    const int64_t gs = (int64_t)(MM + 1) * (NN + 1);
    const double rs = 1.0 / (double)r;
    double *Kf = ws + (int64_t)blockIdx.x * 2 * gs;
    double *Kr = Kf + gs;
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
        const T *inc = inc_c + p * (int64_t)Mc * ld;
        const double v = sweep_pair<T, double, false>(inc, ld, Mc, Nc, d, naive, lds, Kf, nullptr);
        sweep_pair<T, double, true>(inc, ld, Mc, Nc, d, naive, lds, Kr, nullptr);
        __syncthreads();
        if (threadIdx.x == 0 && out_final) out_final[p] = (T)v;
        for (int c = threadIdx.x; c < Mc * Nc; c += WAVE) {
            const int a = c / Nc, b = c - a * Nc;
            double acc = 0.;
            for (int ii = 0; ii < r; ++ii)
                for (int jj = 0; jj < r; ++jj) {
                    const int i = a * r + ii, j = b * r + jj;
                    acc += Kf[(int64_t)i * (NN + 1) + j] * Kr[(int64_t)(MM - 1 - i) * (NN + 1) + (NN - 1 - j)];
                }
            W[p * (int64_t)Mc * ldw + (int64_t)a * ldw + b] = (T)((acc * rs) * rs);
        }
        __syncthreads();
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
    hipLaunchKernelGGL(k_fwd_simple<T>, dim3(pick_blocks(g.P)), dim3(WAVE), lds, s, inc_c, g.ld, g.P, g.Mc, g.Nc,
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
    hipLaunchKernelGGL(k_adj_simple<T>, dim3((int)blocks), dim3(WAVE), lds, s, inc_c, g.ld, g.P, g.Mc, g.Nc, g.dyadic,
                       g.naive, out_final, W, ldw, (double *)ws);
    return check_launch();
}

template int launch_fwd_simple<double>(const double *, const Geom &, double *, double *, double *, hipStream_t);
template int launch_fwd_simple<float>(const float *, const Geom &, float *, float *, double *, hipStream_t);
template int launch_adj_simple<double>(const double *, const Geom &, double *, double *, int64_t, void *, size_t,
                                       hipStream_t);
template int launch_adj_simple<float>(const float *, const Geom &, float *, float *, int64_t, void *, size_t, hipStream_t);

}  // namespace sk
