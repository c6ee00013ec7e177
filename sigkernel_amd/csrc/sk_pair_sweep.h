// sk_pair_sweep.h -- one wavefront solves one pair with its grids stored: the anti-diagonal sweep of sk_simple.hip and the
// stored-grid adjoint built on it, shared with sk_adj_fused_rescue.hip (the fused adjoints' device-side rescue).
#pragma once
#include "sk_internal.h"

namespace sk {
namespace {

#ifndef SK_WAVE_DEFINED
#define SK_WAVE_DEFINED
constexpr int WAVE = 64;
#endif

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

// Robust adjoint of one pair: both solution grids are written to the block's scratch slot, then every
// coarse cell sums its r*r products in the oracle's order (i-major), so W is bit-identical
// to oracle/sigkernel_oracle.c:sk_oracle_adjoint_coarse.
template <typename T>
__device__ void adj_pair(const T *__restrict__ inc, int64_t ld, int Mc, int Nc, int d, int naive, double *lds, double *Kf,
                         double *Kr, T *__restrict__ out_final_p, T *__restrict__ Wp, int64_t ldw) {
    const int MM = Mc << d, NN = Nc << d, r = 1 << d;
    const double rs = 1.0 / (double)r;
    const double v = sweep_pair<T, double, false>(inc, ld, Mc, Nc, d, naive, lds, Kf, nullptr);
    sweep_pair<T, double, true>(inc, ld, Mc, Nc, d, naive, lds, Kr, nullptr);
    __syncthreads();
    if (threadIdx.x == 0 && out_final_p) *out_final_p = (T)v;
    for (int c = threadIdx.x; c < Mc * Nc; c += WAVE) {
        const int a = c / Nc, b = c - a * Nc;
        double acc = 0.;
        for (int ii = 0; ii < r; ++ii)
            for (int jj = 0; jj < r; ++jj) {
                const int i = a * r + ii, j = b * r + jj;
                acc += Kf[(int64_t)i * (NN + 1) + j] * Kr[(int64_t)(MM - 1 - i) * (NN + 1) + (NN - 1 - j)];
            }
        Wp[(int64_t)a * ldw + b] = (T)((acc * rs) * rs);
    }
    __syncthreads();
}

}  // namespace
}  // namespace sk
