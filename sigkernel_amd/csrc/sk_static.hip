// sk_static.hip -- static kernel + increments in one pass, for the two static kernels of the benchmark configs.
//
// Reference: G_static = static_kernel.Gram_matrix(X, Y) / batch_kernel(X, Y) (static_kernels.py:24,33,56,73)
// followed by the 4-corner difference (sigkernel.py:216-217, :362-363).  At the headline size that is a 34 GB
// tensor written, read back and differenced into another 34 GB tensor.  Here one workgroup per pair forms the
// coarse increments directly from the two paths (a few KB, L2-resident) and writes them once, in the padded
// layout the solver kernels stream: the only HBM traffic left is the increment matrix itself.
//   linear : inc[p][q] = s^2 <x[p+1]-x[p], y[q+1]-y[q]>            (s = 1 for Gram_matrix, which ignores `scale`,
//                                                                  s = scale for batch_kernel: static_kernels.py:24,33)
//   rbf    : G[p][q] = exp(-(|x_p|^2 + |y_q|^2 - 2<x_p,y_q>)/sigma), inc = ((G11 + G00) - G10) - G01
// The linear form is algebraically the 4-corner difference of <x_p, y_q>; it is evaluated as a product of
// differences (no cancellation), so it agrees with the reference to rounding (1e-16 absolute), not bit for bit.
//
// Kernels: k_static_linear (dim <= 8, VALU) and k_static_linear_mfma (9..32 dims, v_mfma_f64_16x16x4_f64); k_static_nodes (rbf, and the three
// node arrays of the directional derivative); their adjoints k_static_rbf_adj (dim <= 16, or second paths of more than 128 points),
// k_static_rbf_adj_tiled (17..32 dims) and k_static_linear_adj_tiled (9..32 dims) -- y_b through LDS, pairs loaded a chunk ahead --,
// k_linear_adj_dyt (dim <= 8, pre-differenced dimension-major y), and the second-argument forms k_linear_adj2 / k_rbf_adj2.
#include <type_traits>

#include "sk_internal.h"

namespace sk {
namespace {

constexpr int SK_TPB = 64;   // one wavefront; lane t owns node column c0 + t, outputs for t < 63

// (LIN_CPT: output columns per lane -- 2 amortise the row differences of x, which are wave-uniform; 1 for second paths of <= 65 points, whose
// second column would be all padding: half of the arithmetic of a 64-point path, 0.74 -> 0.5 ms per 256 x 256 pairs at 20 dims)

template <typename T, int DMAX, int LIN_CPT>
__global__ __launch_bounds__(SK_TPB) void k_static_linear(const T *__restrict__ X, const T *__restrict__ Y, int64_t B,
                                                          int M, int N, int D, double s2, T *__restrict__ inc,
                                                          int64_t ld, int col_tiles) {
    const int Mc = M - 1, Nc = N - 1;
    const int64_t p = blockIdx.x / col_tiles;                          // pair
    const int c0 = (int)(blockIdx.x % col_tiles) * (SK_TPB * LIN_CPT);  // first output column of this block
    const int64_t a = B > 0 ? p / B : p, b = B > 0 ? p % B : p;
    const T *x = X + a * (int64_t)M * D;
    const T *y = Y + b * (int64_t)N * D;
    T *o = inc + p * (int64_t)Mc * ld;
    double dy[LIN_CPT][DMAX];
#pragma unroll
    for (int c = 0; c < LIN_CPT; ++c) {
        const int q = c0 + c * SK_TPB + threadIdx.x;
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
            dy[c][k] = (k < D && q < Nc) ? s2 * ((double)y[(int64_t)(q + 1) * D + k] - (double)y[(int64_t)q * D + k]) : 0.0;
    }
    {
        // the row differences of x through LDS -- lane = row forms its DMAX differences once per tile of 64 rows, every row of
        // the sweep reads them back with DMAX / 2 broadcast ds_read_b128.  (As 2 D scalar loads per row they put a scalar-cache round trip
        // on every row's critical path: 256 x 256 pairs of 64 points, dim 32: 3.7 ms, dim 8: 0.75; now 1.1 / 0.39 -- dim <= 8 at the
        // speed the increments can be written; profiles/r06_wide_dims.txt.)
        __shared__ __attribute__((aligned(16))) double xl[SK_TPB * DMAX];
        for (int i0 = 0; i0 < Mc; i0 += SK_TPB) {
            __syncthreads();
            {
                const int ir = min(i0 + (int)threadIdx.x, Mc - 1);
#pragma unroll
                for (int k = 0; k < DMAX; k += 2) {
                    const double d0 = k < D ? (double)x[(int64_t)(ir + 1) * D + k] - (double)x[(int64_t)ir * D + k] : 0.0;
                    const double d1 = k + 1 < D ? (double)x[(int64_t)(ir + 1) * D + k + 1] - (double)x[(int64_t)ir * D + k + 1] : 0.0;
                    *reinterpret_cast<double2 *>(&xl[threadIdx.x * DMAX + k]) = double2{d0, d1};
                }
            }
            __syncthreads();
            const int rows = min(SK_TPB, Mc - i0);
            for (int r = 0; r < rows; ++r) {
                double acc[LIN_CPT];
#pragma unroll
                for (int c = 0; c < LIN_CPT; ++c) acc[c] = 0.0;
#pragma unroll
                for (int k = 0; k < DMAX; k += 2) {
                    const double2 dx = *reinterpret_cast<const double2 *>(&xl[r * DMAX + k]);   // the same address in every lane: a broadcast
#pragma unroll
                    for (int c = 0; c < LIN_CPT; ++c) acc[c] = fma(dx.y, dy[c][k + 1], fma(dx.x, dy[c][k], acc[c]));
                }
#pragma unroll
                for (int c = 0; c < LIN_CPT; ++c) {
                    const int q = c0 + c * SK_TPB + threadIdx.x;
                    if (q < ld) o[(int64_t)(i0 + r) * ld + q] = (T)acc[c];
                }
            }
        }
    }
}

// The same on the matrix cores for paths of 9..32 dims: inc = s^2 dX dY^T IS a matrix product, and beyond 8 dims the kernel above is bound by
// its DMAX / 2 broadcast LDS reads per row (0.63 / 0.78 / 1.11 ms per 256 x 256 pairs of 64 points at 12 / 20 / 32 dims, against 0.39 for the
// 2.1 GB to be written).  One wave per 64 x 64 tile of a pair's increments; v_mfma_f64_16x16x4_f64 takes one double of A = dX (row lane & 15,
// any dim the lanes of a quarter wave agree on) and of B = s^2 dY (the same with columns) per lane, gathered straight from the paths (10 KB each, in L1 / L2),
// and leaves C[row (lane >> 4) + 4 r][col lane & 15] in register r -- sixteen consecutive doubles of a row per quarter wave, whole 128-byte
// lines.  fp64 MFMA issues at the VALU's rate on this part (64 cycles per 1024 FMAs): what is saved is the LDS traffic and the
// instruction count, not arithmetic time.  Sums of four dims at a time inside the instruction: equal to the kernel above to rounding, not bits.
typedef double d4_t __attribute__((ext_vector_type(4)));
template <typename T> struct Pair2;      // two consecutive path values, at the paths' own alignment
template <> struct Pair2<double> { typedef double2 __attribute__((aligned(8))) type; };
template <> struct Pair2<float> { typedef float2 __attribute__((aligned(4))) type; };
// v[k0], v[k0 + 1] of (row r1) - (row r0), zero from dim D on: one vector load per row where both dims exist
template <typename T>
__device__ __forceinline__ void diff_pair(const T *r1, const T *r0, int k0, int D, double &o0, double &o1) {
    typedef typename Pair2<T>::type P2;
    if (k0 + 1 < D) {
        const P2 u = *reinterpret_cast<const P2 *>(r1 + k0), v = *reinterpret_cast<const P2 *>(r0 + k0);
        o0 = (double)u.x - (double)v.x;
        o1 = (double)u.y - (double)v.y;
    } else {
        const int kc = min(k0, D - 1);
        o0 = k0 < D ? (double)r1[kc] - (double)r0[kc] : 0.0;
        o1 = 0.0;
    }
}
template <typename T, int DMAX>
__global__ __launch_bounds__(64) void k_static_linear_mfma(const T *__restrict__ X, const T *__restrict__ Y, int64_t B, int M, int N, int D,
                                                           double s2, T *__restrict__ inc, int64_t ld, int row_tiles, int col_tiles) {
    constexpr int KS = DMAX / 4;      // MFMA steps; step ks takes dim (lane >> 4) KS + ks from each lane: a lane's dims are consecutive in memory
    const int Mc = M - 1, Nc = N - 1;
    int64_t blk = blockIdx.x;
    const int ct = (int)(blk % col_tiles);
    blk /= col_tiles;
    const int rt = (int)(blk % row_tiles);
    const int64_t p = blk / row_tiles;
    const int64_t a = B > 0 ? p / B : p, b = B > 0 ? p % B : p;
    const T *x = X + a * (int64_t)M * D;
    const T *y = Y + b * (int64_t)N * D;
    T *o = inc + p * (int64_t)Mc * ld;
    const int li = threadIdx.x & 15, lk = threadIdx.x >> 4, kb = lk * KS;
    double bf[4][KS];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int q = ct * 64 + nt * 16 + li, qc = min(q, Nc - 1);
        const T *y0 = y + (int64_t)qc * D;
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            double v0, v1;
            diff_pair<T>(y0 + D, y0, kb + ks, D, v0, v1);
            bf[nt][ks] = q < Nc ? s2 * v0 : 0.0;
            bf[nt][ks + 1] = q < Nc ? s2 * v1 : 0.0;
        }
    }
#pragma unroll 1
    for (int mt = 0; mt < 4; ++mt) {
        const int p0 = rt * 64 + mt * 16;
        if (p0 >= Mc) break;
        const int pr = p0 + li;
        const T *x0 = x + (int64_t)min(pr, Mc - 1) * D;
        double af[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            double v0, v1;
            diff_pair<T>(x0 + D, x0, kb + ks, D, v0, v1);
            af[ks] = pr < Mc ? v0 : 0.0;
            af[ks + 1] = pr < Mc ? v1 : 0.0;
        }
        d4_t acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], bf[nt][ks], acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = ct * 64 + nt * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = p0 + lk + 4 * r;
                if (row < Mc && col < ld) o[(int64_t)row * ld + col] = (T)acc[nt][r];
            }
        }
    }
}

// ---- node-evaluating kernels: rbf increments, and the increments of the directional-derivative path ------------
//
// One wavefront per (pair, tile of 64*CPT output columns); lane l owns output columns c0 + 64c + l (c < CPT), i.e.
// whole 128-byte lines per store.  An output needs the static kernel at node columns q and q+1: q+1 comes from the
// next lane (next chunk for lane 63); the one node column past the tile is evaluated for 64 node rows at a time with
// lanes = rows, and read back row by row with v_readlane -- 1/64 extra evaluation instead of a second, nearly empty
// tile.  NV = 1: G -> inc (rbf).  NV = 3: the directional-derivative path, fusing the three Gram_matrix calls of
// k_kgrad (sigkernel.py:526, :530, :537) with its finite-difference pre-processing (:527-541): X0 = X,
// X1 = X + eps*gamma, X2 = X + 2*eps*gamma are formed by the caller (tiny); each node value is scaled like the
// reference (-(1/eps)*G0, (1/eps)*G1, -(1/eps)*(-(1/eps)*G0), -(2/eps)*((1/eps)*G1), (1/eps^2)*G2), every scaled
// array is 4-corner-differenced and the differences are added left to right -- the operand order of
// k_deriv_increments, without the three (A,B,M,N) Gram matrices in HBM.
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float readlane_f64(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// static kernel at one node: x (wave-uniform), its squared norm xs, y node (per lane) and its squared norm ys
// FAST: the stripped exp of sk_internal.h (19 instructions, finite arguments <= a rounding above 0) -- the plain increments;
// the derivative increments keep the library exp: their 1/eps^2 amplification makes the last bit of a node part of the result
// the fixtures pin (section 4.6 of DESIGN.md)
template <int DMAX, int KIND, bool FAST = false>
__device__ __forceinline__ double static_node(const double (&xv)[DMAX], double xs, const double (&yv)[DMAX], double ys,
                                               double inv_sigma) {
    double xy = 0.0;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) xy = fma(xv[k], yv[k], xy);
    // rbf: dist = -2 xy + (xs + ys);  G = exp(-dist / sigma)          (static_kernels.py:53-56, :70-73)
    if constexpr (KIND == 0) return xy;
    const double e = -(fma(-2.0, xy, xs + ys)) * inv_sigma;
    if constexpr (FAST) return exp_nonpos(e);
    return exp(e);
}
template <int DMAX>
__device__ __forceinline__ double sqnorm(const double (&v)[DMAX]) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) s = fma(v[k], v[k], s);
    return s;
}

template <typename T, int DMAX, int KIND, int NV, int CPT>
__global__ __launch_bounds__(SK_TPB) void k_static_nodes(const T *__restrict__ X0, const T *__restrict__ X1,
                                                         const T *__restrict__ X2, const T *__restrict__ Y, int64_t B, int M,
                                                         int N, int D, double inv_sigma, T c1, T c2, T c3,
                                                         T *__restrict__ inc, T *__restrict__ inc_d, T *__restrict__ inc_dd,
                                                         int64_t ld, int col_tiles) {
    typedef typename std::conditional<NV == 1, double, T>::type TA;   // NV = 1 keeps G in double until the store
    constexpr int NS = NV == 1 ? 1 : 6;                              // scaled arrays per node
    const int Mc = M - 1, Nc = N - 1;
    const int64_t p = blockIdx.x / col_tiles;
    const int c0 = (int)(blockIdx.x % col_tiles) * (SK_TPB * CPT);
    const int64_t a = B > 0 ? p / B : p, b = B > 0 ? p % B : p;
    const T *xs[3] = {X0 + a * (int64_t)M * D, X1 + a * (int64_t)M * D, X2 + a * (int64_t)M * D};
    const T *y = Y + b * (int64_t)N * D;
    const int64_t o = p * (int64_t)Mc * ld;
    const int lane = threadIdx.x;

    double yn[CPT][DMAX], ys[CPT], ye[DMAX], yse = 0.0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int n = min(c0 + c * SK_TPB + lane, N - 1);
        ys[c] = 0.0;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            yn[c][k] = k < D ? (double)y[(int64_t)n * D + k] : 0.0;
            ys[c] = fma(yn[c][k], yn[c][k], ys[c]);
        }
    }
    {
        const int ne = min(c0 + CPT * SK_TPB, N - 1);   // the node column just past the tile
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            ye[k] = k < D ? (double)y[(int64_t)ne * D + k] : 0.0;
            yse = fma(ye[k], ye[k], yse);
        }
    }

    auto scaled = [&](const TA (&g)[NV], TA (&v)[NS]) {
#pragma clang fp contract(off)
        v[0] = g[0];
        if constexpr (NV == 3) {
            v[1] = -c1 * g[0];
            v[2] = c1 * g[1];
            v[3] = -c1 * v[1];
            v[4] = -c2 * v[2];
            v[5] = c3 * g[2];
        }
    };

    TA vp[CPT][NS], vpr[CPT][NS];   // scaled values of the previous node row at columns q and q + 1
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int k = 0; k < NS; ++k) vp[c][k] = vpr[c][k] = (TA)0;

    for (int i0 = 0; i0 < M; i0 += SK_TPB) {
        // the x rows of this block, lanes = rows: they serve the extra node column, and (narrow paths) every row of the
        // sweep below reads its x through v_readlane from here -- a scalar load per row and dimension would put one
        // scalar-cache round trip on the critical path of every iteration
        // (9..32 dims, plain increments: through LDS instead -- lane = row writes its DMAX values once per tile, every row of the sweep is
        // DMAX / 2 broadcast ds_read_b128, issued ONE ROW AHEAD of its use: the kernel is 89 % VALU-busy and 64 v_readlane per row were a
        // third of its vector instructions, profiles/r06_wide_dims.txt)
        constexpr bool XREG = DMAX <= 8, XLDS = NV == 1 && DMAX > 8;
        __shared__ __attribute__((aligned(16))) double xl[XLDS ? SK_TPB * DMAX : 2];
        double xr[NV][XREG ? DMAX : 1], xsq[NV];   // xsq: |x_row|^2, computed once per row here instead of once per node
        TA E[NV];
        {
            const int ir = min(i0 + lane, M - 1);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double xv[DMAX];
#pragma unroll
                for (int k = 0; k < DMAX; ++k) xv[k] = k < D ? (double)xs[v][(int64_t)ir * D + k] : 0.0;
                if constexpr (XREG) {
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) xr[v][k] = xv[k];
                }
                if constexpr (XLDS) {
                    __syncthreads();      // (the rows of the tile before have been swept)
#pragma unroll
                    for (int k = 0; k < DMAX; k += 2) *reinterpret_cast<double2 *>(&xl[lane * DMAX + k]) = double2{xv[k], xv[k + 1]};
                    __syncthreads();
                }
                xsq[v] = sqnorm<DMAX>(xv);
                E[v] = (TA)static_node<DMAX, KIND, NV == 1>(xv, xsq[v], ye, yse, inv_sigma);
            }
        }
        const int rows = min(SK_TPB, M - i0);
        double xnext[XLDS ? DMAX : 1];      // XLDS: the x of the row after the one being evaluated, already on its way from LDS
        if constexpr (XLDS) {
#pragma unroll
            for (int k = 0; k < DMAX; k += 2) {
                const double2 t2 = *reinterpret_cast<const double2 *>(&xl[k]);
                xnext[k] = t2.x; xnext[k + 1] = t2.y;
            }
        }
        for (int r = 0; r < rows; ++r) {
            const int i = i0 + r;
            TA g[CPT][NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double xv[DMAX];
                if constexpr (XLDS) {
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) xv[k] = xnext[k];
                    const int rn = min(r + 1, SK_TPB - 1);
#pragma unroll
                    for (int k = 0; k < DMAX; k += 2) {      // the same address in every lane: a broadcast
                        const double2 t2 = *reinterpret_cast<const double2 *>(&xl[rn * DMAX + k]);
                        xnext[k] = t2.x; xnext[k + 1] = t2.y;
                    }
                }
#pragma unroll
                for (int k = 0; k < DMAX; ++k) {
                    if constexpr (XREG) xv[k] = readlane_f64(xr[v][k], r);                       // wave-uniform
                    else if constexpr (!XLDS) xv[k] = (double)xs[v][(int64_t)i * D + min(k, D - 1)];   // unconditional (mergeable) scalar loads
                }
                if constexpr (!XREG && !XLDS) {
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) xv[k] = k < D ? xv[k] : 0.0;
                }
                const double xs_row = readlane_f64(xsq[v], r);
#pragma unroll
                for (int c = 0; c < CPT; ++c) g[c][v] = (TA)static_node<DMAX, KIND, NV == 1>(xv, xs_row, yn[c], ys[c], inv_sigma);
            }
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                TA gr[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const TA nxt = __shfl_down(g[c][v], 1, SK_TPB);
                    const TA wrap = c + 1 < CPT ? readlane_f64(g[c + 1 < CPT ? c + 1 : c][v], 0) : readlane_f64(E[v], r);
                    gr[v] = lane == SK_TPB - 1 ? wrap : nxt;
                }
                TA v0[NS], v1[NS];
                scaled(g[c], v0);
                scaled(gr, v1);
                const int q = c0 + c * SK_TPB + lane;
                if (i > 0 && q < ld) {
#pragma clang fp contract(off)
                    TA d[NS];
#pragma unroll
                    for (int k = 0; k < NS; ++k) d[k] = ((v1[k] + vp[c][k]) - v0[k]) - vpr[c][k];   // ((G11 + G00) - G10) - G01
                    const bool in = q < Nc;
                    const int64_t at = o + (int64_t)(i - 1) * ld + q;
                    inc[at] = in ? (T)d[0] : (T)0;
                    if constexpr (NV == 3) {
                        inc_d[at] = in ? (T)(d[1] + d[2]) : (T)0;
                        inc_dd[at] = in ? (T)((d[3] + d[4]) + d[5]) : (T)0;
                    }
                }
#pragma unroll
                for (int k = 0; k < NS; ++k) { vp[c][k] = v0[k]; vpr[c][k] = v1[k]; }
            }
        }
    }
}

// ---- adjoints: dL/dX from W = dL/d inc_c, without materialising dL/dG_static ------------------------------
// (replaces the finite-difference contraction of sigkernel.py:313-341 / :472-500 together with the
//  `grad_output * grad_points` reduction of :343 / :410-416, for these two static kernels)

// block-wide sum of `v` over all threads (64 or 128 threads); result valid in thread 0
template <int NT>
__device__ __forceinline__ double block_sum(double v, double *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (NT > 64) {
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < NT / 64; ++w) v += red[w];
        }
        __syncthreads();
    }
    return v;
}

// linear, from pre-differenced paths: T[a][p][k] = sum_b s_ab sum_q W[a,b,p,q] * dYt[b][k][q], with dYt [Bn][DP][ldy]
// dimension-major and zero-padded (the array sk_solve_fwd_linear_* takes, DP = 8).  Thread = column q: the 8 dy values
// and the RS rows of W are 16 fully coalesced loads per pair, no per-element predicates; two pairs are in flight per
// iteration.  (k_static_linear_adj above differences y in the kernel with 8-byte loads strided by the path dimension
// and a predicate per element: 11.7 ms per headline tile against 3.4 ms of W traffic.)
template <typename T, int NT>
__global__ __launch_bounds__(NT) void k_linear_adj_dyt(const double *__restrict__ dYt, int64_t ldy, const T *__restrict__ W,
                                                       int64_t ldw, const T *__restrict__ scale, int64_t B, int Mc, int Nc,
                                                       int D, int strips, T *__restrict__ Tout) {
    constexpr int DP = 8, RS = 8;
    __shared__ double red[NT / 64 + 1];
    const int64_t a = blockIdx.x / strips;
    const int p0 = (int)(blockIdx.x % strips) * RS;
    const int rows = min(RS, Mc - p0);
    double acc[RS][DP];
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int k = 0; k < DP; ++k) acc[r][k] = 0.0;
    const int64_t nb = B > 0 ? B : 1;
    auto one_pair = [&](int64_t bb, int q, double (&dy)[DP], double (&wv)[RS]) {
        const int64_t b = B > 0 ? bb : a, p = B > 0 ? a * B + bb : a;
        const double s = scale ? (double)scale[p] : 1.0;
        const double *y = dYt + b * DP * ldy + q;
        const T *w = W + (p * Mc + p0) * ldw + q;
#pragma unroll
        for (int k = 0; k < DP; ++k) dy[k] = y[k * ldy];
        // unconditional loads from clamped rows (a predicated load gets its own basic block and its own s_waitcnt)
#pragma unroll
        for (int r = 0; r < RS; ++r) wv[r] = (double)w[(int64_t)min(r, rows - 1) * ldw];
#pragma unroll
        for (int r = 0; r < RS; ++r) wv[r] *= r < rows ? s : 0.0;
    };
    for (int q = threadIdx.x; q < Nc; q += NT) {
        int64_t bb = 0;
        for (; bb + 1 < nb; bb += 2) {
            double dy0[DP], w0[RS], dy1[DP], w1[RS];
            one_pair(bb, q, dy0, w0);
            one_pair(bb + 1, q, dy1, w1);
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int k = 0; k < DP; ++k) acc[r][k] = fma(w0[r], dy0[k], acc[r][k]);
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int k = 0; k < DP; ++k) acc[r][k] = fma(w1[r], dy1[k], acc[r][k]);
        }
        if (bb < nb) {
            double dy0[DP], w0[RS];
            one_pair(bb, q, dy0, w0);
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int k = 0; k < DP; ++k) acc[r][k] = fma(w0[r], dy0[k], acc[r][k]);
        }
    }
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            const double v = block_sum<NT>(acc[r][k], red);
            if (threadIdx.x == 0 && r < rows && k < D) Tout[(a * Mc + p0 + r) * (int64_t)D + k] = (T)v;
        }
}

// rbf: dL/dx[a,m,k] = (-2/sigma) sum_b s_ab sum_n dG[m,n] G[m,n] (x[a,m,k] - y[b,n,k]),
//      dG[m,n] = W[m-1,n-1] + W[m,n] - W[m-1,n] - W[m,n-1]   (transpose of the 4-corner difference)
// One block per (path a, RM consecutive node rows): thread = node column n; per pair b the thread loads its y node
// once for the RM rows, and RM + 1 rows of W (t_r = W[r][n] - W[r][n-1], dG[m,n] = t_m - t_{m-1}) instead of 4 W values
// per node -- the first version (one row per block) re-read every W row twice and every y node RM times more often.
template <typename T, int DMAX, int NT, int RM>
// (fp64, 17..32 dims: 260 VGPRs, one wave per SIMD.  Held to 256 -- two waves -- it is SLOWER, 12.1 -> 13.0 ms per 256 x 256 pairs of 64
// points at dim 20, profiles/r06_wide_dims.txt: the kernel is bound by what it re-reads from L2 -- every block streams all of Y -- not by latency)
__global__ __launch_bounds__(NT) void k_static_rbf_adj(const T *__restrict__ X, const T *__restrict__ Y,
                                                       const T *__restrict__ W, int64_t ldw, const T *__restrict__ scale,
                                                       int64_t B, int M, int N, int D, double inv_sigma, int row_groups,
                                                       T *__restrict__ gX) {
    __shared__ double red[NT / 64 + 1];
    const int Mc = M - 1, Nc = N - 1;
    const int64_t a = blockIdx.x / row_groups;
    const int m0 = (int)(blockIdx.x % row_groups) * RM;
    double xm[RM][DMAX], xs[RM], acc[RM][DMAX], cs[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        cs[r] = 0.0;
        const int m = min(m0 + r, M - 1);
        const T *x = X + (a * M + m) * (int64_t)D;
        xs[r] = 0.0;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            xm[r][k] = k < D ? (double)x[k] : 0.0;
            xs[r] = fma(xm[r][k], xm[r][k], xs[r]);
            acc[r][k] = 0.0;
        }
    }
    const int64_t nb = B > 0 ? B : 1;
    // everything one (pair, column) needs from memory, loaded without using it: two pairs are in flight per iteration
    auto fetch = [&](int64_t bb, int n, double (&yn)[DMAX], double (&wl)[RM + 1], double (&wv)[RM + 1], double &s) {
        const int64_t b = B > 0 ? bb : a, p = B > 0 ? a * B + bb : a;
        s = scale ? (double)scale[p] : 1.0;
        const T *y = Y + b * (int64_t)N * D;
        const T *w = W + p * (int64_t)Mc * ldw;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) yn[k] = k < D ? (double)y[(int64_t)n * D + k] : 0.0;
        const int nl = max(n - 1, 0), nr = min(n, Nc - 1);
#pragma unroll
        for (int r = 0; r <= RM; ++r) {   // unconditional loads from clamped positions, masked in use()
            const int64_t ro = (int64_t)min(max(m0 - 1 + r, 0), Mc - 1) * ldw;
            wl[r] = (double)w[ro + nl];
            wv[r] = (double)w[ro + nr];
        }
    };
    // Per node the contribution is c (x_r - y): with d = y - x_ref (x_ref = the block's first row, so |d| is the size of the
    // true difference and nothing cancels) it is accumulated as  cs[r] += c,  accd[r][k] += c d[k]  -- 9 operations instead
    // of 16 -- and put together after the loops:  sum c (x_r - y) = (x_r - x_ref) cs[r] - accd[r][k].
    auto use = [&](int n, const double (&yn)[DMAX], const double (&wl)[RM + 1], const double (&wv)[RM + 1], double s) {
        const bool lf = n >= 1, rt = n < Nc;
        double ys = 0.0, t[RM + 1], dk[DMAX];   // t[r] = W[r][n] - W[r][n-1] for W rows m0 - 1 + r (zero outside the matrix)
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            ys = fma(yn[k], yn[k], ys);
            dk[k] = yn[k] - xm[0][k];
        }
#pragma unroll
        for (int r = 0; r <= RM; ++r) {
            const int wr = m0 - 1 + r;
            const bool ok = wr >= 0 && wr < Mc;
            t[r] = ((ok && rt) ? wv[r] : 0.0) - ((ok && lf) ? wl[r] : 0.0);
        }
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            double xy = 0.0;
#pragma unroll
            for (int k = 0; k < DMAX; ++k) xy = fma(xm[r][k], yn[k], xy);
            // rbf: dist = -2 xy + (xs + ys) as the reference forms it (static_kernels.py:70-73); G = exp(-dist / sigma)
            const double g = exp_nonpos(-(fma(-2.0, xy, xs[r] + ys)) * inv_sigma);   // (a rounding-size positive argument is fine)
            const double c = (m0 + r < M ? s : 0.0) * (t[r + 1] - t[r]) * g;
            cs[r] += c;
#pragma unroll
            for (int k = 0; k < DMAX; ++k) acc[r][k] = fma(c, dk[k], acc[r][k]);
        }
    };
    for (int n = threadIdx.x; n < N; n += NT) {
        int64_t bb = 0;
        for (; bb + 1 < nb; bb += 2) {
            double y0[DMAX], l0[RM + 1], v0[RM + 1], s0, y1[DMAX], l1[RM + 1], v1[RM + 1], s1;
            fetch(bb, n, y0, l0, v0, s0);
            fetch(bb + 1, n, y1, l1, v1, s1);
            use(n, y0, l0, v0, s0);
            use(n, y1, l1, v1, s1);
        }
        if (bb < nb) {
            double y0[DMAX], l0[RM + 1], v0[RM + 1], s0;
            fetch(bb, n, y0, l0, v0, s0);
            use(n, y0, l0, v0, s0);
        }
    }
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            const double v = block_sum<NT>(fma(xm[r][k] - xm[0][k], cs[r], -acc[r][k]), red);
            if (threadIdx.x == 0 && k < D && m0 + r < M) gX[(a * M + m0 + r) * (int64_t)D + k] = (T)(-2.0 * inv_sigma * v);
        }
}

// ---- second-argument adjoints (Gram only): dL/dY from W, for the pairs (a, b) with b >= b0 ---------------------------
// Used for compute_Gram(X, X, sym=True) with a gradient: only the blocks on and above the diagonal are solved, and a pair
// (a, b) above the diagonal also stands for (b, a) -- whose first-argument gradient is this pair's second-argument one.
// The reference never differentiates its second argument (sigkernel.py:343, :412); these kernels exist for that shortcut.
// Thread = column (q or n), so the reads of W are coalesced along rows as in the first-argument kernels.

// linear: T2[b][q][k] = sum_a s_ab sum_p W[a,b,p,q] * dXr[a][p][k]   (dXr [A][Mrows][8]: scaled row differences, as the
// fused forward takes them);  the caller forms dL/dy[b][n] = T2[b][n-1] - T2[b][n].
// A block is NT columns x NW waves: the waves split the sum over a (wave w takes a = w, w + NW, ...) and add their partial
// sums through LDS in a fixed order, so the result does not depend on scheduling.  One wave per (b, column tile) left the
// chip with a few waves per CU, each serially dependent on its own loads (measured on the rbf kernel: 1.1 TB/s).
template <typename T, int NT, int NW>
__global__ __launch_bounds__(NT *NW) void k_linear_adj2(const double *__restrict__ dXr, int Mrows, const T *__restrict__ W,
                                                         int64_t ldw, const T *__restrict__ scale, int64_t A, int64_t B, int b0,
                                                         int Mc, int Nc, int D, int col_tiles, T *__restrict__ Tout) {
    constexpr int DP = 8;
    __shared__ double red[NW][DP][NT];
    const int wv = threadIdx.x / NT, col = threadIdx.x % NT;
    const int64_t b = b0 + blockIdx.x / col_tiles;
    const int q = (int)(blockIdx.x % col_tiles) * NT + col;
    const int qc = min(q, Nc - 1);
    double acc[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) acc[k] = 0.0;
    for (int64_t a = wv; a < A; a += NW) {
        const int64_t p = a * B + b;
        const double s = scale ? (double)scale[p] : 1.0;
        const T *w = W + p * Mc * ldw + qc;
        const double *dx = dXr + a * (int64_t)Mrows * DP;
        int i = 0;
        for (; i + 4 <= Mc; i += 4) {   // four rows in flight
            double wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = (double)w[(int64_t)(i + j) * ldw];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double c = s * wv[j];
#pragma unroll
                for (int k = 0; k < DP; ++k) acc[k] = fma(c, dx[(i + j) * DP + k], acc[k]);   // dx: wave-uniform
            }
        }
        for (; i < Mc; ++i) {
            const double c = s * (double)w[(int64_t)i * ldw];
#pragma unroll
            for (int k = 0; k < DP; ++k) acc[k] = fma(c, dx[i * DP + k], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < DP; ++k) red[wv][k][col] = acc[k];
    __syncthreads();
    if (wv == 0 && q < Nc) {
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            double v = red[0][k][col];
            for (int w = 1; w < NW; ++w) v += red[w][k][col];
            if (k < D) Tout[((b - b0) * Nc + q) * (int64_t)D + k] = (T)v;
        }
    }
}

// rbf: dL/dy[b][n][k] = (2/sigma) sum_a s_ab sum_m dG[m][n] G[m][n] (x[a][m][k] - y[b][n][k]),
//      dG[m][n] = t_m - t_{m-1},  t_m = W[m][n] - W[m][n-1] (zero outside the matrix)
template <typename T, int DMAX, int NT, int NW>
__global__ __launch_bounds__(NT *NW) void k_rbf_adj2(const T *__restrict__ X, const T *__restrict__ Y, const T *__restrict__ W,
                                                      int64_t ldw, const T *__restrict__ scale, int64_t A, int64_t B, int b0, int M,
                                                      int N, int D, double inv_sigma, int col_tiles, T *__restrict__ gY) {
    __shared__ double red[NW][DMAX][NT];   // see k_linear_adj2: the waves of a block split the sum over a
    const int wv = threadIdx.x / NT, col = threadIdx.x % NT;
    const int Mc = M - 1, Nc = N - 1;
    const int64_t b = b0 + blockIdx.x / col_tiles;
    const int n = (int)(blockIdx.x % col_tiles) * NT + col;
    const int nn = min(n, N - 1);
    const bool lf = nn >= 1, rt = nn < Nc;
    const int nl = max(nn - 1, 0), nr = min(nn, Nc - 1);
    double yn[DMAX], ys = 0.0, acc[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        yn[k] = k < D ? (double)Y[(b * N + nn) * (int64_t)D + k] : 0.0;
        ys = fma(yn[k], yn[k], ys);
        acc[k] = 0.0;
    }
    constexpr int MR = 4;   // node rows per iteration and path: their loads of W are issued before any is used
    constexpr int AU = 2;   // paths x_a in flight: independent exp chains (one wave per block leaves the SIMD little else)
    for (int64_t a0 = (int64_t)wv * AU; a0 < A; a0 += NW * AU) {
        double s[AU], tprev[AU];
        const T *w[AU], *x[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int64_t aa = a0 + u < A ? a0 + u : A - 1;
            const int64_t p = aa * B + b;
            s[u] = a0 + u < A ? (scale ? (double)scale[p] : 1.0) : 0.0;
            w[u] = W + p * Mc * ldw;
            x[u] = X + aa * (int64_t)M * D;
            tprev[u] = 0.0;
        }
        for (int m0 = 0; m0 < M; m0 += MR) {
            double wl[AU][MR], wv[AU][MR];
#pragma unroll
            for (int u = 0; u < AU; ++u)
#pragma unroll
                for (int j = 0; j < MR; ++j) {   // unconditional loads from clamped rows, masked below
                    const int64_t ro = (int64_t)min(m0 + j, Mc - 1) * ldw;
                    wl[u][j] = (double)w[u][ro + nl];
                    wv[u][j] = (double)w[u][ro + nr];
                }
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const int m = m0 + j, mm = min(m, M - 1);
#pragma unroll
                for (int u = 0; u < AU; ++u) {
                    const double tcur = m < Mc ? ((rt ? wv[u][j] : 0.0) - (lf ? wl[u][j] : 0.0)) : 0.0;
                    double xv[DMAX], xs = 0.0, xy = 0.0;
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) {
                        xv[k] = (double)x[u][(int64_t)mm * D + min(k, D - 1)];   // wave-uniform
                        xv[k] = k < D ? xv[k] : 0.0;
                        xs = fma(xv[k], xv[k], xs);
                        xy = fma(xv[k], yn[k], xy);
                    }
                    const double g = exp_nonpos(-(fma(-2.0, xy, xs + ys)) * inv_sigma);
                    const double c = (m < M ? s[u] : 0.0) * (tcur - tprev[u]) * g;
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) acc[k] = fma(c, xv[k] - yn[k], acc[k]);
                    tprev[u] = m < M ? tcur : tprev[u];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < DMAX; ++k) red[wv][k][col] = acc[k];
    __syncthreads();
    if (wv == 0 && n < N) {
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            double v = red[0][k][col];
            for (int w = 1; w < NW; ++w) v += red[w][k][col];
            if (k < D) gY[((b - b0) * N + n) * (int64_t)D + k] = (T)(2.0 * inv_sigma * v);
        }
    }
}

constexpr int ADJT_NMAX = 128;      // second paths the tiled adjoints below stage whole in LDS

// LINEAR static kernel, paths of 9..32 dims: T[a][m][k] = sum_b s_ab sum_n W[a, b][m][n] (y_b[n + 1][k] - y_b[n][k]) -- until round 6 a batched
// library GEMM per column block of Y plus three elementwise passes over its (A, b, Mc, D) products (0.9 + 0.9 ms of a 5.3 ms gradient step at
// 256 x 256 pairs of 64 points and 20 dims, profiles/r06_api_profile.txt).  Here every element of W is read ONCE and nothing else touches
// HBM: a block owns NW * RM rows of W[a, .] (RM per wave, a lane per POINT n of y_b, coefficient W[m][n - 1] - W[m][n] -- the
// neighbour's value by DPP), its 64 NW threads copy y_b to LDS (rows of DMAX + 2 doubles, read two at a time) and each value read from there feeds RM
// accumulators.  Pairs go in CHUNKS of PB: a pair is ~0.15 us of arithmetic per wave against ~2 us for a load from HBM, and with the
// barriers a block has nothing else to hide its loads behind -- so the whole NEXT chunk's W values and y points are loaded into registers
// before this chunk's arithmetic and consumed after it (one pair ahead: 2.0 ms at 256 x 256 pairs of 64 points and 20 dims, bound by
// that latency; the GEMM route it replaces 1.8).  (Rounding: sum_n (W[n - 1] - W[n]) y[n] loses digits against the GEMM's sum_n W[n] dy[n] in
// proportion to |y| / |dy| -- 1e-12 relative for a path 1e4 increments away from the origin; centring y_b at its first point would undo
// that at the price of the 16-dim instance's 128 registers, 0.61 -> 0.75 ms: not done.)
__device__ __forceinline__ double lane_shr1(double v, double lane0) {      // lane l <- lane l - 1; lane 0 <- its own `lane0`
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63_of(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <typename T, int DMAX, int RM, int NP, int PB, int NW>
__global__ __launch_bounds__(64 * NW) void k_static_linear_adj_tiled(const T *__restrict__ Y, const T *__restrict__ W, int64_t ldw,
                                                                          const T *__restrict__ scale, int64_t B, int M, int N, int D,
                                                                          int row_groups, T *__restrict__ Tout) {
    constexpr int DS = DMAX + 2;      // 16-byte rows for ds_read_b128: 2 DS = 4 (mod 8) dwords, the 16 lanes of a read group on 16 different bank quads
    constexpr int SREG = NP * 64 * DMAX / (64 * NW);         // values of one y_b a thread stages, at most (N <= 64 NP)
    extern __shared__ __attribute__((aligned(16))) double ysh[];      // [PB][N][DS], columns D..DMAX-1 zero
    const int Mc = M - 1, Nc = N - 1;
    const int64_t a = blockIdx.x / row_groups;
    const int wv_id = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = ((int)(blockIdx.x % row_groups) * NW + wv_id) * RM;      // this wave's first row of W (>= Mc: it only keeps the barriers)
    const bool rows_ok = m0 < Mc;
    double acc[RM][DMAX];
#pragma unroll
    for (int j = 0; j < RM; ++j)
#pragma unroll
        for (int k = 0; k < DMAX; ++k) acc[j][k] = 0.0;
    for (int i = threadIdx.x; i < PB * N * DS; i += 64 * NW) ysh[i] = 0.0;
    const int64_t nb = B > 0 ? B : 1;
    const int nd = N * D;
    int soff[SREG];      // where this thread's r-th value of a y_b goes in its LDS image (-1: none)
#pragma unroll
    for (int r = 0; r < SREG; ++r) {
        const int i = threadIdx.x + r * 64 * NW, n = i / D;
        soff[r] = i < nd ? n * DS + (i - n * D) : -1;
    }
    // the NEXT chunk of PB pairs, in registers while this one is worked on: its points of y (this thread's share), its W values
    // W[m0 + j][lane + 64 pp] -- raw: nothing may wait for these loads before the arithmetic below
    T yq[PB][SREG], wq[PB][NP][RM];
    auto loads = [&](int64_t b0) {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int64_t bb = b0 + u;
            if (bb >= nb) break;
            const T *y = Y + (B > 0 ? bb : a) * (int64_t)N * D;
#pragma unroll
            for (int r = 0; r < SREG; ++r) yq[u][r] = y[min((int)threadIdx.x + r * 64 * NW, nd - 1)];
            if (rows_ok) {
                const int64_t p = B > 0 ? a * B + bb : a;
                const T *w = W + p * (int64_t)Mc * ldw;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    const int n = lane + 64 * pp;
#pragma unroll
                    for (int j = 0; j < RM; ++j)      // (rows past Mc: never stored; columns past Nc: masked where the value is used --
                        wq[u][pp][j] = w[(int64_t)min(m0 + j, Mc - 1) * ldw + min(n, Nc - 1)];      // nothing here may wait for a load)
                }
            }
        }
    };
    loads(0);
    for (int64_t b0 = 0; b0 < nb; b0 += PB) {
        __syncthreads();      // every wave is done with the previous chunk's points (the first time: the zero fill above)
        double wc[PB][NP][RM], sc[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (b0 + u >= nb) break;
#pragma unroll
            for (int r = 0; r < SREG; ++r)
                if (soff[r] >= 0) ysh[u * N * DS + soff[r]] = (double)yq[u][r];
            sc[u] = scale ? (double)scale[B > 0 ? a * B + b0 + u : a] : 1.0;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                for (int j = 0; j < RM; ++j) wc[u][pp][j] = lane + 64 * pp < Nc ? (double)wq[u][pp][j] : 0.0;
        }
        __syncthreads();
        if (b0 + PB < nb) loads(b0 + PB);
        if (rows_ok) {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                if (b0 + u >= nb) break;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    if (64 * pp >= N) break;
                    double c[RM];
#pragma unroll
                    for (int j = 0; j < RM; ++j) {
                        const double left = lane_shr1(wc[u][pp][j], pp ? lane63_of(wc[u][pp ? pp - 1 : 0][j]) : 0.0);
                        c[j] = sc[u] * (left - wc[u][pp][j]);      // 0 from point N on
                    }
                    const double *yr = ysh + u * N * DS + min(lane + 64 * pp, N - 1) * DS;
#pragma unroll
                    for (int k = 0; k < DMAX; k += 2) {
                        const double2 yk = *reinterpret_cast<const double2 *>(yr + k);
#pragma unroll
                        for (int j = 0; j < RM; ++j) {
                            acc[j][k] = fma(c[j], yk.x, acc[j][k]);
                            acc[j][k + 1] = fma(c[j], yk.y, acc[j][k + 1]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RM; ++j)
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            double v = acc[j][k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && k < D && m0 + j < Mc) Tout[(a * Mc + m0 + j) * (int64_t)D + k] = (T)v;
        }
}

template <typename T, int DMAX, int NP, int NW>
int launch_static_linear_adj_tiled_np(const T *Y, const T *W, int64_t ldw, const T *scale, int64_t A, int64_t B, int M, int N, int D, T *out,
                                      hipStream_t s) {
    // RM rows per wave and PB pairs per chunk: what the registers hold beside the RM * DMAX sums -- 16 dims fit 128 registers (two blocks
    // per CU) with PB = 2, and that beats longer chunks (0.61 ms against 0.75 at 256 x 256 pairs of 64 points, profiles/r06_lin_adj.txt);
    // beyond, one block per CU whatever PB is
    constexpr int RM = 2;
    constexpr int PB = DMAX <= 16 ? 2 : DMAX <= 24 ? 4 / NP : 2 / NP;
    const int rgs = (M - 1 + NW * RM - 1) / (NW * RM);
    const int64_t blk = A * rgs;
    if (blk > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    const size_t lds = sizeof(double) * PB * (size_t)N * (DMAX + 2);
    auto kern = k_static_linear_adj_tiled<T, DMAX, RM, NP, PB, NW>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SK_LAUNCH(kern, dim3((unsigned)blk), dim3(64 * NW), lds, s, Y, W, ldw, scale, B, M, N, D, rgs, out);
    return check_launch();
}
template <typename T, int DMAX>
int launch_static_linear_adj_tiled(const T *Y, const T *W, int64_t ldw, const T *scale, int64_t A, int64_t B, int M, int N, int D, T *out,
                                   hipStream_t s) {
    if (N > ADJT_NMAX) return SK_ERR_UNSUPPORTED;      // (the caller's batched GEMM)
    if (N <= 64) return launch_static_linear_adj_tiled_np<T, DMAX, 1, 8>(Y, W, ldw, scale, A, B, M, N, D, out, s);
    return launch_static_linear_adj_tiled_np<T, DMAX, 2, 8>(Y, W, ldw, scale, A, B, M, N, D, out, s);
}

// k_static_rbf_adj for paths of 17..32 dims (one node row per wave there), with the y points of a pair shared by the EIGHT waves of a block
// through LDS: in k_static_rbf_adj every (a, row) block streams all of Y from L2 -- 43 GB per launch at 256 x 256 pairs of 64 points and
// 20 dims, 7.3 ms, 54 % of that gradient step (profiles/r06_api_profile.txt).  The linear kernel's schedule (above): pairs in chunks of PB
// whose W values (rows m - 1 and m, raw) and y points are loaded a chunk ahead -- pair by pair with a barrier each, loads waited for where
// they were issued, the first tiled form took 4.5 ms there, this one 2.4 --, the left neighbour's W by DPP instead of a second load, y_b
// in 16-byte rows read twice (once for the exponent, once for the sums: the differences y - x_m are not kept, which is what lets the
// queue into the registers).  Operand for operand the arithmetic of k_static_rbf_adj: bit-identical results.
template <typename T, int DMAX, int NP, int PB, int NW>
__global__ __launch_bounds__(64 * NW) void k_static_rbf_adj_tiled(const T *__restrict__ X, const T *__restrict__ Y, const T *__restrict__ W, int64_t ldw,
                                                                    const T *__restrict__ scale, int64_t B, int M, int N, int D, double inv_sigma,
                                                                    int row_groups, T *__restrict__ gX) {
    constexpr int DS = DMAX + 2;
    constexpr int SREG = NP * 64 * DMAX / (64 * NW);
    extern __shared__ __attribute__((aligned(16))) double ysh[];      // [PB][N][DS], columns D..DMAX-1 zero
    const int Mc = M - 1, Nc = N - 1;
    const int64_t a = blockIdx.x / row_groups;
    const int wv_id = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = (int)(blockIdx.x % row_groups) * NW + wv_id;      // this wave's node row (>= M: it only keeps the barriers)
    const bool row_ok = m < M;
    const int mr = min(m, M - 1);
    const bool ok0 = m - 1 >= 0 && m - 1 < Mc, ok1 = m < Mc;      // W rows m - 1, m inside the matrix
    const int64_t ro0 = (int64_t)min(max(m - 1, 0), Mc - 1) * ldw, ro1 = (int64_t)min(max(m, 0), Mc - 1) * ldw;
    double xm[DMAX], acc[DMAX], xs = 0.0;
    {
        const T *x = X + (a * M + mr) * (int64_t)D;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            xm[k] = k < D ? (double)x[k] : 0.0;
            xs = fma(xm[k], xm[k], xs);
            acc[k] = 0.0;
        }
    }
    for (int i = threadIdx.x; i < PB * N * DS; i += 64 * NW) ysh[i] = 0.0;
    const int64_t nb = B > 0 ? B : 1;
    const int nd = N * D;
    int soff[SREG];
#pragma unroll
    for (int r = 0; r < SREG; ++r) {
        const int i = threadIdx.x + r * 64 * NW, n = i / D;
        soff[r] = i < nd ? n * DS + (i - n * D) : -1;
    }
    T yq[PB][SREG], wq[PB][NP][2];
    auto loads = [&](int64_t b0) {      // (nothing here may wait for a load: the values are used a chunk later)
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int64_t bb = b0 + u;
            if (bb >= nb) break;
            const T *y = Y + (B > 0 ? bb : a) * (int64_t)N * D;
#pragma unroll
            for (int r = 0; r < SREG; ++r) yq[u][r] = y[min((int)threadIdx.x + r * 64 * NW, nd - 1)];
            if (row_ok) {
                const T *w = W + (B > 0 ? a * B + bb : a) * (int64_t)Mc * ldw;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    const int nc = min(lane + 64 * pp, Nc - 1);
                    wq[u][pp][0] = w[ro0 + nc];
                    wq[u][pp][1] = w[ro1 + nc];
                }
            }
        }
    };
    loads(0);
    for (int64_t b0 = 0; b0 < nb; b0 += PB) {
        __syncthreads();
        double wc[PB][NP][2], sc[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (b0 + u >= nb) break;
#pragma unroll
            for (int r = 0; r < SREG; ++r)
                if (soff[r] >= 0) ysh[u * N * DS + soff[r]] = (double)yq[u][r];
            sc[u] = scale ? (double)scale[B > 0 ? a * B + b0 + u : a] : 1.0;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const bool rt = lane + 64 * pp < Nc;
                wc[u][pp][0] = (ok0 && rt) ? (double)wq[u][pp][0] : 0.0;
                wc[u][pp][1] = (ok1 && rt) ? (double)wq[u][pp][1] : 0.0;
            }
        }
        __syncthreads();
        if (b0 + PB < nb) loads(b0 + PB);
        if (row_ok) {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                if (b0 + u >= nb) break;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    if (64 * pp >= N) break;
                    double t[2];
#pragma unroll
                    for (int r = 0; r < 2; ++r)      // t_r = W[r][n] - W[r][n - 1], zero outside the matrix
                        t[r] = wc[u][pp][r] - lane_shr1(wc[u][pp][r], pp ? lane63_of(wc[u][pp ? pp - 1 : 0][r]) : 0.0);
                    const double *yr = ysh + u * N * DS + min(lane + 64 * pp, N - 1) * DS;
                    double ys = 0.0, xy = 0.0;
#pragma unroll
                    for (int k = 0; k < DMAX; k += 2) {
                        const double2 yk = *reinterpret_cast<const double2 *>(yr + k);
                        ys = fma(yk.x, yk.x, ys);
                        xy = fma(xm[k], yk.x, xy);
                        ys = fma(yk.y, yk.y, ys);
                        xy = fma(xm[k + 1], yk.y, xy);
                    }
                    const double g = exp_nonpos(-(fma(-2.0, xy, xs + ys)) * inv_sigma);
                    const double c = sc[u] * (t[1] - t[0]) * g;
                    asm volatile("" ::: "memory");      // (read y_b[n] again: held across the exponential it costs DMAX registers)
#pragma unroll
                    for (int k = 0; k < DMAX; k += 2) {
                        const double2 yk = *reinterpret_cast<const double2 *>(yr + k);
                        acc[k] = fma(c, yk.x - xm[k], acc[k]);
                        acc[k + 1] = fma(c, yk.y - xm[k + 1], acc[k + 1]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        double v = -acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && k < D && row_ok) gX[(a * M + m) * (int64_t)D + k] = (T)(-2.0 * inv_sigma * v);
    }
}

template <typename T, int DMAX, int NP>
int launch_static_rbf_adj_tiled(double param, const T *X, const T *Y, const T *W, int64_t ldw, const T *scale, int64_t A, int64_t B, int M,
                                  int N, int D, T *out, hipStream_t s) {
    constexpr int NW = 8;
    constexpr int PB = DMAX > 24 && NP > 1 ? 1 : 2;
    const int rgs = (M + NW - 1) / NW;
    const int64_t blk = A * rgs;
    if (blk > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    const size_t lds = sizeof(double) * PB * (size_t)N * (DMAX + 2);
    auto kern = k_static_rbf_adj_tiled<T, DMAX, NP, PB, NW>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SK_LAUNCH(kern, dim3((unsigned)blk), dim3(64 * NW), lds, s, X, Y, W, ldw, scale, B, M, N, D, 1.0 / param, rgs, out);
    return check_launch();
}

template <typename T, int DMAX, int NT>
int launch_static_adj_d(int kind, double param, const T *X, const T *Y, const T *W, int64_t ldw, const T *scale, int64_t A,
                        int64_t B, int M, int N, int D, T *out, hipStream_t s) {
    // (kind 0, the LINEAR static kernel: its contraction T[a] = sum_b W[a, b] dY[b] runs from pre-differenced, dimension-major paths in
    // sk_linear_adjoint_* for dim <= 8 and in k_static_linear_adj_tiled for 9..32 dims, see launch_static_adjoint)
    if (kind == 0) {
        return SK_ERR_UNSUPPORTED;
    } else {
        constexpr int RM = DMAX <= 8 ? 4 : DMAX == 16 ? 2 : 1;   // node rows per block (RM * DMAX accumulators per thread)
        const int row_groups = (M + RM - 1) / RM;
        const int64_t blocks = A * row_groups;
        if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
        SK_LAUNCH((k_static_rbf_adj<T, DMAX, NT, RM>), dim3((unsigned)blocks), dim3(NT), 0, s, X, Y, W, ldw, scale, B,
                           M, N, D, 1.0 / param, row_groups, out);
    }
    return check_launch();
}

template <typename T, int DMAX>
int launch_static_adj_nt(int kind, double param, const T *X, const T *Y, const T *W, int64_t ldw, const T *scale, int64_t A,
                         int64_t B, int M, int N, int D, T *out, hipStream_t s) {
    if constexpr (DMAX >= 24) {     // 17..32 dims: the y points of a pair through LDS, shared by eight rows' waves (k_static_rbf_adj_tiled;
        if (kind == 1 && N <= ADJT_NMAX) {      // at 16 dims it is no faster than two rows per thread: 1.87 / 1.66 against 1.80 ms, r06_lin_adj.txt)
            if (N <= 64) return launch_static_rbf_adj_tiled<T, DMAX, 1>(param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
            return launch_static_rbf_adj_tiled<T, DMAX, 2>(param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
        }
        return launch_static_adj_d<T, DMAX, 128>(kind, param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
    } else {
        if (N <= 80) return launch_static_adj_d<T, DMAX, 64>(kind, param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
        return launch_static_adj_d<T, DMAX, 128>(kind, param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
    }
}

template <typename T, int DMAX>
int launch_static_d(int kind, double param, const T *X, const T *Y, int64_t A, int64_t B, int M, int N, int D, T *inc,
                    int64_t ld, hipStream_t s) {
    const int64_t P = B > 0 ? A * B : A;
    if (kind == 0) {
        if constexpr (DMAX >= 16) {
            const int rts = (M - 1 + 63) / 64, cts = (int)((ld + 63) / 64);
            const int64_t blocks = P * rts * cts;
            if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
            SK_LAUNCH((k_static_linear_mfma<T, DMAX>), dim3((unsigned)blocks), dim3(64), 0, s, X, Y, B, M, N, D, param * param, inc, ld, rts, cts);
        } else if (ld <= SK_TPB) {
            if (P > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
            SK_LAUNCH((k_static_linear<T, DMAX, 1>), dim3((unsigned)P), dim3(SK_TPB), 0, s, X, Y, B, M, N, D, param * param, inc, ld, 1);
        } else {
            const int col_tiles = (int)((ld + SK_TPB * 2 - 1) / (SK_TPB * 2));
            const int64_t blocks = P * col_tiles;
            if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
            SK_LAUNCH((k_static_linear<T, DMAX, 2>), dim3((unsigned)blocks), dim3(SK_TPB), 0, s, X, Y, B, M, N, D,
                               param * param, inc, ld, col_tiles);
        }
    } else {
        const T z = (T)0;
        if (ld <= SK_TPB) {
            const int col_tiles = 1;
            const int64_t blocks = P;
            if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
            SK_LAUNCH((k_static_nodes<T, DMAX, 1, 1, 1>), dim3((unsigned)blocks), dim3(SK_TPB), 0, s, X, X, X, Y, B, M,
                               N, D, 1.0 / param, z, z, z, inc, (T *)nullptr, (T *)nullptr, ld, col_tiles);
        } else {
            const int col_tiles = (int)((ld + 2 * SK_TPB - 1) / (2 * SK_TPB));
            const int64_t blocks = P * col_tiles;
            if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
            SK_LAUNCH((k_static_nodes<T, DMAX, 1, 1, 2>), dim3((unsigned)blocks), dim3(SK_TPB), 0, s, X, X, X, Y, B, M,
                               N, D, 1.0 / param, z, z, z, inc, (T *)nullptr, (T *)nullptr, ld, col_tiles);
        }
    }
    return check_launch();
}

template <typename T, int DMAX>
int launch_static_deriv_d(int kind, double param, const T *X0, const T *X1, const T *X2, const T *Y, int64_t A, int64_t B,
                          int M, int N, int D, double eps, T *inc, T *inc_d, T *inc_dd, int64_t ld, hipStream_t s) {
    const T c1 = (T)(1. / eps), c2 = (T)(2. / eps), c3 = (T)(1. / (eps * eps));
    const double inv_sigma = kind == 0 ? 0.0 : 1.0 / param;
    const bool narrow = ld <= SK_TPB;
    const int col_tiles = narrow ? 1 : (int)((ld + 2 * SK_TPB - 1) / (2 * SK_TPB));
    const int64_t blocks = A * B * col_tiles;
    if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
#define SK_LAUNCH_NODES(KIND, CPT)                                                                                      \
    SK_LAUNCH((k_static_nodes<T, DMAX, KIND, 3, CPT>), dim3((unsigned)blocks), dim3(SK_TPB), 0, s, X0, X1, X2, Y, B, \
                       M, N, D, inv_sigma, c1, c2, c3, inc, inc_d, inc_dd, ld, col_tiles)
    if (kind == 0) {
        if (narrow) SK_LAUNCH_NODES(0, 1); else SK_LAUNCH_NODES(0, 2);
    } else {
        if (narrow) SK_LAUNCH_NODES(1, 1); else SK_LAUNCH_NODES(1, 2);
    }
#undef SK_LAUNCH_NODES
    return check_launch();
}

}  // namespace

template <typename T>
int launch_static_deriv_increments(int kind, double param, const T *X0, const T *X1, const T *X2, const T *Y, int64_t A,
                                   int64_t B, int M, int N, int D, double eps, T *inc, T *inc_d, T *inc_dd, int64_t ld,
                                   hipStream_t s) {
    if (D <= 4) return launch_static_deriv_d<T, 4>(kind, param, X0, X1, X2, Y, A, B, M, N, D, eps, inc, inc_d, inc_dd, ld, s);
    if (D <= 8) return launch_static_deriv_d<T, 8>(kind, param, X0, X1, X2, Y, A, B, M, N, D, eps, inc, inc_d, inc_dd, ld, s);
    if (D <= 16) return launch_static_deriv_d<T, 16>(kind, param, X0, X1, X2, Y, A, B, M, N, D, eps, inc, inc_d, inc_dd, ld, s);
    if (D <= 32) return launch_static_deriv_d<T, 32>(kind, param, X0, X1, X2, Y, A, B, M, N, D, eps, inc, inc_d, inc_dd, ld, s);
    return SK_ERR_UNSUPPORTED;
}
template int launch_static_deriv_increments<double>(int, double, const double *, const double *, const double *, const double *,
                                                    int64_t, int64_t, int, int, int, double, double *, double *, double *,
                                                    int64_t, hipStream_t);
template int launch_static_deriv_increments<float>(int, double, const float *, const float *, const float *, const float *,
                                                   int64_t, int64_t, int, int, int, double, float *, float *, float *, int64_t,
                                                   hipStream_t);

template <typename T>
int launch_static_increments(int kind, double param, const T *X, const T *Y, int64_t A, int64_t B, int M, int N, int D,
                             T *inc, int64_t ld, hipStream_t s) {
    if (D <= 4) return launch_static_d<T, 4>(kind, param, X, Y, A, B, M, N, D, inc, ld, s);
    if (D <= 8) return launch_static_d<T, 8>(kind, param, X, Y, A, B, M, N, D, inc, ld, s);
    if (D <= 16) return launch_static_d<T, 16>(kind, param, X, Y, A, B, M, N, D, inc, ld, s);
    if (D <= 24) return launch_static_d<T, 24>(kind, param, X, Y, A, B, M, N, D, inc, ld, s);   // (lead-lag of 8..11 dims + time: 17..23)
    if (D <= 32) return launch_static_d<T, 32>(kind, param, X, Y, A, B, M, N, D, inc, ld, s);
    return SK_ERR_UNSUPPORTED;   // wide paths: the caller uses the generic static kernel + sk_increments
}

// out: kind 0 -> T [A, M-1, D] (the caller differences it along M and applies scale^2); kind 1 -> dL/dX [A, M, D]
template <typename T>
int launch_static_adjoint(int kind, double param, const T *X, const T *Y, const T *W, int64_t ldw, const T *scale, int64_t A,
                          int64_t B, int M, int N, int D, T *out, hipStream_t s) {
    if (kind == 0) {
        if (D <= 8) return SK_ERR_UNSUPPORTED;      // sk_linear_adjoint_*
        if (D <= 16) return launch_static_linear_adj_tiled<T, 16>(Y, W, ldw, scale, A, B, M, N, D, out, s);
        if (D <= 24) return launch_static_linear_adj_tiled<T, 24>(Y, W, ldw, scale, A, B, M, N, D, out, s);
        if (D <= 32) return launch_static_linear_adj_tiled<T, 32>(Y, W, ldw, scale, A, B, M, N, D, out, s);
        return SK_ERR_UNSUPPORTED;
    }
    if (D <= 4) return launch_static_adj_nt<T, 4>(kind, param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
    if (D <= 8) return launch_static_adj_nt<T, 8>(kind, param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
    if (D <= 16) return launch_static_adj_nt<T, 16>(kind, param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
    if (D <= 24) return launch_static_adj_nt<T, 24>(kind, param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);   // (lead-lag of 8..11 dims + time)
    if (D <= 32) return launch_static_adj_nt<T, 32>(kind, param, X, Y, W, ldw, scale, A, B, M, N, D, out, s);
    return SK_ERR_UNSUPPORTED;
}

// out: T [A, Mc, D] (the caller differences it along the path and applies scale^2)
template <typename T>
int launch_linear_adjoint_dyt(const double *dYt, int64_t ldy, const T *W, int64_t ldw, const T *scale, int64_t A, int64_t B,
                              int Mc, int Nc, int D, T *out, hipStream_t s) {
    if (D > 8) return SK_ERR_UNSUPPORTED;
    const int strips = (Mc + 7) / 8;
    const int64_t blocks = A * strips;
    if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    if (Nc <= 80)
        SK_LAUNCH((k_linear_adj_dyt<T, 64>), dim3((unsigned)blocks), dim3(64), 0, s, dYt, ldy, W, ldw, scale, B, Mc, Nc, D,
                           strips, out);
    else
        SK_LAUNCH((k_linear_adj_dyt<T, 128>), dim3((unsigned)blocks), dim3(128), 0, s, dYt, ldy, W, ldw, scale, B, Mc, Nc,
                           D, strips, out);
    return check_launch();
}
template int launch_linear_adjoint_dyt<double>(const double *, int64_t, const double *, int64_t, const double *, int64_t,
                                               int64_t, int, int, int, double *, hipStream_t);
template int launch_linear_adjoint_dyt<float>(const double *, int64_t, const float *, int64_t, const float *, int64_t, int64_t,
                                              int, int, int, float *, hipStream_t);

// kind 0: out = T2 [B - b0, Nc, D] from dXr (aux); kind 1: out = dL/dY [B - b0, N, D]
template <typename T>
int launch_static_adjoint2(int kind, double param, const T *X, const T *Y, const double *dXr, int Mrows, const T *W, int64_t ldw,
                           const T *scale, int64_t A, int64_t B, int b0, int M, int N, int D, T *out, hipStream_t s) {
    const int64_t nbk = B - b0;
    if (nbk <= 0) return SK_OK;
    if (kind == 0) {
        if (D > 8 || !dXr) return SK_ERR_UNSUPPORTED;
        const int Nc = N - 1;
        const int col_tiles = (Nc + 63) / 64;
        if (nbk * col_tiles > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
        SK_LAUNCH((k_linear_adj2<T, 64, 8>), dim3((unsigned)(nbk * col_tiles)), dim3(64 * 8), 0, s, dXr, Mrows, W, ldw,
                           scale, A, B, b0, M - 1, Nc, D, col_tiles, out);
        return check_launch();
    }
    const int col_tiles = (N + 63) / 64;
    if (nbk * col_tiles > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    // 32 KB of LDS for the partial sums whatever the dimension: 8 waves up to dim 8, 4 at 16, 2 at 32
#define SK_RBF2(DM, NW)                                                                                                  \
    SK_LAUNCH((k_rbf_adj2<T, DM, 64, NW>), dim3((unsigned)(nbk * col_tiles)), dim3(64 * NW), 0, s, X, Y, W, ldw, scale, \
                       A, B, b0, M, N, D, 1.0 / param, col_tiles, out)
    if (D <= 4) SK_RBF2(4, 8);
    else if (D <= 8) SK_RBF2(8, 8);
    else if (D <= 16) SK_RBF2(16, 4);
    else if (D <= 32) SK_RBF2(32, 2);
    else return SK_ERR_UNSUPPORTED;
#undef SK_RBF2
    return check_launch();
}
template int launch_static_adjoint2<double>(int, double, const double *, const double *, const double *, int, const double *,
                                            int64_t, const double *, int64_t, int64_t, int, int, int, int, double *, hipStream_t);
template int launch_static_adjoint2<float>(int, double, const float *, const float *, const double *, int, const float *, int64_t,
                                           const float *, int64_t, int64_t, int, int, int, int, float *, hipStream_t);

template int launch_static_adjoint<double>(int, double, const double *, const double *, const double *, int64_t,
                                           const double *, int64_t, int64_t, int, int, int, double *, hipStream_t);
template int launch_static_adjoint<float>(int, double, const float *, const float *, const float *, int64_t, const float *,
                                          int64_t, int64_t, int, int, int, float *, hipStream_t);

template int launch_static_increments<double>(int, double, const double *, const double *, int64_t, int64_t, int, int, int,
                                              double *, int64_t, hipStream_t);
template int launch_static_increments<float>(int, double, const float *, const float *, int64_t, int64_t, int, int, int,
                                             float *, int64_t, hipStream_t);

}  // namespace sk
