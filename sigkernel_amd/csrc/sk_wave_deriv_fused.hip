// sk_wave_deriv_fused.hip -- the signature kernel and its first and second directional derivatives (K, K_gamma, K_gamma_gamma) in
// one sweep WITH the static kernel fused in: the three increment arrays of k_kgrad (sigkernel.py:526-541) are formed inside the
// solver from the paths -- x, x + eps gamma, x + 2 eps gamma and y -- and never exist in HBM (3 x 8.7 GB per 65 536 pairs of
// length 128).  Replaces sk_static_deriv_increments_* + sk_solve_deriv_* (cuda_backend.py:165-223, sigkernel.py:526-566) for
// LinearKernel and RBFKernel.
//
// Stream and band boundary: sk_wave_fused_mb.hip (one pair per 64-lane wave, band after band, the bottom lane's last fine row and
// node pair through a per-wave row in L2, LDS-DMA one window ahead, work queue).  Sweep: sk_wave_deriv.hip (three PDE states, six
// coefficients per coarse cell, one coarse row per lane).  Nodes "from above" as in sk_wave_fused_mb.hip, for BOTH static kernels
// (the finite differences are taken on node VALUES): a lane evaluates the bottom node row of its coarse row one unit ahead of the
// sweep, for the three shifted paths, and takes the row above from the lane above.
//
// Arithmetic of the increments: exactly k_static_nodes<NV = 3> (sk_static.hip) -- the same dot-product chains, the library exp, each
// node value scaled like the reference (-(1/eps) G0, (1/eps) G1, -(1/eps)(-(1/eps) G0), -(2/eps)((1/eps) G1), (1/eps^2) G2), every
// scaled array 4-corner-differenced as ((G11 + G00) - G10) - G01 and the differences added left to right, without FMA
// contraction: the sums cancel eight orders of magnitude and their rounding is part of the result the fixtures pin.
// Scope: fp64, dyadic <= 2, path dim <= 8, any M, second path of N >= 126 points (rows of >= 64 units).
#include "sk_wave_common.h"

namespace sk {
namespace {

constexpr int DF_L = WAVE;
constexpr int DF_X_SLOTS = 2;

struct DerivFusedParams {
    const double *Xr[3];   // [A][Mrows][FD] points of x, x + eps gamma, x + 2 eps gamma, zero rows / dims beyond the path
    const double *Yt;      // [Bn][FD][Ncp] points of y, dimension-major, zero-padded
    double *out[3];        // [P] each
    double *ws;            // per wave: [NUp + 8][E] band-boundary row + the constant chunk of band 0
    int64_t P, B;
    int Mrows, Ncp, Mc, Nc, NUp, nb;
    int u_f, lam_f, band_f, sel_f;
    double inv_sigma, c1, c2, c3;   // 1/eps, 2/eps, 1/eps^2
    int64_t ws_stride;
    WaveGroup wg;
    unsigned long long *queue;
    int64_t q_first;
    int C0;
};

__device__ __forceinline__ void df_store_through(double *p, d2_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void df_begin(d2_t &t) { asm volatile("" : "=v"(t)); }
template <int OFF>
__device__ __forceinline__ void df_read_pend(d2_t &t, unsigned a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(t) : "v"(a), "n"(OFF) : "memory");
}
__device__ __forceinline__ void df_take(d2_t &o, d2_t &t) { asm volatile("" : "=v"(o) : "0"(t)); }
// pieces [I0, I0 + N) of an entry from `a` (+ 16 (i - I0) bytes), no wait
template <int I0, int N, int NP>
__device__ __forceinline__ void df_read_pieces(d2_t (&t)[NP], unsigned a) {
    static_assert(N == 3 || N == 6 || N == 12, "");
    df_read_pend<0>(t[I0], a); df_read_pend<16>(t[I0 + 1], a); df_read_pend<32>(t[I0 + 2], a);
    if constexpr (N > 3) { df_read_pend<48>(t[I0 + 3], a); df_read_pend<64>(t[I0 + 4], a); df_read_pend<80>(t[I0 + 5], a); }
    if constexpr (N > 6) {
        df_read_pend<96>(t[I0 + 6], a); df_read_pend<112>(t[I0 + 7], a); df_read_pend<128>(t[I0 + 8], a); df_read_pend<144>(t[I0 + 9], a);
        df_read_pend<160>(t[I0 + 10], a); df_read_pend<176>(t[I0 + 11], a);
    }
}
template <int NP>
__device__ __forceinline__ void df_read_entry(d2_t (&t)[NP], unsigned a) {
#pragma unroll
    for (int i = 0; i < NP; ++i) df_begin(t[i]);
    df_read_pend<0>(t[0], a); df_read_pend<16>(t[1], a); df_read_pend<32>(t[2], a); df_read_pend<48>(t[3], a);
    df_read_pend<64>(t[4], a); df_read_pend<80>(t[5], a);
    if constexpr (NP > 6) { df_read_pend<96>(t[6], a); df_read_pend<112>(t[7], a); df_read_pend<128>(t[8], a); }
    if constexpr (NP > 9) {
        df_read_pend<144>(t[9], a); df_read_pend<160>(t[10], a); df_read_pend<176>(t[11], a); df_read_pend<192>(t[12], a);
        df_read_pend<208>(t[13], a); df_read_pend<224>(t[14], a);
    }
}

template <int ND>
__device__ __forceinline__ void df_read_ydims(d2_t (&v)[ND], unsigned a_even, unsigned a_odd) {
    static_assert(ND == 8 || ND == 16, "");
    if constexpr (ND == 8) {
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %8 offset:256\n\tds_read_b128 %3, %9 offset:256\n\t"
                     "ds_read_b128 %4, %8 offset:512\n\tds_read_b128 %5, %9 offset:512\n\tds_read_b128 %6, %8 offset:768\n\t"
                     "ds_read_b128 %7, %9 offset:768\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(a_even), "v"(a_odd) : "memory");
    } else {
        asm volatile("ds_read_b128 %0, %16\n\tds_read_b128 %1, %17\n\t"
                     "ds_read_b128 %2, %16 offset:256\n\tds_read_b128 %3, %17 offset:256\n\t"
                     "ds_read_b128 %4, %16 offset:512\n\tds_read_b128 %5, %17 offset:512\n\t"
                     "ds_read_b128 %6, %16 offset:768\n\tds_read_b128 %7, %17 offset:768\n\t"
                     "ds_read_b128 %8, %16 offset:1024\n\tds_read_b128 %9, %17 offset:1024\n\t"
                     "ds_read_b128 %10, %16 offset:1280\n\tds_read_b128 %11, %17 offset:1280\n\t"
                     "ds_read_b128 %12, %16 offset:1536\n\tds_read_b128 %13, %17 offset:1536\n\t"
                     "ds_read_b128 %14, %16 offset:1792\n\tds_read_b128 %15, %17 offset:1792\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                       "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
                     : "v"(a_even), "v"(a_odd) : "memory");
    }
}
template <int ND>
__device__ __forceinline__ void df_read_xrow(double (&x)[ND], unsigned a) {
    double lo[8];
    lds_read_row1<8>(lo, a);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = lo[i];
    if constexpr (ND == 16) {
        double hi[8];
        lds_read_row1<8>(hi, a + 64u);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[8 + i] = hi[i];
    }
}

// the static kernel at one node, the arithmetic of sk_static.hip:static_node (dot product chain over the staged dimensions -- the
// zero padding adds exact zeros --, rbf: dist = -2 xy + (xs + ys), G = exp(-dist / sigma) with the library exp)
template <int FD, int KIND>
__device__ __forceinline__ double df_node(const double (&xv)[FD], double xs, const d2_t (&yv)[FD], int q, double ys, double inv_sigma) {
    double xy = 0.0;
#pragma unroll
    for (int k = 0; k < FD; ++k) xy = fma(xv[k], yv[k][q], xy);
    if constexpr (KIND == 0) return xy;
    const double e = -(fma(-2.0, xy, xs + ys)) * inv_sigma;
    return exp(e);
}

// LDSB (two or more bands of a SHORT second path, 64 <= NUp < 80 units): the band boundary lives in LDS -- a ring of 32 entries, lane
// 63 writes its entry, lane 0 reads it NUp - 63 = 1..16 macro-steps later in program order -- instead of travelling through the
// per-wave row in L2, whose flush-and-refetch needs 17 macro-steps of slack.
// SHIFT (whenever it costs no extra band: M - 1 not a multiple of 64): the lanes own the coarse rows one LOWER -- lane (band, lam) owns
// coarse row 64 band + lam - 1 and evaluates node row 64 band + lam -- so node row 0 is lane 0's regular row in band 0 and its coarse
// row -1 is padding (zero increments: the states stay at their boundary values).  Without it lane 0 evaluates node row 0 itself while
// it is in band 0, a branch the whole wave pays for in every macro-step of band 0: 60-90 instructions, a fifth of the step with one band.
template <int DY, int KIND, int FD, bool LDSB, bool SHIFT>
__global__ __launch_bounds__(4 * WAVE) void k_deriv_fused(const DerivFusedParams prm) {
    constexpr int CW = 2;
    constexpr int R = 1 << DY, S = CW << DY, r = 1 << DY;
    constexpr int L = DF_L;
    constexpr int XROW = FD * 8, PPR = FD / 2;
    constexpr int XSLAB = 8 * 3 * XROW;                // [lane of the window][shifted path][FD]
    constexpr int YSLAB = FD * 128, NSLAB = L / 8 + 2, NDMA_Y = YSLAB / 1024, NDMA_X = (XSLAB + 1023) / 1024;
    // band boundary entry of one unit: the bottom fine row of the three states (3 S) and the 2 x 3 node values under it
    constexpr int E = 3 * S + 6, NP = E / 2, CHUNK = 8 * E * 8, CPIECES = CHUNK / 16;
    extern __shared__ __attribute__((aligned(16))) char lds_block[];
    char *lds;
    const int64_t wave_id = wave_slot(prm.wg, lds_block, lds);
    if (wave_id < 0) return;
    const unsigned lds0 = lds_offset(lds);
    // LDS map of a wave: [y ring][x ring: 2 slabs][boundary chunks in: 2 slots][boundary chunk out][node row 0 of the pair: 2 x 3 rows]
    constexpr unsigned X_BASE = NSLAB * YSLAB, BI_BASE = X_BASE + DF_X_SLOTS * XSLAB, BO_BASE = BI_BASE + (LDSB ? 0 : 2 * CHUNK),
                       T_BASE = BO_BASE + (LDSB ? 0 : CHUNK), CE_BASE = T_BASE + 2 * 3 * XROW, LB_BASE = CE_BASE + E * 8;
    // (LDSB: [constant entry][boundary ring: 32 entries -- an entry is read NUp - 63 <= 16 macro-steps after it was written] behind the
    // node rows instead of the three chunks)
    constexpr int LBR = 32;

    const int lam = threadIdx.x & (WAVE - 1);
    const int NUp = prm.NUp, nb = prm.nb;
    const double sc = 1.0 / (double)(1 << (2 * DY));
    const bool is_top = lam == 0, is_bot = lam == L - 1;

    // ---- cursors: (u, band, ps) = this lane's node evaluation / path reads; (uk, bandk, psk) = its block sweep, one unit behind
    int u, band, ps, uk, bandk, psk;
    {
        int sig = floor_div(-lam, NUp);
        u = -lam - sig * NUp;
        ps = floor_div(sig, nb);
        band = sig - ps * nb;
        sig = floor_div(-lam - 1, NUp);
        uk = -lam - 1 - sig * NUp;
        psk = floor_div(sig, nb);
        bandk = sig - psk * nb;
    }
    int yslab, ypar;
    {
        const int s0 = floor_div(-lam, 8);
        yslab = ((s0 % NSLAB) + NSLAB) % NSLAB;
        ypar = s0 & 1;
    }
    const int lam7 = lam & 7;
    // ---- the wave's stream of pairs (sk_wave_fused_mb.hip): C0 fixed, then one pair per draw from the launch's counter
    constexpr unsigned NOPAIR = 0xffffffffu;
    const unsigned P32 = (unsigned)prm.P;
    const int C0 = prm.C0;
    const unsigned base0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(wave_id * C0));
    unsigned cb1 = NOPAIR, cb2 = NOPAIR, cb3 = NOPAIR, cb0 = NOPAIR;
    int have = 0;
    int t_end = 0x7fffffff;
    const int tail = (DF_L - 1) + 1;
    auto stream_pair = [&](int i) __attribute__((always_inline)) -> unsigned {
        if (i < 0) return NOPAIR;
        if (i < C0) { const unsigned p = base0 + (unsigned)i; return p < P32 ? p : NOPAIR; }
        const int kk = (i - C0) & 3;
        return (cb0 & -(unsigned)(kk == 0)) | (cb1 & -(unsigned)(kk == 1)) | (cb2 & -(unsigned)(kk == 2)) | (cb3 & -(unsigned)(kk == 3));
    };
    auto ensure = [&](int f) __attribute__((always_inline)) {
        while (C0 + have <= f) {
            unsigned b = NOPAIR;
            if (prm.queue && t_end == 0x7fffffff) {
                unsigned long long v = 0;
                if (lam == 0) v = atomicAdd(prm.queue, 1ULL);
                const unsigned long long q = (unsigned long long)prm.q_first +
                                             (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                              (unsigned)__builtin_amdgcn_readfirstlane((int)v));
                b = q < (unsigned long long)P32 ? (unsigned)q : NOPAIR;
            }
            if (b == NOPAIR && t_end == 0x7fffffff) t_end = (C0 + have) * prm.nb * prm.NUp + tail;
            const int kk = have & 3;
            const unsigned m0 = -(unsigned)(kk == 0), m1 = -(unsigned)(kk == 1), m2 = -(unsigned)(kk == 2), m3 = -(unsigned)(kk == 3);
            cb0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb0 & ~m0) | (b & m0)));
            cb1 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb1 & ~m1) | (b & m1)));
            cb2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb2 & ~m2) | (b & m2)));
            cb3 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb3 & ~m3) | (b & m3)));
            have += 1;
        }
    };
    const int my_uf = lam == prm.lam_f ? prm.u_f : -1;
    const unsigned my_x = lds0 + X_BASE + (unsigned)(lam7 * 3 * XROW);

    auto split_b = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p % (uint32_t)prm.B);   // (32-bit: the launcher refuses P >= 2^31 - 2^20, and B <= P)
    };
    auto split_a = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p / (uint32_t)prm.B);
    };
    double *const wsrow = prm.ws + wave_id * prm.ws_stride;

    // ---- producers (wave-uniform control), once per window of 8 macro-steps ------------------------------------------------
    int y_pi = 0, y_band = 0, y_u0 = 0, y_slot = 0, y_par = 0;
    auto issue_y = [&]() {
        ensure(y_pi);
        const unsigned spy = stream_pair(y_pi);
        const int64_t b = split_b(spy == NOPAIR ? 0 : (int64_t)spy);
#pragma unroll
        for (int c = 0; c < NDMA_Y; ++c) {
            const int krow = (c * 8 + (lam >> 3)) ^ (y_par & 1);     // odd slabs: dimension rows swapped in pairs
            const double *src = prm.Yt + ((b * FD + krow) * (int64_t)prm.Ncp + (int64_t)(y_u0 + (lam & 7)) * 2);
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(lds + y_slot * YSLAB + c * 1024), 16, 0, 0);
        }
        y_slot = y_slot + 1 == NSLAB ? 0 : y_slot + 1;
        y_par ^= 1;
        y_u0 += 8;
        if (y_u0 == NUp) {
            y_u0 = 0;
            y_band += 1;
            if (y_band == nb) { y_band = 0; y_pi += 1; }
        }
    };
    int x_pi = 0, x_band = 0, x_lam0 = 0, x_slot = 0;
    auto issue_x = [&]() {
        ensure(x_pi);
        const unsigned spx = stream_pair(x_pi);
        const int64_t a = split_a(spx == NOPAIR ? 0 : (int64_t)spx);
        const int lamj = x_lam0 < L ? x_lam0 : 0;     // nobody starts: fetch something valid
        char *dst = lds + X_BASE + x_slot * XSLAB;
        // piece idx = (lane i of the window, shifted path v, 16-byte piece): node row (x_band L + lamj + i) + 1 of path v
#pragma unroll
        for (int c = 0; c < NDMA_X; ++c) {
            const int idx = c * 64 + lam;
            if (XSLAB % 1024 == 0 || idx < XSLAB / 16) {
                const int i = idx / (3 * PPR), rem = idx - i * (3 * PPR), v = rem / PPR, piece = rem - v * PPR;
                const double *xb = v == 0 ? prm.Xr[0] : v == 1 ? prm.Xr[1] : prm.Xr[2];
                const double *src = xb + (a * prm.Mrows + (int64_t)(x_band * L + lamj + i) + (SHIFT ? 0 : 1)) * FD + piece * 2;
                __builtin_amdgcn_global_load_lds(src, (lds_void *)(dst + c * 1024), 16, 0, 0);
            }
        }
        if (!SHIFT && lam < 3 * PPR) {   // node row 0 of the pair, the three shifted paths (lane 0 evaluates it itself in band 0)
            const int v = lam / PPR, piece = lam - v * PPR;
            const double *xb = v == 0 ? prm.Xr[0] : v == 1 ? prm.Xr[1] : prm.Xr[2];
            __builtin_amdgcn_global_load_lds(xb + a * prm.Mrows * FD + piece * 2, (lds_void *)(lds + T_BASE + (x_pi & 1) * 3 * XROW), 16, 0, 0);
        }
        // lane 0's boundary entries, past the L1.  The state part of an entry belongs to the sweep one unit behind: the windows
        // whose sweep is in band 0 take it from the constant chunk (1, 0, 0); with the lag entry 0 of a band's first window still
        // belongs to the PREVIOUS band's sweep (sk_wave_fused_mb.hip)
#pragma unroll
        for (int c = 0; c < (LDSB ? 0 : (CPIECES + 63) / 64); ++c) {
            const int idx = c * 64 + lam;
            if (idx < CPIECES) {
                const bool ones_rest = x_band == 0;
                const bool ones_first = x_lam0 > 0 ? ones_rest : x_band == (nb > 1 ? 1 : 0);
                const int piece = idx * 2;
                const bool ones = piece < 3 * S ? ones_first : ones_rest;
                const double *sb = (ones ? wsrow + (int64_t)NUp * E : wsrow + (int64_t)x_lam0 * E) + piece;
                __builtin_amdgcn_global_load_lds(sb, (lds_void *)(lds + BI_BASE + x_slot * CHUNK + c * 1024), 16, 0, 17);
            }
        }
        x_slot ^= 1;
        x_lam0 += 8;
        if (x_lam0 == NUp) {
            x_lam0 = 0;
            x_band += 1;
            if (x_band == nb) { x_band = 0; x_pi += 1; }
        }
    };
    int f_pos = 0;
    auto flush_chunk = [&]() {
#pragma unroll
        for (int c = 0; c < (CPIECES + 63) / 64; ++c) {
            const int idx = c * 64 + lam;
            if (idx < CPIECES) {
                d2_t v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds0 + BO_BASE + (unsigned)(idx * 16)) : "memory");
                df_store_through(wsrow + (int64_t)f_pos * E + idx * 2, v);
            }
        }
        f_pos += 8;
        if (f_pos == NUp) f_pos = 0;
    };

    // ---- state ------------------------------------------------------------------------------------------------------------
    double xr[3][FD], xsq[3];   // the lane's bottom node row of the three shifted paths, |x|^2 of each
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        xsq[v] = 0.0;
#pragma unroll
        for (int j = 0; j < FD; ++j) xr[v][j] = 0.0;
    }
    // node values of this lane's bottom row at the columns of units uk (0..1) and uk+1 (2..3), per shifted path; the row above
    double own[4][3], abv[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int v = 0; v < 3; ++v) { own[c][v] = 0.0; abv[c][v] = 0.0; }
    // state s: 0 = K, 1 = K_gamma, 2 = K_gamma_gamma; boundary values 1, 0, 0
    double left[3][R], bot[3][S], corner[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double bv = s == 0 ? 1.0 : 0.0;
        corner[s] = bv;
#pragma unroll
        for (int i = 0; i < R; ++i) left[s][i] = bv;
#pragma unroll
        for (int i = 0; i < S; ++i) bot[s][i] = bv;
    }
    const double c1 = prm.c1, c2 = prm.c2, c3 = prm.c3;

    {   // zero the slice (lanes ahead of their first band read slabs no DMA has written yet)
        const d2_t z = {0.0, 0.0};
        const unsigned lds_end = LDSB ? LB_BASE + (unsigned)(LBR * E * 8) : T_BASE + 2 * 3 * XROW;
        for (unsigned o = (unsigned)lam * 16u; o < lds_end; o += WAVE * 16) lds_write_b128(lds0 + o, z);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the constant chunk: per entry the boundary values of the three states (1, 0, 0) and zero node values (never used)
    if constexpr (LDSB) {
        if (lam < NP) {
            const double v = lam * 2 < S ? 1.0 : 0.0;
            lds_write_b128(lds0 + CE_BASE + (unsigned)(lam * 16), d2_t{v, v});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
#pragma unroll
        for (int c = 0; c < (CPIECES + 63) / 64; ++c) {
            const int idx = c * 64 + lam;
            if (idx < CPIECES) {
                const double v = (idx * 2) % E < S ? 1.0 : 0.0;
                df_store_through(wsrow + (int64_t)NUp * E + idx * 2, d2_t{v, v});
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    issue_y();
    issue_x();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue_y();
    issue_x();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int t = 0; t < t_end; ++t) {
        // -- lane 0's boundary entry of its unit u (a uniform address, broadcast read; no wait: complete at the y read below)
        // Variants beyond 256 VGPRs (16 staged dimensions: one wave per SIMD, registers spill to AGPRs) read the entry with blocking
        // reads; the others leave it in flight until the y read's wait
        constexpr bool PEND = FD < 16 && DY < 2;   // (dyadic 2: 280-300 registers)
        d2_t pend[NP], bnd[NP];
        if constexpr (!PEND) {
            double ent[E];
            if constexpr (LDSB) {
                const int bk0 = __builtin_amdgcn_readfirstlane(bandk);
                const unsigned ra = lds0 + LB_BASE + (unsigned)(((t - NUp) & (LBR - 1)) * (E * 8));
                lds_read_block<3 * S>(ent, bk0 == 0 ? lds0 + CE_BASE : ra);
                lds_read_block<6>(ent + 3 * S, ra + 3 * S * 8u);
            } else {
                lds_read_block<E>(ent, lds0 + BI_BASE + (unsigned)(((t >> 3) & 1) * CHUNK + (t & 7) * (E * 8)));
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) bnd[i] = d2_t{ent[2 * i], ent[2 * i + 1]};
        } else if constexpr (LDSB) {
            // lane 0's cursors are wave-uniform through readfirstlane: the state part from the constant entry while its sweep is in
            // band 0, else from the row; the node part from the row (unused in band 0: lane 0 evaluates node row 0 itself)
            const int bk0 = __builtin_amdgcn_readfirstlane(bandk);
            // (ring slot = the WRITER's virtual unit modulo 32: lane 63 wrote this entry NUp macro-steps before lane 0's step minus its
            //  63 of skew, i.e. at virtual unit t - NUp)
            const unsigned ra = lds0 + LB_BASE + (unsigned)(((t - NUp) & (LBR - 1)) * (E * 8));
#pragma unroll
            for (int i = 0; i < NP; ++i) df_begin(pend[i]);
            df_read_pieces<0, 3 * S / 2>(pend, bk0 == 0 ? lds0 + CE_BASE : ra);
            df_read_pieces<3 * S / 2, 3>(pend, ra + 3 * S * 8u);
        } else {
            df_read_entry<NP>(pend, lds0 + BI_BASE + (unsigned)(((t >> 3) & 1) * CHUNK + (t & 7) * (E * 8)));
        }

        if (uk == 0) {
            asm volatile("");
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const double bv = s == 0 ? 1.0 : 0.0;
                corner[s] = bv;
#pragma unroll
                for (int i = 0; i < R; ++i) left[s][i] = bv;
            }
        }
        if (u == 0) {
            asm volatile("");
            const unsigned xa = my_x + (unsigned)(((t >> 3) & 1) * XSLAB);
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                df_read_xrow<FD>(xr[v], xa + v * XROW);
                double q2 = 0.0;
#pragma unroll
                for (int j = 0; j < FD; ++j) q2 = fma(xr[v][j], xr[v][j], q2);
                xsq[v] = q2;
            }
        }

        // -- y points of the two node columns of unit u, their squared norms
        d2_t yv[FD];
        {
            const unsigned ya = lds0 + (unsigned)(yslab * YSLAB + ((u & 7) << 4));
            df_read_ydims<FD>(yv, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
        }
        if constexpr (PEND) {
#pragma unroll
            for (int i = 0; i < NP; ++i) df_take(bnd[i], pend[i]);
        }
        double ysq[CW];
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            double s2 = 0.0;
#pragma unroll
            for (int j = 0; j < FD; ++j) s2 = fma(yv[j][q], yv[j][q], s2);
            ysq[q] = s2;
        }

        // -- top row of the block for the three states: the lane above's bottom row; lane 0: the boundary entry
        double top[3][S];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < S; ++i) top[s][i] = dpp_shr1(bot[s][i], bnd[(s * S + i) >> 1][(s * S + i) & 1]);
        // -- the row above at the columns of unit u = uk + 1: the lane above evaluated them one macro-step ago
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int q = 0; q < CW; ++q) abv[2 + q][v] = dpp_shr1(own[q][v], bnd[(3 * S + 2 * v + q) >> 1][(3 * S + 2 * v + q) & 1]);
        if (!SHIFT && is_top && band == 0) {   // node row 0 of the pair: nobody above has it
            asm volatile("");
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                double x0[FD];
                df_read_xrow<FD>(x0, lds0 + T_BASE + (unsigned)((ps & 1) * 3 * XROW + v * XROW));
                double q2 = 0.0;
#pragma unroll
                for (int j = 0; j < FD; ++j) q2 = fma(x0[j], x0[j], q2);
#pragma unroll
                for (int q = 0; q < CW; ++q) abv[2 + q][v] = df_node<FD, KIND>(x0, q2, yv, q, ysq[q], prm.inv_sigma);
            }
        }
        // -- this lane's bottom node row at the two columns of unit u
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int q = 0; q < CW; ++q) own[2 + q][v] = df_node<FD, KIND>(xr[v], xsq[v], yv, q, ysq[q], prm.inv_sigma);

        // -- the three increments of the two coarse cells of unit uk (reference order, no contraction: see the header)
        double ginc[3][CW];
        {
#pragma clang fp contract(off)
            double sv_a[3][6], sv_o[3][6];   // scaled values at columns 0..2 of the row above / the own row
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                {
                    const double g0 = abv[c][0], g1 = abv[c][1], g2 = abv[c][2];
                    sv_a[c][0] = g0;
                    sv_a[c][1] = -c1 * g0;
                    sv_a[c][2] = c1 * g1;
                    sv_a[c][3] = -c1 * sv_a[c][1];
                    sv_a[c][4] = -c2 * sv_a[c][2];
                    sv_a[c][5] = c3 * g2;
                }
                {
                    const double g0 = own[c][0], g1 = own[c][1], g2 = own[c][2];
                    sv_o[c][0] = g0;
                    sv_o[c][1] = -c1 * g0;
                    sv_o[c][2] = c1 * g1;
                    sv_o[c][3] = -c1 * sv_o[c][1];
                    sv_o[c][4] = -c2 * sv_o[c][2];
                    sv_o[c][5] = c3 * g2;
                }
            }
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                double d[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) d[k] = ((sv_o[q + 1][k] + sv_a[q][k]) - sv_o[q][k]) - sv_a[q + 1][k];   // ((G11 + G00) - G10) - G01
                ginc[0][q] = d[0];
                ginc[1][q] = d[1] + d[2];
                ginc[2][q] = (d[3] + d[4]) + d[5];
            }
        }

        // -- coefficients per coarse cell (sk_wave_deriv.hip)
        double ca[CW], cb[CW], c_t[CW], c_s[CW], c_k[CW], c_m[CW], c_tdd[CW], c_td[CW], c_kdd[CW], c_kd[CW];
        const double sc_l = (SHIFT && is_top && bandk == 0) ? 0.0 : sc;   // SHIFT: lane 0's coarse row in band 0 is padding
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            const double g = ginc[0][q] * sc_l, gd = ginc[1][q] * sc_l, gdd = ginc[2][q] * sc_l;
            const double g2 = g * g, qg = 0.25 * g;
            ca[q] = fma(g2, 1.0 / 12.0, fma(g, 0.5, 1.0));
            cb[q] = fma(g2, -1.0 / 12.0, 1.0);
            c_t[q] = 0.25 * gd;
            c_s[q] = fma(g, 0.5, 1.0);
            c_k[q] = qg * gd;
            c_m[q] = fma(qg, g, -1.0);
            c_tdd[q] = 0.25 * gdd;
            c_td[q] = 0.5 * gd;
            c_kdd[q] = qg * gdd;
            c_kd[q] = (qg + qg) * gd;
        }

        // -- sweep the R x S block, three states
        double cand[3][CW];
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
            const int q = cc >> DY;
            double above[3], diag[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                above[s] = top[s][cc];
                diag[s] = cc == 0 ? corner[s] : top[s][cc - 1];
            }
            double pk = diag[0] + above[0], pd = diag[1] + above[1];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const double k10 = left[0][rr], k10d = left[1][rr], k10dd = left[2][rr];
                const double k01 = above[0], k01d = above[1], k01dd = above[2];
                const double k00 = diag[0], k00d = diag[1], k00dd = diag[2];
                const double k11 = fma(k01, ca[q], fma(k10, ca[q], -(k00 * cb[q])));
                const double nk = k10 + k11;
                const double tt = pk + nk;
                const double s1 = k01d + k10d;
                const double k11d = fma(c_t[q], tt, fma(c_s[q], s1, fma(c_k[q], k00, c_m[q] * k00d)));
                const double nd = k10d + k11d;
                const double td = pd + nd;
                const double s1dd = k01dd + k10dd;
                const double k11dd = fma(c_tdd[q], tt, fma(c_td[q], td, fma(c_s[q], s1dd,
                                     fma(c_kdd[q], k00, fma(c_kd[q], k00d, c_m[q] * k00dd)))));
                pk = nk; pd = nd;
                diag[0] = k10; diag[1] = k10d; diag[2] = k10dd;
                above[0] = k11; above[1] = k11d; above[2] = k11dd;
                left[0][rr] = k11; left[1][rr] = k11d; left[2][rr] = k11dd;
                if (rr == R - 1 && (cc & (r - 1)) == r - 1) {
                    cand[0][q] = k11; cand[1][q] = k11d; cand[2][q] = k11dd;
                }
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) bot[s][cc] = above[s];
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) corner[s] = top[s][S - 1];

        // -- lane 63: this step's boundary entry (position = its unit u) into the outgoing chunk
        if (is_bot) {
            const unsigned ea = LDSB ? lds0 + LB_BASE + (unsigned)(((t - (L - 1)) & (LBR - 1)) * (E * 8)) : lds0 + BO_BASE + (unsigned)((u & 7) * (E * 8));
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int cc = 0; cc < S; cc += 2) lds_write_b128(ea + (unsigned)(s * S + cc) * 8u, d2_t{bot[s][cc], bot[s][cc + 1]});
#pragma unroll
            for (int v = 0; v < 3; ++v) lds_write_b128(ea + (unsigned)(3 * S + 2 * v) * 8u, d2_t{own[2][v], own[3][v]});
        }

        // -- the three values of a pair
        if (uk == my_uf) {
            int pv = psk, bv = bandk;
            asm volatile("" : "+v"(pv), "+v"(bv));
            const unsigned pair_v = bv == prm.band_f ? stream_pair(pv) : NOPAIR;
            if (pair_v != NOPAIR) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    double v = cand[s][0];
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        double cv = cand[s][q];
                        asm volatile("" : "+v"(cv));
                        if (q == prm.sel_f) v = cv;
                    }
                    prm.out[s][pair_v] = v;
                }
            }
        }

        // -- shift the node history
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            own[0][v] = own[2][v]; own[1][v] = own[3][v];
            abv[0][v] = abv[2][v]; abv[1][v] = abv[3][v];
        }

        // -- advance the cursors
        uk += 1;
        if (uk == NUp) {
            uk = 0;
            bandk += 1;
            if (bandk == nb) { bandk = 0; psk += 1; }
        }
        u += 1;
        if (((t + 1) & 7) == lam7) {   // (u & 7) == 0
            yslab = yslab + 1 == NSLAB ? 0 : yslab + 1;
            ypar ^= 1;
            if (u == NUp) {
                u = 0;
                band += 1;
                if (band == nb) { band = 0; ps += 1; }
            }
        }
        if (!LDSB && ((t + 1) & 7) == 7 && t >= L - 1 + 7) flush_chunk();
        if (((t + 1) & 7) == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            issue_y();
            issue_x();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct DfPlan {
    int S, NUp, nb, fd, E;
    bool ldsb, shift;
    size_t lds_bytes;
    int64_t ws_stride;
    bool ok;
};

DfPlan df_plan(int Mc, int Nc, int dyadic, int D) {
    DfPlan pl{};
    pl.ok = false;
    // (path dims beyond 8 stay on the unfused route: the 16-dim variants need 300-380 registers -- one wave per SIMD, little gained --
    // and one of them, linear at dyadic 1, gave last-bit run-to-run differences whose cause was not found)
    if (dyadic < 0 || dyadic > 2 || D < 1 || D > 8) return pl;
    pl.S = 2 << dyadic;
    pl.fd = 8;
    const int NU = (Nc + 2) / 2;
    pl.NUp = (NU + LINE_UNITS - 1) / LINE_UNITS * LINE_UNITS;
    pl.nb = (Mc + DF_L - 1) / DF_L;                         // one coarse row per lane
    pl.shift = Mc % DF_L != 0 && !knobs().derivf_noshift;                              // the lanes one row lower fit the same bands: node row 0 becomes a regular row
    // several bands: the boundary through L2 needs NUp >= 80 (flush + refetch slack, sk_wave_fused_mb.hip); shorter rows keep it in
    // LDS (NUp >= 64: lane 63 must have written an entry before lane 0 reads it); one band: no boundary at all
    if (pl.NUp < DF_L) return pl;        // (the stream logic -- one band start per window, a ring of four pairs -- needs rows of >= 64 units)
    pl.ldsb = pl.nb > 1 && pl.NUp < DF_L + 16;
    pl.E = 3 * pl.S + 6;
    const size_t xslab = (size_t)8 * 3 * pl.fd * 8, chunk = (size_t)8 * pl.E * 8;
    pl.lds_bytes = (size_t)(DF_L / 8 + 2) * pl.fd * 128 + DF_X_SLOTS * xslab + (size_t)2 * 3 * pl.fd * 8 +
                   (pl.ldsb ? (size_t)(32 + 1) * pl.E * 8 : 3 * chunk);
    pl.ws_stride = (int64_t)(pl.NUp + 8) * pl.E;
    pl.ok = true;
    return pl;
}

template <int DY, int KIND, int FD, bool LDSB, bool SHIFT>
int launch_df(DerivFusedParams prm, const DfPlan &pl, void *ws, size_t ws_bytes, hipStream_t s) {
    auto kern = k_deriv_fused<DY, KIND, FD, LDSB, SHIFT>;
    static const int vgprs = [&] {
        hipFuncAttributes attr;
        return hipFuncGetAttributes(&attr, (const void *)kern) == hipSuccess && attr.numRegs > 0 ? attr.numRegs : 256;
    }();
    const int wpb0 = wave_group(pl.lds_bytes, 1 << 20, knobs().derivf_wpb).wpb;
    int wpc = (int)((160 * 1024) / (pl.lds_bytes * wpb0)) * wpb0;
    const int by_regs = 4 * (512 / ((vgprs + 7) & ~7));
    if (wpc > by_regs) wpc = by_regs;
    if (knobs().derivf_wpc > 0 && wpc > knobs().derivf_wpc) wpc = knobs().derivf_wpc;
    if (wpc > 8) wpc = 8;
    if (wpc >= wpb0) wpc = wpc / wpb0 * wpb0;
    if (wpc < 1) wpc = 1;
    const int64_t P = prm.P;
    const int64_t max_waves = (int64_t)device_cu_count() * wpc;
    int64_t waves = P < max_waves ? P : max_waves;
    if (P >= 0x7ff00000LL) return SK_ERR_UNSUPPORTED;
    const int64_t per = (P + waves - 1) / waves;
    if (per > 0x1fffffff / ((int64_t)prm.nb * prm.NUp)) return SK_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < (size_t)waves * (size_t)pl.ws_stride * sizeof(double) + 64) return SK_ERR_WORKSPACE;
    double *w = static_cast<double *>(ws);
    if (waves == max_waves && per >= 8) {   // the launch fills the chip: half the equal share up front, the rest drawn pair by pair
        prm.C0 = (int)(per / 2);
        prm.queue = reinterpret_cast<unsigned long long *>(w + (size_t)waves * (size_t)pl.ws_stride);
        prm.q_first = waves * (int64_t)prm.C0;
        if (hipMemsetAsync(prm.queue, 0, sizeof(unsigned long long), s) != hipSuccess) return SK_ERR_LAUNCH;
    } else {
        waves = (P + per - 1) / per;
        prm.C0 = (int)per;
        prm.queue = nullptr;
        prm.q_first = P;
    }
    prm.ws = w;
    prm.ws_stride = pl.ws_stride;
    prm.wg = wave_group(pl.lds_bytes, waves, knobs().derivf_wpb);
    const size_t lds_block = wave_group_lds(prm.wg);
    if (lds_block > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
    SK_LAUNCH(kern, dim3(wave_group_blocks(prm.wg)), dim3(WAVE * prm.wg.wpb), lds_block, s, prm);
    return check_launch();
}

template <int DY, int KIND, bool SHIFT>
int launch_df_s(const DerivFusedParams &prm, const DfPlan &pl, void *ws, size_t ws_bytes, hipStream_t s) {
    if (pl.fd != 8) return SK_ERR_UNSUPPORTED;
    return pl.ldsb ? launch_df<DY, KIND, 8, true, SHIFT>(prm, pl, ws, ws_bytes, s) : launch_df<DY, KIND, 8, false, SHIFT>(prm, pl, ws, ws_bytes, s);
}
template <int DY, int KIND>
int launch_df_k(const DerivFusedParams &prm, const DfPlan &pl, void *ws, size_t ws_bytes, hipStream_t s) {
    return pl.shift ? launch_df_s<DY, KIND, true>(prm, pl, ws, ws_bytes, s) : launch_df_s<DY, KIND, false>(prm, pl, ws, ws_bytes, s);
}
template <int DY>
int launch_df_dy(const DerivFusedParams &prm, const DfPlan &pl, int kind, void *ws, size_t ws_bytes, hipStream_t s) {
    return kind == 0 ? launch_df_k<DY, 0>(prm, pl, ws, ws_bytes, s) : launch_df_k<DY, 1>(prm, pl, ws, ws_bytes, s);
}

}  // namespace

// workspace bytes of sk_solve_deriv_static_f64 (0: outside the kernel's scope) and the rows of the three staged x arrays
size_t deriv_fused_workspace_bytes(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows) {
    const DfPlan pl = df_plan(Mc, Nc, dyadic, D);
    if (!pl.ok || P <= 0) return 0;
    if (mrows) *mrows = pl.nb * DF_L + 8 + 1;
    const int64_t max_waves = (int64_t)device_cu_count() * 8;
    return (size_t)(P < max_waves ? P : max_waves) * (size_t)pl.ws_stride * sizeof(double) + 64;
}

int launch_deriv_fused(int kind, const double *X0r, const double *X1r, const double *X2r, const double *Yt, int64_t A, int64_t B, int Mrows,
                       int Ncp, int D, int fd, const Geom &g, double inv_sigma, double eps, double *out_k, double *out_kd, double *out_kdd,
                       void *ws, size_t ws_bytes, hipStream_t s) {
    if (g.naive || B < 0 || g.P != (B > 0 ? A * B : A) || (kind != 0 && kind != 1)) return SK_ERR_UNSUPPORTED;
    const DfPlan pl = df_plan(g.Mc, g.Nc, g.dyadic, D);
    if (!pl.ok || fd != pl.fd) return SK_ERR_UNSUPPORTED;
    if (Ncp < pl.NUp * 2 || (Ncp & 1) || Mrows < pl.nb * DF_L + 8 + 1) return SK_ERR_UNSUPPORTED;
    DerivFusedParams prm{};
    prm.Xr[0] = X0r; prm.Xr[1] = X1r; prm.Xr[2] = X2r; prm.Yt = Yt;
    prm.out[0] = out_k; prm.out[1] = out_kd; prm.out[2] = out_kdd;
    prm.P = g.P; prm.B = B; prm.Mrows = Mrows; prm.Ncp = Ncp; prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = pl.NUp; prm.nb = pl.nb;
    prm.inv_sigma = inv_sigma;
    prm.c1 = 1. / eps; prm.c2 = 2. / eps; prm.c3 = 1. / (eps * eps);
    const int row_unit = g.Mc - 1 + (pl.shift ? 1 : 0);      // the lane-row that owns the last coarse row
    prm.u_f = (g.Nc - 1) / 2;
    prm.lam_f = row_unit % DF_L;
    prm.band_f = row_unit / DF_L;
    prm.sel_f = (g.Nc - 1) % 2;
    switch (g.dyadic) {
        case 0: return launch_df_dy<0>(prm, pl, kind, ws, ws_bytes, s);
        case 1: return launch_df_dy<1>(prm, pl, kind, ws, ws_bytes, s);
        default: return launch_df_dy<2>(prm, pl, kind, ws, ws_bytes, s);
    }
}

}  // namespace sk
