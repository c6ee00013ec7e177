// sk_wave_adj_fused_mb.hip -- the fused RBF adjoint (sk_wave_adj_fused_rbf.hip) for pairs that need SEVERAL BANDS of a wavefront
// (long paths: M > 64 RC) and for path dimensions up to 16: adjoint PDE, node evaluation and the static kernel's chain rule in
// one kernel, from the paths and the terminal edges that sk_solve_fwd_static_* kept.  Neither the increments nor W exist in HBM.
//
// The sweep is the one of sk_wave_adj_fused_rbf.hip (flipped coordinates, two states per node -- the reverse PDE and the forward
// solution recomputed backwards from its terminal row and column --, RC x 2 exponentials per macro-step, V completed from the
// lane above / the previous step, 1 + D accumulators per node row); see there for the mathematics.  The stream is the one of
// sk_wave_fused_mb.hip: one 64-lane wavefront sweeps one pair at a time, band after band (band 0 = the LAST rows of the pair:
// the sweep is flipped), lane `lam` runs `lam` macro-steps behind lane 0.  What is new:
//
//   * Band boundary.  What lane 0 takes from "the lane above" -- the bottom fine row of the reverse state, the last node row's
//     two values and the last coarse row's two weights -- is what lane 63 had for the same unit one band earlier: S + 4 doubles
//     per unit, staged in LDS, written through to the wave's row in global memory (L2) every 8 macro-steps and brought back by
//     LDS-DMA one window ahead (sk_wave_fused_mb.hip); in band 0 the entries come from a constant chunk (ones, zero weights).
//     The FORWARD state does not cross bands: the forward pass kept the bottom row of every band (the terminal row being the
//     last), lane 0 takes its top row from there (chunks as in sk_wave_adj.hip), so the backward recompute of K restarts from
//     exact values every 64 R fine rows and its error is that of a single band whatever the path length.
//   * Accumulators per (pair, band).  A lane's node rows change with the band, so its 1 + D sums per node row are written out
//     when it enters the next band -- to Gpart[pair][node row][2 + FD] -- one macro-step late: node column 0 of a band is
//     completed during the first macro-step of the next one (sk_wave_adj_fused_rbf.hip), and that part still belongs to the
//     rows being left.  The caller adds the pairs of an x_a.
//
// Scope: fp64 sweep, dyadic 0..2 (at 0 two coarse rows per lane, bands of 128 rows: k_fwd_fused_mb keeps its edges in that layout),
// path dim <= 16, either stencil, any M and N (second paths shorter than ~160 points are swept with masked padding units: NUp >= 80).
// Replaces, for RBFKernel on long or wide paths, sk_static_increments + sk_solve_fwd(EDGES) + sk_solve_adj + sk_static_adjoint,
// i.e. sigkernel.py:419-502 (prep_backward) + :404-416.
#include "sk_wave_common.h"

namespace sk {
namespace {

constexpr int AMB_L = WAVE;
constexpr int AMB_X_SLOTS = 2;

struct AdjMbParams {
    const double *Xr;      // [A][Mrows][FD]  x_p (points), zero rows / dims beyond M / D
    const double *Yt;      // [B][FD][Ncp]    y_q, dimension-major, zero-padded;  Y32: [B][FD/2 + 1][Ncp] fp32 points packed two
                           // dimensions per 16-byte unit + a row of |y_q|^2 in fp64 (sk_prep_paths_f32 layout 2, as sk_wave_fused_mb.hip)
    const double *edges;   // [P][nb NNp + MMp]  K[MMp - b 64 R][1..NNp] (b < nb), K[1..MMp][NNp] of the padded grid (sk_solve_fwd_static_* with edges)
    const double *scale;   // [P] upstream gradient per pair, nullable
    double *Gpart;         // [P][Mcp + 1][FD + 2] per node row >= 1: cs, 0, accd[0..FD)  (row 0 is not written: N0)
    double *N0;            // [P][2 NUp] node row 0: per node column c the weight V[0][c] G[0][c] s_ab; the caller contracts it with y_b
    double *err;           // [P] zero-initialised: worst |Kf - 1| on the recomputed boundary
    double *ws;            // per wave: [NUp + 8][E] band-boundary row + the constant chunk of band 0
    int64_t P, B;          // B > 0: Gram, pair p = (p / B, p % B); B == 0: paired
    int Mrows, Ncp, Mc, Nc, NUp, nb;
    int per;               // pairs per wave (the equal share)
    double inv_sigma;
    int64_t ws_stride;     // doubles per wave
    WaveGroup wg;
    // the wave's stream of pairs (sk_wave_fused_mb.hip): C0 pairs fixed per wave, then one pair at a time drawn from `queue`
    unsigned long long *queue;   // the launch's counter (zeroed by the launcher; behind the boundary rows), nullptr: equal static shares
    int64_t q_first;
    int C0;
    int naive;             // _naive_solver stencil: c_12 = 0 (a = 1 + g/2, b = 1 exactly; see sk_wave_fused_mb.hip)
};

__device__ __forceinline__ void amb_store_through(double *p, d2_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void amb_begin(d2_t &t) { asm volatile("" : "=v"(t)); }
// one 16-byte LDS read, issued WITHOUT a wait (taken after a later s_waitcnt lgkmcnt(0): amb_take)
template <int OFF>
__device__ __forceinline__ void amb_read_pend(d2_t &t, unsigned a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(t) : "v"(a), "n"(OFF) : "memory");
}
__device__ __forceinline__ void amb_take(d2_t &o, d2_t &t) { asm volatile("" : "=v"(o) : "0"(t)); }

template <int ND>
__device__ __forceinline__ void amb_read_ydims(d2_t (&v)[ND], unsigned a_even, unsigned a_odd) {
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
__device__ __forceinline__ void amb_read_xrow(double (&x)[ND], unsigned a) {
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

// Y32 (fp32 inputs): the y ring holds the caller's fp32 points, half the bytes -- at FD = 16 what lets two waves share a SIMD -- and the
// squared distance is formed as the forward kernel forms it there, |x|^2 + |y|^2 - 2 <x, y> (static_kernels.py:70-73): FD FMAs per
// node, exact products of fp32 values.  Everything is still computed in fp64.
template <int DY, int RC, int FD, bool Y32>
__global__ __launch_bounds__(4 * WAVE) __attribute__((amdgpu_waves_per_eu((FD == 16 && (DY == 1 || !Y32)) ? 1 : 2))) void k_adj_fused_rbf_mb(const AdjMbParams prm) {
    constexpr int CW = 2;
    constexpr int R = RC << DY, S = CW << DY;
    static_assert(R == 4 || R == 2, "the column-edge reads below take R + 1 doubles out of R / 2 + 1 aligned 16-byte pieces");
    constexpr int L = AMB_L;
    constexpr int XROW = FD * 8, PPR = FD / 2;     // bytes / 16-byte pieces of one x row
    // one x window slab: [the x points of the node rows of the 8 lanes that start a band during the window]
    //                    [the pair's terminal COLUMN for their fine rows: 8 R + 2 doubles][the 16 bytes that hold scale[pair]]
    constexpr int XR_COL = 8 * RC * XROW, NPCOL = 4 * R + 1, XR_SC = XR_COL + NPCOL * 16;
    constexpr int NPIECES = XR_SC / 16 + 1, XSLAB = (XR_SC + 16 + 63) / 64 * 64;
    constexpr int FDY = Y32 ? FD / 2 : FD;   // rows of a y slab
    constexpr int YSLAB = FDY * 128 + (Y32 ? 128 : 0), NSLAB = L / 8 + 2, NDMA_Y = FDY * 128 / 1024;
    static_assert(!Y32 || FD == 16, "the fp32 ring is built for 16 dimensions");
    // band boundary entry of one unit: botR[S], the last node row's two values, the last coarse row's two weights (the forward state
    // restarts from the row the forward pass kept for the band: no error carried from band to band)
    constexpr int E = S + 4, NPB = E / 2, CHUNK = 8 * E * 8, CPIECES = CHUNK / 16;
    constexpr int NPC = 4 * S + 1, ECG = NPC * 16;   // terminal-row chunk (sk_wave_adj.hip)
    constexpr int OUTW = FD + 2;
    // PEND: LDS reads issued at the top of a macro-step stay in flight until the top rows need them.  Only in the variants whose
    // registers neither spill nor overflow into AGPRs: elsewhere the allocator may copy or spill a pending destination right behind the
    // read (seen: NaN gradients from a variant with 150 spilled VGPRs, last-bit run-to-run differences from one with 330 registers).
    constexpr bool PEND = DY == 2 && (FD == 8 || Y32);   // (the variants that compile to <= 256 registers WITHOUT spills: tests/test_abi.py checks)
    constexpr unsigned X_BASE = NSLAB * YSLAB, BI_BASE = X_BASE + AMB_X_SLOTS * XSLAB, BO_BASE = BI_BASE + 2 * CHUNK,
                       EC_BASE = BO_BASE + CHUNK, LDS_END = EC_BASE + 2 * ECG;
    extern __shared__ __attribute__((aligned(16))) char lds_block[];
    char *lds;
    const int64_t wave_id = wave_slot(prm.wg, lds_block, lds);
    if (wave_id < 0) return;
    const unsigned lds0 = lds_offset(lds);

    const int lam = threadIdx.x & (WAVE - 1);
    const int NUp = prm.NUp, nb = prm.nb;
    const int Mcp = nb * L * RC;
    const int MMp = Mcp << DY, NNp = (NUp * CW) << DY;
    const int EE = nb * NNp + MMp;   // edge doubles per pair
    const double sc = 1.0 / (double)(1 << (2 * DY));
    const double c_half = 0.5 * sc, c_12 = prm.naive ? 0.0 : sc * sc / 12.0;
    const double two_inv_sigma = 2.0 * prm.inv_sigma;
    const bool is_bot = lam == L - 1;
    const int lam7 = lam & 7;

    // ---- consumer cursor: unit u of band `band` of the wave's pair number ps (virtual unit t - lam of the stream) --------------
    int u, band, ps;
    {
        const int sig = floor_div(-lam, NUp);
        u = -lam - sig * NUp;
        ps = floor_div(sig, nb);
        band = sig - ps * nb;
    }
    int yslab, ypar;
    {
        const int s0 = floor_div(-lam, 8);
        yslab = ((s0 % NSLAB) + NSLAB) % NSLAB;
        ypar = s0 & 1;
    }
    // ---- the wave's pairs: positions 0 .. per-1 are pairs wave_id per + i ----------------------------------------------------------
    constexpr unsigned NOPAIR = 0xffffffffu;
    const unsigned P32 = (unsigned)prm.P;
    const int C0 = prm.C0;
    const unsigned base0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(wave_id * C0));
    unsigned cb1 = NOPAIR, cb2 = NOPAIR, cb3 = NOPAIR, cb0 = NOPAIR;   // drawn pair of position C0 + j in cb[j & 3]
    int have = 0;                     // drawn positions known so far
    int t_end = 0x7fffffff;           // macro-steps this wave runs: known once a draw comes back empty
    auto pair_at = [&](int i) __attribute__((always_inline)) -> unsigned {
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
            if (b == NOPAIR && t_end == 0x7fffffff) t_end = (C0 + have) * prm.nb * prm.NUp + (AMB_L - 1) + 1;
            const int kk = have & 3;
            const unsigned m0 = -(unsigned)(kk == 0), m1 = -(unsigned)(kk == 1), m2 = -(unsigned)(kk == 2), m3 = -(unsigned)(kk == 3);
            cb0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb0 & ~m0) | (b & m0)));
            cb1 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb1 & ~m1) | (b & m1)));
            cb2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb2 & ~m2) | (b & m2)));
            cb3 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb3 & ~m3) | (b & m3)));
            have += 1;
        }
    };

    auto split_b = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p % (uint32_t)prm.B);   // (32-bit: the launcher refuses P >= 2^31 - 2^20, and B <= P)
    };
    auto split_a = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p / (uint32_t)prm.B);
    };
    double *const wsrow = prm.ws + wave_id * prm.ws_stride;

    const unsigned my_x = lds0 + X_BASE + (unsigned)(lam7 * RC * XROW);
    const unsigned my_col = lds0 + X_BASE + (unsigned)(XR_COL + (7 - lam7) * R * 8);   // doubles (7-lam7) R .. +5 of the column piece
    const unsigned my_sc = lds0 + X_BASE + (unsigned)XR_SC;

    // ---- producers (wave-uniform control), once per window of 8 macro-steps ---------------------------------------------------
    // y slab s = virtual units [8s, 8s+8) in FLIPPED order: flipped unit u' of a band is original unit NUp-1-u' (every band re-reads
    // its pair's y)
    int y_pi = 0, y_band = 0, y_u0 = 0, y_slot = 0, y_par = 0;
    auto issue_y = [&]() {
        ensure(y_pi);
        const unsigned spy = pair_at(y_pi);
        const int64_t b = split_b(spy == NOPAIR ? 0 : (int64_t)spy);
        const int uo = NUp - 1 - (y_u0 + (lam & 7));
        if constexpr (Y32) {
            const double *row0 = prm.Yt + b * (FDY + 1) * (int64_t)prm.Ncp;
            __builtin_amdgcn_global_load_lds(row0 + ((lam >> 3) * (int64_t)prm.Ncp + (int64_t)uo * 2), (lds_void *)(lds + y_slot * YSLAB), 16, 0, 0);
            if (lam < 8)   // the |y|^2 row: 128 bytes
                __builtin_amdgcn_global_load_lds(row0 + (FDY * (int64_t)prm.Ncp + (int64_t)uo * 2), (lds_void *)(lds + y_slot * YSLAB + FDY * 128), 16,
                                                 0, 0);
        } else {
#pragma unroll
            for (int c = 0; c < NDMA_Y; ++c) {
                const int krow = (c * 8 + (lam >> 3)) ^ (y_par & 1);     // odd slabs: dimension rows swapped in pairs
                const double *src = prm.Yt + ((b * FD + krow) * (int64_t)prm.Ncp + (int64_t)uo * 2);
                __builtin_amdgcn_global_load_lds(src, (lds_void *)(lds + y_slot * YSLAB + c * 1024), 16, 0, 0);
            }
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
    // the window in which lane 0 sweeps units x_lam0 .. x_lam0+7 of band x_band of pair number x_pi: the x slab of the lanes
    // x_lam0 .. x_lam0+7, which start that band during the window; lane 0's boundary chunk; in band 0 its terminal-row chunk
    int x_pi = 0, x_band = 0, x_lam0 = 0, x_slot = 0;
    auto issue_x = [&]() {
        ensure(x_pi);
        const unsigned spx = pair_at(x_pi);
        const int64_t p = spx == NOPAIR ? 0 : (int64_t)spx;
        const int64_t a = split_a(p);
        const int lamj = x_lam0 < L ? x_lam0 : 0;        // nobody starts: fetch something valid
        const int gl0 = x_band * L + lamj;
        const double *xa = prm.Xr + a * prm.Mrows * FD;
        const double *ecol = prm.edges + p * EE + (nb * NNp - 2 + MMp - (gl0 + 8) * R);
        const double *scp = prm.scale ? reinterpret_cast<const double *>(reinterpret_cast<uintptr_t>(prm.scale + p) & ~(uintptr_t)15) : prm.Xr;
        char *dst = lds + X_BASE + x_slot * XSLAB;
#pragma unroll
        for (int c = 0; c < (NPIECES + 63) / 64; ++c) {
            const int idx = c * 64 + lam;
            if (idx < NPIECES) {
                const double *src;
                if (idx < 8 * RC * PPR) {
                    const int i = idx / PPR;
                    src = xa + (int64_t)(Mcp - 1 - (gl0 * RC + i)) * FD + (idx % PPR) * 2;
                } else if (idx < 8 * RC * PPR + NPCOL) {
                    src = ecol + 2 * (idx - 8 * RC * PPR);
                } else {
                    src = scp;
                }
                __builtin_amdgcn_global_load_lds(src, (lds_void *)(dst + c * 1024), 16, 0, 0);
            }
        }
        // lane 0's boundary entries, past the L1 (written through to L2 by this wave's flush); band 0: the constant chunk
#pragma unroll
        for (int c = 0; c < (CPIECES + 63) / 64; ++c) {
            const int idx = c * 64 + lam;
            if (idx < CPIECES) {
                const double *sb = (x_band == 0 ? wsrow + (int64_t)NUp * E : wsrow + (int64_t)x_lam0 * E) + idx * 2;
                __builtin_amdgcn_global_load_lds(sb, (lds_void *)(lds + BI_BASE + x_slot * CHUNK + c * 1024), 16, 0, 17);
            }
        }
        if (lam < NPC) {   // the band's top row of K (band 0: the terminal row) for lane 0's 8 units (sk_wave_adj.hip: issue_edge_chunk)
            const int k = NNp - (x_lam0 + LINE_UNITS) * S - 2 + 2 * lam;
            if (k >= 0)
                __builtin_amdgcn_global_load_lds(prm.edges + p * EE + (int64_t)x_band * NNp + k, (lds_void *)(lds + EC_BASE + x_slot * ECG), 16, 0, 0);
        }
        x_slot ^= 1;
        x_lam0 += 8;
        if (x_lam0 == NUp) {
            x_lam0 = 0;
            x_band += 1;
            if (x_band == nb) { x_band = 0; x_pi += 1; }
        }
    };
    // lane 63's staged chunk (its units f_pos .. f_pos+7) -> the wave's global row, written through to L2
    int f_pos = 0;
    auto flush_chunk = [&]() {
#pragma unroll
        for (int c = 0; c < (CPIECES + 63) / 64; ++c) {
            const int idx = c * 64 + lam;
            if (idx < CPIECES) {
                d2_t v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds0 + BO_BASE + (unsigned)(idx * 16)) : "memory");
                amb_store_through(wsrow + (int64_t)f_pos * E + idx * 2, v);
            }
        }
        f_pos += 8;
        if (f_pos == NUp) f_pos = 0;
    };

    // ---- state ------------------------------------------------------------------------------------------------------------
    double xr[RC][FD], xsn[RC];   // Y32: xsn = -|x_row|^2 / sigma
#pragma unroll
    for (int k = 0; k < RC; ++k) {
        xsn[k] = 0.0;
#pragma unroll
        for (int j = 0; j < FD; ++j) xr[k][j] = 0.0;
    }
    // accumulators per node row r_k = p_k + 1 (k < RC).  A macro-step's terms are added ONE STEP LATER, when both node columns they
    // belong to -- c1 of the step before (cv1p) and c2 = that step's first column (cv2) -- are the two columns of ONE unit of the y
    // ring, the previous one: one read per dimension pair, no column history in registers
    double cs[RC], accd[RC][FD], cv1p[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) {
        cs[k] = 0.0;
        cv1p[k] = 0.0;
#pragma unroll
        for (int j = 0; j < FD; ++j) accd[k][j] = 0.0;
    }
    double GownP[RC], GabvP = 0.0, lastOwn[2] = {0.0, 0.0};
    double wkP[RC], wupP = 0.0, lastW[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < RC; ++k) { GownP[k] = 0.0; wkP[k] = 0.0; }
    unsigned yq_e = lds0, yq_o = lds0;     // the previous unit's y (even / odd dimension rows)
    double leftR[R], botR[S], cornerR = 1.0;
    double leftF[R], botF[S], cornerF = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = 1.0; }
#pragma unroll
    for (int i = 0; i < S; ++i) { botR[i] = 1.0; botF[i] = 1.0; }
    ExpCoef expc;
    expc.init();
    double chk_val = 0.0;
    int64_t chk_pair = -1;
    double sx = 0.0, sx_d = 0.0;
    int valid = 0;
    bool row_ok[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) row_ok[k] = false;
    double *gp_cur = nullptr, *gp_prev = nullptr;     // Gpart row r_0 of this lane in the band being swept / the band before (null: none)
    double *n0_cur = nullptr, *n0_prev = nullptr;     // bottom lane in the last band: it also owns node row 0 -- N0 row of the pair
    double *gp_pair = nullptr;                        // Gpart block of the pair the lane is in (looked up once per pair)
    unsigned sp_cur = NOPAIR;                         // ... and its index
    const unsigned sc_par = (unsigned)((reinterpret_cast<uintptr_t>(prm.scale) >> 3) & 1u);

    {   // lanes ahead of their first band read slabs no DMA has written yet: make those finite
        const d2_t z = {0.0, 0.0};
        for (unsigned o = (unsigned)lam * 16u; o < LDS_END; o += WAVE * 16) lds_write_b128(lds0 + o, z);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the constant chunk of band 0: per entry ones (reverse state; the forward part is replaced by the terminal row; node values:
    // anything finite) and zero weights
#pragma unroll
    for (int c = 0; c < (CPIECES + 63) / 64; ++c) {
        const int idx = c * 64 + lam;
        if (idx < CPIECES) {
            const double v = (idx * 2) % E >= S + 2 ? 0.0 : 1.0;
            amb_store_through(wsrow + (int64_t)NUp * E + idx * 2, d2_t{v, v});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue_y();
    issue_x();

    for (int t0 = 0; t0 < t_end; t0 += 8) {
        // window of 8 macro-steps: what it consumes was issued a window ago; what the next one consumes is issued now
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue_y();
        issue_x();
        const unsigned slot_w = (unsigned)((t0 >> 3) & 1);
        const unsigned x_rd = slot_w * XSLAB;
        const unsigned bi_rd = lds0 + BI_BASE + slot_w * CHUNK;
        const unsigned ec_rd = lds0 + EC_BASE + slot_w * ECG;
        const int t_stop = t0 + 8 < t_end ? t0 + 8 : t_end;
        for (int t = t0; t < t_stop; ++t) {
        // -- lane 0's boundary entry of its unit (a uniform address: read by every lane as a broadcast) and, band 0, its S
        //    terminal-row values; no wait here: complete at the y read's lgkmcnt(0) below
        d2_t pend[NPB], bnd[NPB];
        double trow_p[S], trow[S];
        if constexpr (PEND) {
            const unsigned ba = bi_rd + (unsigned)((t & 7) * (E * 8));
#pragma unroll
            for (int i = 0; i < NPB; ++i) amb_begin(pend[i]);
            amb_read_pend<0>(pend[0], ba); amb_read_pend<16>(pend[1], ba); amb_read_pend<32>(pend[2], ba); amb_read_pend<48>(pend[3], ba);
            if constexpr (NPB > 4) { amb_read_pend<64>(pend[4], ba); amb_read_pend<80>(pend[5], ba); }
#pragma unroll
            for (int i = 0; i < S; ++i) async_begin(trow_p[i]);
            lds_read_f64_run<S>(trow_p, ec_rd + (unsigned)(((7 - (t & 7)) * S + 1) * 8));
        } else {   // blocking reads (see PEND)
            double ent[E];
            lds_read_block<E>(ent, bi_rd + (unsigned)((t & 7) * (E * 8)));
#pragma unroll
            for (int i = 0; i < NPB; ++i) bnd[i] = d2_t{ent[2 * i], ent[2 * i + 1]};
            lds_read_f64_block<S>(trow, ec_rd + (unsigned)(((7 - (t & 7)) * S + 1) * 8));
        }

        if (chk_pair >= 0) {
            atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
            chk_pair = -1;
        }
        const int uo = NUp - 1 - u;    // original unit: node columns c0 = 2uo, c1 = 2uo + 1 (c2 = 2uo + 2 is last step's c0)

        // -- start of a band: boundaries, terminal column, upstream gradient, this lane's x points, where its sums go
        if (u == 0) {
            asm volatile("");
            if (band == 0) {   // the lane enters a pair (1 / nb of the band starts): its index and the base of its partial sums
                asm volatile("");
                sp_cur = pair_at(ps);
                gp_pair = sp_cur != NOPAIR ? prm.Gpart + (int64_t)sp_cur * (int64_t)(Mcp + 1) * OUTW : nullptr;
            }
            const unsigned sp = sp_cur;
            const int gl = band * L + lam;
            valid = sp != NOPAIR ? 1 : 0;
#pragma unroll
            for (int k = 0; k < RC; ++k) row_ok[k] = Mcp - 1 - (gl * RC + k) < prm.Mc;
            const unsigned xa = my_x + x_rd;
#pragma unroll
            for (int k = 0; k < RC; ++k) amb_read_xrow<FD>(xr[k], xa + k * XROW);
            if constexpr (Y32) {
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    double q2 = 0.0;
#pragma unroll
                    for (int j = 0; j < FD; ++j) q2 = fma(xr[k][j], xr[k][j], q2);
                    xsn[k] = -q2 * prm.inv_sigma;
                }
            }
            double col[R + 2];
            {
                const unsigned ca_ = my_col + x_rd;
                if constexpr (R == 4) {
                    d2_t c3[3];
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(c3[0]), "=&v"(c3[1]), "=&v"(c3[2]) : "v"(ca_) : "memory");
                    col[0] = c3[0][0]; col[1] = c3[0][1]; col[2] = c3[1][0]; col[3] = c3[1][1]; col[4] = c3[2][0]; col[5] = c3[2][1];
                } else {
                    d2_t c2[2];
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(c2[0]), "=&v"(c2[1]) : "v"(ca_) : "memory");
                    col[0] = c2[0][0]; col[1] = c2[0][1]; col[2] = c2[1][0]; col[3] = c2[1][1];
                }
            }
            // col[1 + m] = K[MMp - gl R - R + m][NN], m = 0..R: the lane's fine rows bottom to top; K[0][NN] = 1 is not stored
            cornerR = 1.0;
            cornerF = col[1 + R];
#pragma unroll
            for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = col[R - i]; }
            if (gl * R + R == MMp) leftF[R - 1] = 1.0;
            double sv = 1.0;
            if (prm.scale && valid) sv = lds_read_f64(my_sc + x_rd + ((sc_par ^ (sp & 1u)) << 3));   // which half of the aligned 16 bytes holds scale[pair]
            gp_prev = gp_cur;
            n0_prev = n0_cur;
            gp_cur = gp_pair ? gp_pair + (Mcp - gl * RC) * OUTW : nullptr;
            n0_cur = nullptr;
            if (is_bot && band == nb - 1 && valid) n0_cur = prm.N0 + (int64_t)sp * (int64_t)(2 * NUp);
            if (sv != sv) valid = 0;      // NaN: a pair the rescue's screen took out of the sweep (its sums are stored as zeros)
            sx = valid ? sv : 0.0;
        }

        // -- top rows of the two states: the lane above's bottom row; lane 0: the boundary entry (reverse state: ones in band 0),
        //    in band 0 the pair's terminal row for the forward state
        if constexpr (PEND) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NPB; ++i) amb_take(bnd[i], pend[i]);
            lds_take<S>(trow, trow_p);
        }
        double topR[S], topF[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            double tf = trow[S - 1 - i];
            if (i == S - 1 && u == NUp - 1) tf = 1.0;     // K[.][0] = 1 is not stored
            topR[i] = dpp_shr1(botR[i], bnd[i >> 1][i & 1]);
            topF[i] = dpp_shr1(botF[i], tf);
        }
        // what the lane above evaluated / weighted one macro-step ago, for this unit's two columns; lane 0: what lane 63 had for
        // this unit one band earlier (band 0: nothing above contributes: zero weights, its first coarse row is padding)
        double Gabv[2], wup0[2];
        Gabv[0] = dpp_shr1(lastOwn[0], bnd[S / 2][0]);
        Gabv[1] = dpp_shr1(lastOwn[1], bnd[S / 2][1]);
        wup0[0] = dpp_shr1(lastW[0], bnd[S / 2 + 1][0]);
        wup0[1] = dpp_shr1(lastW[1], bnd[S / 2 + 1][1]);

        // -- nodes of this lane's rows at the unit's two node columns, eight dimensions of y at a time
        const unsigned ya = lds0 + (unsigned)(yslab * YSLAB + ((u & 7) << 4));
        const unsigned ya_e = Y32 ? ya : ya + (unsigned)(ypar << 7), ya_o = Y32 ? ya + 128u : ya + (unsigned)((ypar ^ 1) << 7);
        double Gown[RC][2];
        if constexpr (Y32) {
            d2_t raw[FDY];
            d2_t ysq_p, ysq;
            if constexpr (PEND) {
                amb_begin(ysq_p);
                amb_read_pend<FDY * 128>(ysq_p, ya);          // |y|^2 of the two columns
                amb_read_ydims<FDY>(raw, ya_e, ya_o);         // (its lgkmcnt(0) covers the read above)
                amb_take(ysq, ysq_p);
            } else {   // blocking (see PEND)
                amb_read_ydims<FDY>(raw, ya_e, ya_o);
                double t2[2];
                lds_read_row1<2>(t2, ya + (unsigned)(FDY * 128));
                ysq = d2_t{t2[0], t2[1]};
            }
            double xy[RC][CW];
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) xy[k][q] = 0.0;
#pragma unroll
            for (int jp = 0; jp < FDY; ++jp) {
                const f4_t f = __builtin_bit_cast(f4_t, raw[jp]);
#pragma unroll
                for (int q = 0; q < CW; ++q) {
                    const double y0 = (double)f[q], y1 = (double)f[2 + q];
#pragma unroll
                    for (int k = 0; k < RC; ++k) xy[k][q] = fma(xr[k][2 * jp + 1], y1, fma(xr[k][2 * jp], y0, xy[k][q]));
                }
            }
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q)   // -(|x|^2 + |y|^2 - 2<x,y>) / sigma (an infinite coordinate gives inf - inf = NaN as in the reference)
                    Gown[k][q] = exp_nonpos(fma(xy[k][q], two_inv_sigma, xsn[k] - ysq[q] * prm.inv_sigma), expc);
        } else {
            double d2[RC][CW];
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) d2[k][q] = 0.0;
#pragma unroll
            for (int h = 0; h < FD / 8; ++h) {
                d2_t yh[8];
                amb_read_ydims<8>(yh, ya_e + h * 1024u, ya_o + h * 1024u);
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int q = 0; q < CW; ++q)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const double df = xr[k][8 * h + j] - yh[j][q];
                            d2[k][q] = fma(df, df, d2[k][q]);
                        }
            }
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) Gown[k][q] = exp_nonpos(fma(-d2[k][q], prm.inv_sigma, d2[k][q] * 0.0), expc);
        }
        // -- increments of the RC x 2 coarse cells, the reference's order ((G11 + G00) - G10) - G01 (sigkernel.py:362-363);
        //    padding rows / columns carry none
        const bool c0_ok = 2 * uo < prm.Nc, c1_ok = 2 * uo + 1 < prm.Nc;
        double ginc[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const double b0 = k == 0 ? Gabv[0] : Gown[(k + RC - 1) % RC][0];      // G[p_k + 1][c0]
            const double b1 = k == 0 ? Gabv[1] : Gown[(k + RC - 1) % RC][1];      // G[p_k + 1][c1]
            const double b2 = k == 0 ? GabvP : GownP[(k + RC - 1) % RC];          // G[p_k + 1][c2]
            const double g0 = ((b1 + Gown[k][0]) - b0) - Gown[k][1];
            const double g1 = ((b2 + Gown[k][1]) - b1) - GownP[k];
            ginc[k][0] = (row_ok[k] && c0_ok) ? g0 : 0.0;
            ginc[k][1] = (row_ok[k] && c1_ok) ? g1 : 0.0;
        }
        double ca[RC][CW], cb[RC][CW], ca2[RC][CW], cib[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const double g = ginc[k][CW - 1 - q];   // flipped column order inside the unit
                const double g2 = g * g;
                ca[k][q] = fma(g2, c_12, fma(g, c_half, 1.0));
                cb[k][q] = fma(g2, -c_12, 1.0);
                cib[k][q] = fast_rcp(cb[k][q]);
                ca2[k][q] = ca[k][q] * cib[k][q];
            }

        // -- sweep the block, accumulate K * Krev per coarse cell
        double acc[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) acc[k][q] = 0.0;
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
            double aboveR = topR[cc], diagR = cc == 0 ? cornerR : topR[cc - 1];
            double aboveF = topF[cc], diagF = cc == 0 ? cornerF : topF[cc - 1];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int k = rr >> DY, q = cc >> DY;
                const double a = ca[k][q], b = cb[k][q], a2 = ca2[k][q], ib = cib[k][q];
                const double lR = leftR[rr], lF = leftF[rr];
                const double vR = fma(aboveR, a, fma(lR, a, -(diagR * b)));
                const double vF = fma(aboveF, a2, fma(lF, a2, -(diagF * ib)));
                acc[k][q] = fma(vF, diagR, acc[k][q]);
                diagR = lR; aboveR = vR; leftR[rr] = vR;
                diagF = lF; aboveF = vF; leftF[rr] = vF;
            }
            botR[cc] = aboveR;
            botF[cc] = aboveF;
        }
        cornerR = topR[S - 1];
        cornerF = topF[S - 1];

        // -- weights of the cells (original columns c0, c1), WITHOUT the pair's upstream gradient; zero outside the pair and in
        //    padding rows / columns (SELECTED: leftovers may hold anything, NaN included)
        double wk[RC][2];
        {
            const bool live = valid != 0;
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                wk[k][0] = (live && row_ok[k] && c0_ok) ? acc[k][1] * sc : 0.0;
                wk[k][1] = (live && row_ok[k] && c1_ok) ? acc[k][0] * sc : 0.0;
            }
        }
        // -- lane 63: this step's boundary entry (position = its unit) into the outgoing chunk
        if (is_bot) {
            const unsigned ea = lds0 + BO_BASE + (unsigned)((u & 7) * (E * 8));
#pragma unroll
            for (int cc = 0; cc < S; cc += 2) lds_write_b128(ea + cc * 8u, d2_t{botR[cc], botR[cc + 1]});
            lds_write_b128(ea + S * 8u, d2_t{Gown[RC - 1][0], Gown[RC - 1][1]});
            lds_write_b128(ea + (S + 2) * 8u, d2_t{wk[RC - 1][0], wk[RC - 1][1]});
        }
        // -- contraction: node rows r_k = p_k + 1 at node columns c1 (this unit's second) and c2 (the previous unit's first)
        double cv1[RC], cv2[RC];
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const double u0 = k == 0 ? wup0[0] : wk[(k + RC - 1) % RC][0];     // cells of coarse row p_k + 1
            const double u1 = k == 0 ? wup0[1] : wk[(k + RC - 1) % RC][1];
            const double u2 = k == 0 ? wupP : wkP[(k + RC - 1) % RC];
            const double g1 = k == 0 ? Gabv[1] : Gown[(k + RC - 1) % RC][1];   // G[r_k][c1]
            const double g2 = k == 0 ? GabvP : GownP[(k + RC - 1) % RC];       // G[r_k][c2]
            const double V1 = ((wk[k][0] + u1) - wk[k][1]) - u0;
            const double V2 = ((wk[k][1] + u2) - wkP[k]) - u1;
            cv1[k] = V1 * g1 * sx;
            cv2[k] = V2 * g2 * sx_d;
        }
        // node row p_{RC-1} from its own cells only (V[0][c] = w[0][c] - w[0][c-1]): node row 0 on the bottom lane of the last band.
        // Its weights go out per node column (N0); the c2 term of a band's first macro-step is node column 0 of the band BEFORE
        if (n0_cur || (u == 0 && n0_prev)) {
            asm volatile("");
            const double V1 = wk[RC - 1][1] - wk[RC - 1][0];
            const double V2 = wkP[RC - 1] - wk[RC - 1][1];
            const double c1v = V1 * Gown[RC - 1][1] * sx, c2v = V2 * GownP[RC - 1] * sx_d;
            if (n0_cur) n0_cur[2 * uo + 1] = c1v;
            if (u == 0) {
                if (n0_prev) n0_prev[0] = c2v;
            } else if (n0_cur) {
                n0_cur[2 * uo + 2] = c2v;
            }
        }
        // the terms of the PREVIOUS macro-step's c1 and of this step's c2: the two columns of the previous unit
#pragma unroll
        for (int k = 0; k < RC; ++k) cs[k] += cv1p[k] + cv2[k];
        if constexpr (Y32) {
            d2_t raw[FDY];
            amb_read_ydims<FDY>(raw, yq_e, yq_o);
#pragma unroll
            for (int jp = 0; jp < FDY; ++jp) {
                const f4_t f = __builtin_bit_cast(f4_t, raw[jp]);
                const double a0 = (double)f[0], a1 = (double)f[1], b0 = (double)f[2], b1 = (double)f[3];
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    accd[k][2 * jp] = fma(cv1p[k], a1, fma(cv2[k], a0, accd[k][2 * jp]));
                    accd[k][2 * jp + 1] = fma(cv1p[k], b1, fma(cv2[k], b0, accd[k][2 * jp + 1]));
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < FD / 8; ++h) {
                d2_t yh[8];
                amb_read_ydims<8>(yh, yq_e + h * 1024u, yq_o + h * 1024u);
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) accd[k][8 * h + j] = fma(cv1p[k], yh[j][1], fma(cv2[k], yh[j][0], accd[k][8 * h + j]));
            }
        }
        if (u == 0) {
            // a band's first macro-step: with the above the sums of the rows being left are complete -- write them out, start over
            asm volatile("");
            if (gp_prev) {
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    double *dst = gp_prev - (int64_t)k * OUTW;
                    *reinterpret_cast<d2_t *>(dst) = d2_t{cs[k], 0.0};
#pragma unroll
                    for (int j = 0; j < FD; j += 2) *reinterpret_cast<d2_t *>(dst + 2 + j) = d2_t{accd[k][j], accd[k][j + 1]};
                }
            }
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                cs[k] = 0.0;
#pragma unroll
                for (int j = 0; j < FD; ++j) accd[k][j] = 0.0;
            }
        }
#pragma unroll
        for (int k = 0; k < RC; ++k) cv1p[k] = cv1[k];
        yq_e = ya_e;
        yq_o = ya_o;
        // -- histories for the next macro-step (and for the lane below, which reads lastOwn / lastW at its top)
        wupP = wup0[0];
        GabvP = Gabv[0];
#pragma unroll
        for (int k = 0; k < RC; ++k) { wkP[k] = wk[k][0]; GownP[k] = Gown[k][0]; }
        lastOwn[0] = Gown[RC - 1][0]; lastOwn[1] = Gown[RC - 1][1];
        lastW[0] = wk[RC - 1][0]; lastW[1] = wk[RC - 1][1];
        sx_d = sx;

        // -- self-check on the last flipped unit of the band (see sk_wave_adj.hip): K on the j = 0 boundary must come out as 1
        if (u == NUp - 1 && prm.err && valid) {
            double e = 0.0;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) e = fmax(e, fabs(leftF[rr] - 1.0));
            chk_val = e;
            chk_pair = (int64_t)sp_cur;
        }

        // -- close the step
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
        // lane 63's unit is t - 63: its chunk of 8 units is complete when (t + 1) & 7 == 7 and goes out then; the window's closing
        // wait (top of the next window) acknowledges it before the fetch of the window that may need it is issued
        if (((t + 1) & 7) == 7 && t >= L - 1 + 7) flush_chunk();
        }
    }
    if (chk_pair >= 0)
        atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the LINEAR static kernel on long paths: the same stream, a much lighter step -----------------------------------------------------
// Increments straight from the path differences (no nodes), W contracted on the spot with the y differences of its own unit
// (sk_wave_adj_fused.hip): T[pair][coarse row][:] = s_ab sum_q W[p][q] (y[q+1] - y[q]), no coupling to the lane above -- the band
// boundary entry is the reverse state's bottom row alone (S doubles), and a lane's sums go out when it enters the next band.
// Gpart: [P][Mcp][FD], FLIPPED coarse rows (row f = Mcp - 1 - p, the layout of sk_wave_adj_fused.hip); Xr = s^2 (x[p+1] - x[p]),
// Yt = y[q+1] - y[q] dimension-major; edges from sk_solve_fwd_static_* (kind 0) with edges.
template <int DY, int RC, int FD>
__global__ __launch_bounds__(4 * WAVE) __attribute__((amdgpu_waves_per_eu(FD == 16 ? 1 : 2))) void k_adj_fused_linear_mb(const AdjMbParams prm) {
    constexpr int CW = 2;
    constexpr int R = RC << DY, S = CW << DY;
    static_assert(R == 4, "the column-edge reads below take five doubles out of three aligned 16-byte pieces");
    constexpr int L = AMB_L;
    constexpr int XROW = FD * 8, PPR = FD / 2;
    constexpr int XR_COL = 8 * RC * XROW, NPCOL = 4 * R + 1, XR_SC = XR_COL + NPCOL * 16;
    constexpr int NPIECES = XR_SC / 16 + 1, XSLAB = (XR_SC + 16 + 63) / 64 * 64;
    constexpr int YSLAB = FD * 128, NSLAB = L / 8 + 2, NDMA_Y = YSLAB / 1024;
    constexpr int E = S, NPB = E / 2, CHUNK = 8 * E * 8, CPIECES = CHUNK / 16;     // boundary entry: botR[S]
    constexpr int NPC = 4 * S + 1, ECG = NPC * 16;
    constexpr int OUTW = FD;
    constexpr bool PEND = false;   // blocking reads throughout (see k_adj_fused_rbf_mb): the step is light, two variants spill a register
    constexpr unsigned X_BASE = NSLAB * YSLAB, BI_BASE = X_BASE + AMB_X_SLOTS * XSLAB, BO_BASE = BI_BASE + 2 * CHUNK,
                       EC_BASE = BO_BASE + CHUNK, LDS_END = EC_BASE + 2 * ECG;
    extern __shared__ __attribute__((aligned(16))) char lds_block[];
    char *lds;
    const int64_t wave_id = wave_slot(prm.wg, lds_block, lds);
    if (wave_id < 0) return;
    const unsigned lds0 = lds_offset(lds);

    const int lam = threadIdx.x & (WAVE - 1);
    const int NUp = prm.NUp, nb = prm.nb;
    const int Mcp = nb * L * RC;
    const int MMp = Mcp << DY, NNp = (NUp * CW) << DY;
    const int EE = nb * NNp + MMp;
    const double sc = 1.0 / (double)(1 << (2 * DY));
    const double c_half = 0.5 * sc, c_12 = prm.naive ? 0.0 : sc * sc / 12.0;
    const bool is_bot = lam == L - 1;
    const int lam7 = lam & 7;

    int u, band, ps;
    {
        const int sig = floor_div(-lam, NUp);
        u = -lam - sig * NUp;
        ps = floor_div(sig, nb);
        band = sig - ps * nb;
    }
    int yslab, ypar;
    {
        const int s0 = floor_div(-lam, 8);
        yslab = ((s0 % NSLAB) + NSLAB) % NSLAB;
        ypar = s0 & 1;
    }
    constexpr unsigned NOPAIR = 0xffffffffu;
    const unsigned P32 = (unsigned)prm.P;
    const int C0 = prm.C0;
    const unsigned base0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(wave_id * C0));
    unsigned cb1 = NOPAIR, cb2 = NOPAIR, cb3 = NOPAIR, cb0 = NOPAIR;   // drawn pair of position C0 + j in cb[j & 3]
    int have = 0;                     // drawn positions known so far
    int t_end = 0x7fffffff;           // macro-steps this wave runs: known once a draw comes back empty
    auto pair_at = [&](int i) __attribute__((always_inline)) -> unsigned {
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
            if (b == NOPAIR && t_end == 0x7fffffff) t_end = (C0 + have) * prm.nb * prm.NUp + (AMB_L - 1) + 1;
            const int kk = have & 3;
            const unsigned m0 = -(unsigned)(kk == 0), m1 = -(unsigned)(kk == 1), m2 = -(unsigned)(kk == 2), m3 = -(unsigned)(kk == 3);
            cb0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb0 & ~m0) | (b & m0)));
            cb1 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb1 & ~m1) | (b & m1)));
            cb2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb2 & ~m2) | (b & m2)));
            cb3 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb3 & ~m3) | (b & m3)));
            have += 1;
        }
    };

    auto split_b = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p % (uint32_t)prm.B);   // (32-bit: the launcher refuses P >= 2^31 - 2^20, and B <= P)
    };
    auto split_a = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p / (uint32_t)prm.B);
    };
    double *const wsrow = prm.ws + wave_id * prm.ws_stride;
    const unsigned my_x = lds0 + X_BASE + (unsigned)(lam7 * RC * XROW);
    const unsigned my_col = lds0 + X_BASE + (unsigned)(XR_COL + (7 - lam7) * R * 8);
    const unsigned my_sc = lds0 + X_BASE + (unsigned)XR_SC;

    int y_pi = 0, y_band = 0, y_u0 = 0, y_slot = 0, y_par = 0;
    auto issue_y = [&]() {
        ensure(y_pi);
        const unsigned spy = pair_at(y_pi);
        const int64_t b = split_b(spy == NOPAIR ? 0 : (int64_t)spy);
        const int uo = NUp - 1 - (y_u0 + (lam & 7));
#pragma unroll
        for (int c = 0; c < NDMA_Y; ++c) {
            const int krow = (c * 8 + (lam >> 3)) ^ (y_par & 1);
            const double *src = prm.Yt + ((b * FD + krow) * (int64_t)prm.Ncp + (int64_t)uo * 2);
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
        const unsigned spx = pair_at(x_pi);
        const int64_t p = spx == NOPAIR ? 0 : (int64_t)spx;
        const int64_t a = split_a(p);
        const int lamj = x_lam0 < L ? x_lam0 : 0;
        const int gl0 = x_band * L + lamj;
        const double *xa = prm.Xr + a * prm.Mrows * FD;
        const double *ecol = prm.edges + p * EE + (nb * NNp - 2 + MMp - (gl0 + 8) * R);
        const double *scp = prm.scale ? reinterpret_cast<const double *>(reinterpret_cast<uintptr_t>(prm.scale + p) & ~(uintptr_t)15) : prm.Xr;
        char *dst = lds + X_BASE + x_slot * XSLAB;
#pragma unroll
        for (int c = 0; c < (NPIECES + 63) / 64; ++c) {
            const int idx = c * 64 + lam;
            if (idx < NPIECES) {
                const double *src;
                if (idx < 8 * RC * PPR) {
                    const int i = idx / PPR;
                    src = xa + (int64_t)(Mcp - 1 - (gl0 * RC + i)) * FD + (idx % PPR) * 2;    // coarse row (>= Mc: zero padding)
                } else if (idx < 8 * RC * PPR + NPCOL) {
                    src = ecol + 2 * (idx - 8 * RC * PPR);
                } else {
                    src = scp;
                }
                __builtin_amdgcn_global_load_lds(src, (lds_void *)(dst + c * 1024), 16, 0, 0);
            }
        }
        if (lam < CPIECES) {
            const double *sb = (x_band == 0 ? wsrow + (int64_t)NUp * E : wsrow + (int64_t)x_lam0 * E) + lam * 2;
            __builtin_amdgcn_global_load_lds(sb, (lds_void *)(lds + BI_BASE + x_slot * CHUNK), 16, 0, 17);
        }
        if (lam < NPC) {
            const int k = NNp - (x_lam0 + LINE_UNITS) * S - 2 + 2 * lam;
            if (k >= 0)
                __builtin_amdgcn_global_load_lds(prm.edges + p * EE + (int64_t)x_band * NNp + k, (lds_void *)(lds + EC_BASE + x_slot * ECG), 16, 0, 0);
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
        if (lam < CPIECES) {
            d2_t v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds0 + BO_BASE + (unsigned)(lam * 16)) : "memory");
            amb_store_through(wsrow + (int64_t)f_pos * E + lam * 2, v);
        }
        f_pos += 8;
        if (f_pos == NUp) f_pos = 0;
    };

    double dxr[RC][FD], tacc[RC][FD];
#pragma unroll
    for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int j = 0; j < FD; ++j) { dxr[k][j] = 0.0; tacc[k][j] = 0.0; }
    double leftR[R], botR[S], cornerR = 1.0;
    double leftF[R], botF[S], cornerF = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = 1.0; }
#pragma unroll
    for (int i = 0; i < S; ++i) { botR[i] = 1.0; botF[i] = 1.0; }
    double chk_val = 0.0;
    int64_t chk_pair = -1;
    double sx = 0.0;
    int valid = 0;
    bool row_ok[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) row_ok[k] = false;
    double *gp_cur = nullptr, *gp_pair = nullptr;   // Gpart rows of this lane in the band being swept; the pair's block
    unsigned sp_cur = NOPAIR;
    const unsigned sc_par = (unsigned)((reinterpret_cast<uintptr_t>(prm.scale) >> 3) & 1u);

    {
        const d2_t z = {0.0, 0.0};
        for (unsigned o = (unsigned)lam * 16u; o < LDS_END; o += WAVE * 16) lds_write_b128(lds0 + o, z);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lam < CPIECES) amb_store_through(wsrow + (int64_t)NUp * E + lam * 2, d2_t{1.0, 1.0});   // the constant chunk of band 0: ones
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue_y();
    issue_x();

    for (int t0 = 0; t0 < t_end; t0 += 8) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue_y();
        issue_x();
        const unsigned slot_w = (unsigned)((t0 >> 3) & 1);
        const unsigned x_rd = slot_w * XSLAB;
        const unsigned bi_rd = lds0 + BI_BASE + slot_w * CHUNK;
        const unsigned ec_rd = lds0 + EC_BASE + slot_w * ECG;
        const int t_stop = t0 + 8 < t_end ? t0 + 8 : t_end;
        for (int t = t0; t < t_stop; ++t) {
        d2_t pend[NPB], bnd[NPB];
        double trow_p[S], trow[S];
        if constexpr (PEND) {
            const unsigned ba = bi_rd + (unsigned)((t & 7) * (E * 8));
#pragma unroll
            for (int i = 0; i < NPB; ++i) amb_begin(pend[i]);
            amb_read_pend<0>(pend[0], ba);
            if constexpr (NPB > 1) amb_read_pend<16>(pend[1], ba);
            if constexpr (NPB > 2) { amb_read_pend<32>(pend[2], ba); amb_read_pend<48>(pend[3], ba); }
#pragma unroll
            for (int i = 0; i < S; ++i) async_begin(trow_p[i]);
            lds_read_f64_run<S>(trow_p, ec_rd + (unsigned)(((7 - (t & 7)) * S + 1) * 8));
        } else {   // blocking reads (see PEND)
            double ent[E];
            lds_read_block<E>(ent, bi_rd + (unsigned)((t & 7) * (E * 8)));
#pragma unroll
            for (int i = 0; i < NPB; ++i) bnd[i] = d2_t{ent[2 * i], ent[2 * i + 1]};
            lds_read_f64_block<S>(trow, ec_rd + (unsigned)(((7 - (t & 7)) * S + 1) * 8));
        }

        if (chk_pair >= 0) {
            atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
            chk_pair = -1;
        }
        const int uo = NUp - 1 - u;

        if (u == 0) {
            asm volatile("");
            // the sums of the band being left: complete (a cell's weight is contracted in its own macro-step)
            if (gp_cur) {
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int j = 0; j < FD; j += 2) *reinterpret_cast<d2_t *>(gp_cur + (int64_t)k * OUTW + j) = d2_t{tacc[k][j], tacc[k][j + 1]};
            }
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int j = 0; j < FD; ++j) tacc[k][j] = 0.0;
            if (band == 0) {   // the lane enters a pair (1 / nb of the band starts): its index and the base of its partial sums
                asm volatile("");
                sp_cur = pair_at(ps);
                gp_pair = sp_cur != NOPAIR ? prm.Gpart + (int64_t)sp_cur * (int64_t)Mcp * OUTW : nullptr;
            }
            const unsigned sp = sp_cur;
            const int gl = band * L + lam;
            valid = sp != NOPAIR ? 1 : 0;
#pragma unroll
            for (int k = 0; k < RC; ++k) row_ok[k] = Mcp - 1 - (gl * RC + k) < prm.Mc;
            const unsigned xa = my_x + x_rd;
#pragma unroll
            for (int k = 0; k < RC; ++k) amb_read_xrow<FD>(dxr[k], xa + k * XROW);
            double col[6];
            {
                d2_t c3[3];
                const unsigned ca_ = my_col + x_rd;
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(c3[0]), "=&v"(c3[1]), "=&v"(c3[2]) : "v"(ca_) : "memory");
                col[0] = c3[0][0]; col[1] = c3[0][1]; col[2] = c3[1][0]; col[3] = c3[1][1]; col[4] = c3[2][0]; col[5] = c3[2][1];
            }
            cornerR = 1.0;
            cornerF = col[1 + R];
#pragma unroll
            for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = col[R - i]; }
            if (gl * R + R == MMp) leftF[R - 1] = 1.0;
            double sv = 1.0;
            if (prm.scale && valid) sv = lds_read_f64(my_sc + x_rd + ((sc_par ^ (sp & 1u)) << 3));
            gp_cur = gp_pair ? gp_pair + gl * RC * OUTW : nullptr;   // flipped coarse row gl RC + k
            if (sv != sv) valid = 0;      // NaN: taken out of the sweep by the rescue's screen (its sums are stored as zeros)
            sx = valid ? sv : 0.0;
        }

        // -- top rows
        if constexpr (PEND) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NPB; ++i) amb_take(bnd[i], pend[i]);
            lds_take<S>(trow, trow_p);
        }
        double topR[S], topF[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            double tf = trow[S - 1 - i];
            if (i == S - 1 && u == NUp - 1) tf = 1.0;
            topR[i] = dpp_shr1(botR[i], bnd[i >> 1][i & 1]);
            topF[i] = dpp_shr1(botF[i], tf);
        }

        // -- y differences of the unit (original column order inside the unit), increments, coefficients
        const unsigned ya = lds0 + (unsigned)(yslab * YSLAB + ((u & 7) << 4));
        const unsigned ya_e = ya + (unsigned)(ypar << 7), ya_o = ya + (unsigned)((ypar ^ 1) << 7);
        const bool c0_ok = 2 * uo < prm.Nc, c1_ok = 2 * uo + 1 < prm.Nc;
        double ginc[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) ginc[k][q] = 0.0;
#pragma unroll
        for (int h = 0; h < FD / 8; ++h) {
            d2_t yh[8];
            amb_read_ydims<8>(yh, ya_e + h * 1024u, ya_o + h * 1024u);
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) ginc[k][q] = fma(dxr[k][8 * h + j], yh[j][q], ginc[k][q]);
        }
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            ginc[k][0] = (row_ok[k] && c0_ok) ? ginc[k][0] : 0.0;
            ginc[k][1] = (row_ok[k] && c1_ok) ? ginc[k][1] : 0.0;
        }
        double ca[RC][CW], cb[RC][CW], ca2[RC][CW], cib[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const double g = ginc[k][CW - 1 - q];   // flipped column order inside the unit
                const double g2 = g * g;
                ca[k][q] = fma(g2, c_12, fma(g, c_half, 1.0));
                cb[k][q] = fma(g2, -c_12, 1.0);
                cib[k][q] = fast_rcp(cb[k][q]);
                ca2[k][q] = ca[k][q] * cib[k][q];
            }

        double acc[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) acc[k][q] = 0.0;
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
            double aboveR = topR[cc], diagR = cc == 0 ? cornerR : topR[cc - 1];
            double aboveF = topF[cc], diagF = cc == 0 ? cornerF : topF[cc - 1];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int k = rr >> DY, q = cc >> DY;
                const double a = ca[k][q], b = cb[k][q], a2 = ca2[k][q], ib = cib[k][q];
                const double lR = leftR[rr], lF = leftF[rr];
                const double vR = fma(aboveR, a, fma(lR, a, -(diagR * b)));
                const double vF = fma(aboveF, a2, fma(lF, a2, -(diagF * ib)));
                acc[k][q] = fma(vF, diagR, acc[k][q]);
                diagR = lR; aboveR = vR; leftR[rr] = vR;
                diagF = lF; aboveF = vF; leftF[rr] = vF;
            }
            botR[cc] = aboveR;
            botF[cc] = aboveF;
        }
        cornerR = topR[S - 1];
        cornerF = topF[S - 1];

        if (is_bot) {
            const unsigned ea = lds0 + BO_BASE + (unsigned)((u & 7) * (E * 8));
#pragma unroll
            for (int cc = 0; cc < S; cc += 2) lds_write_b128(ea + cc * 8u, d2_t{botR[cc], botR[cc + 1]});
        }

        // -- W of the RC x 2 coarse cells (with the pair's upstream gradient), contracted with the y differences of their columns
        {
            const bool live = valid != 0;
            const double wsc = sc * sx;
            double w0[RC], w1[RC];
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                w0[k] = (live && row_ok[k] && c0_ok) ? acc[k][1] * wsc : 0.0;     // original columns 0, 1
                w1[k] = (live && row_ok[k] && c1_ok) ? acc[k][0] * wsc : 0.0;
            }
#pragma unroll
            for (int h = 0; h < FD / 8; ++h) {
                d2_t yh[8];
                amb_read_ydims<8>(yh, ya_e + h * 1024u, ya_o + h * 1024u);
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) tacc[k][8 * h + j] = fma(w0[k], yh[j][0], fma(w1[k], yh[j][1], tacc[k][8 * h + j]));
            }
        }

        if (u == NUp - 1 && prm.err && valid) {
            double e = 0.0;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) e = fmax(e, fabs(leftF[rr] - 1.0));
            chk_val = e;
            chk_pair = (int64_t)sp_cur;
        }

        u += 1;
        if (((t + 1) & 7) == lam7) {
            yslab = yslab + 1 == NSLAB ? 0 : yslab + 1;
            ypar ^= 1;
            if (u == NUp) {
                u = 0;
                band += 1;
                if (band == nb) { band = 0; ps += 1; }
            }
        }
        if (((t + 1) & 7) == 7 && t >= L - 1 + 7) flush_chunk();
        }
    }
    if (chk_pair >= 0)
        atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct AmbPlan {
    int RC, S, NUp, nb, fd;
    size_t lds_bytes;
    int64_t ws_stride, edge_doubles;
    bool ok;
};

AmbPlan amb_plan(int Mc, int Nc, int dyadic, int D, bool y32 = false, int kind = 1) {
    AmbPlan pl{};
    pl.ok = false;
    if (dyadic < 0 || dyadic > 2 || D < 1 || D > 16) return pl;
    pl.RC = dyadic == 0 ? (kind == 1 ? 2 : 4) : dyadic == 1 ? 2 : 1;   // (rbf at dyadic 0: two rows per lane -- four would need > 256 registers)
    pl.S = 2 << dyadic;
    pl.fd = D <= 8 ? 8 : 16;
    const int NU = kind == 1 ? (Nc + 2) / 2 : (Nc + 1) / 2;  // the forward's units (sk_wave_fused_mb.hip: mb_plan)
    pl.NUp = (NU + LINE_UNITS - 1) / LINE_UNITS * LINE_UNITS;
    if (pl.NUp < AMB_L + 16) pl.NUp = AMB_L + 16;            // band boundary slack: shorter second paths are swept with (masked) padding units
    // rbf: the node rows must fit the lanes (the first lane-row is padding); linear: the coarse rows
    pl.nb = (Mc + (kind == 1 ? 1 : 0) + AMB_L * pl.RC - 1) / (AMB_L * pl.RC);
    const int R = pl.RC << dyadic, E = kind == 1 ? pl.S + 4 : pl.S;
    const size_t xslab = ((size_t)8 * pl.RC * pl.fd * 8 + (4 * R + 1) * 16 + 16 + 63) / 64 * 64;
    pl.lds_bytes = (size_t)(AMB_L / 8 + 2) * ((y32 ? pl.fd / 2 + 1 : pl.fd) * 128) + AMB_X_SLOTS * xslab + (size_t)3 * 8 * E * 8 + (size_t)2 * (4 * pl.S + 1) * 16;
    pl.ws_stride = (int64_t)(pl.NUp + 8) * E;
    pl.edge_doubles = (int64_t)pl.nb * pl.NUp * pl.S + (int64_t)pl.nb * AMB_L * R;
    pl.ok = true;
    return pl;
}

template <int DY, int RC, int FD, bool Y32 = false, int KIND = 1>
int launch_amb(AdjMbParams prm, const AmbPlan &pl, void *ws, size_t ws_bytes, hipStream_t s) {
    auto kern = [] {
        if constexpr (KIND == 0) return k_adj_fused_linear_mb<DY, RC, FD>;
        else return k_adj_fused_rbf_mb<DY, RC, FD, Y32>;
    }();
    static const int vgprs = [&] {
        hipFuncAttributes attr;
        return hipFuncGetAttributes(&attr, (const void *)kern) == hipSuccess && attr.numRegs > 0 ? attr.numRegs : 256;
    }();
    // resident waves per CU: whole workgroups of wpb waves by LDS, by registers, at most two per SIMD
    const int wpb0 = wave_group(pl.lds_bytes, 1 << 20, knobs().adjmb_wpb).wpb;
    int wpc = (int)((160 * 1024) / (pl.lds_bytes * wpb0)) * wpb0;
    const int by_regs = 4 * (512 / ((vgprs + 7) & ~7));
    if (wpc > by_regs) wpc = by_regs;
    if (knobs().adjmb_wpc > 0 && wpc > knobs().adjmb_wpc) wpc = knobs().adjmb_wpc;
    if (wpc > 8) wpc = 8;
    if (wpc >= wpb0) wpc = wpc / wpb0 * wpb0;
    if (wpc < 1) wpc = 1;
    const int64_t max_waves = (int64_t)device_cu_count() * wpc;
    int64_t waves = prm.P < max_waves ? prm.P : max_waves;
    const int64_t per = (prm.P + waves - 1) / waves;
    if (prm.P >= 0x7ff00000LL || per > 0x1fffffff / ((int64_t)prm.nb * prm.NUp)) return SK_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < (size_t)waves * (size_t)pl.ws_stride * sizeof(double) + 64) return SK_ERR_WORKSPACE;
    const int pct = knobs().adjmb_q_static > 0 ? (knobs().adjmb_q_static > 100 ? 100 : knobs().adjmb_q_static) : 50;
    if (waves == max_waves && per >= 8 && pct < 100) {
        // the launch fills the chip: `pct` per cent of the equal share is dealt out up front, the rest is drawn pair by pair
        prm.C0 = (int)(per * pct / 100);
        prm.queue = reinterpret_cast<unsigned long long *>(static_cast<double *>(ws) + (size_t)waves * (size_t)pl.ws_stride);
        prm.q_first = waves * (int64_t)prm.C0;
        if (hipMemsetAsync(prm.queue, 0, sizeof(unsigned long long), s) != hipSuccess) return SK_ERR_LAUNCH;
    } else {
        waves = (prm.P + per - 1) / per;
        prm.C0 = (int)per;
        prm.queue = nullptr;
        prm.q_first = prm.P;
    }
    prm.per = (int)per;
    prm.ws = static_cast<double *>(ws);
    prm.ws_stride = pl.ws_stride;
    prm.wg = wave_group(pl.lds_bytes, waves, knobs().adjmb_wpb);
    const size_t lds_block = wave_group_lds(prm.wg);
    if (lds_block > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
    SK_LAUNCH(kern, dim3(wave_group_blocks(prm.wg)), dim3(WAVE * prm.wg.wpb), lds_block, s, prm);
    return check_launch();
}

}  // namespace

// Layout query of sk_rbf_adjoint_fused_mb_f64 (0 / false outside the kernel's scope): rows of Xr the caller provides per path,
// node rows and doubles per node row of gpart, node columns of n0, edge doubles per pair (what sk_solve_fwd_static_* writes with
// `edges`), bands, units, workspace bytes.
bool adj_fused_mb_layout(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows, int *rows, int *outw, int *ncols, int64_t *edge_doubles,
                         int *nb, int *nup, size_t *ws_bytes, int kind) {
    const AmbPlan pl = amb_plan(Mc, Nc, dyadic, D, false, kind);
    if (!pl.ok || P <= 0) return false;
    if (mrows) *mrows = pl.nb * AMB_L * pl.RC + 8;
    if (rows) *rows = pl.nb * AMB_L * pl.RC + (kind == 1 ? 1 : 0);
    if (outw) *outw = kind == 1 ? pl.fd + 2 : pl.fd;
    if (ncols) *ncols = 2 * pl.NUp;
    if (edge_doubles) *edge_doubles = pl.edge_doubles;
    if (nb) *nb = pl.nb;
    if (nup) *nup = pl.NUp;
    const int64_t max_waves = (int64_t)device_cu_count() * 8;
    if (ws_bytes) *ws_bytes = (size_t)(P < max_waves ? P : max_waves) * (size_t)pl.ws_stride * sizeof(double) + 64;   // + the work counter
    return true;
}

int launch_adj_fused_rbf_mb(const double *Xr, const void *Yt_any, int yt_f32, int64_t A, int64_t B, int Mrows, int Ncp, int D, int fd, const Geom &g,
                            double inv_sigma, const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *n0,
                            size_t n0_doubles, double *err, void *ws, size_t ws_bytes, const FusedRescue *rescue, const double *Yt64,
                            hipStream_t s) {
    if (B < 0 || g.P != (B > 0 ? A * B : A)) return SK_ERR_UNSUPPORTED;
    const double *Yt = static_cast<const double *>(Yt_any);
    const bool y32 = yt_f32 != 0;
    const AmbPlan pl = amb_plan(g.Mc, g.Nc, g.dyadic, D, y32);
    if (!pl.ok || fd != pl.fd || (y32 && pl.fd != 16)) return SK_ERR_UNSUPPORTED;
    if (Ncp < pl.NUp * 2 || (Ncp & 1) || Mrows < pl.nb * AMB_L * pl.RC + 1) return SK_ERR_UNSUPPORTED;
    if (g.Nc > 2 * pl.NUp - 1) return SK_ERR_UNSUPPORTED;         // node column 2 NUp must be padding
    const int64_t rows = (int64_t)pl.nb * AMB_L * pl.RC + 1;
    if (gpart_doubles < (size_t)(g.P * rows * (pl.fd + 2)) || n0_doubles < (size_t)(g.P * 2 * pl.NUp)) return SK_ERR_WORKSPACE;
    AdjMbParams prm{};
    prm.Xr = Xr; prm.Yt = Yt; prm.edges = edges; prm.scale = scale; prm.Gpart = gpart; prm.N0 = n0; prm.err = err;
    prm.P = g.P; prm.B = B; prm.Mrows = Mrows; prm.Ncp = Ncp; prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = pl.NUp; prm.nb = pl.nb;
    prm.inv_sigma = inv_sigma;
    prm.naive = g.naive;
    // device-side rescue (sk_adj_fused_rescue.hip): the workspace starts with the swept upstream gradient (screened pairs NaN: the
    // sweep stores zeros for them), and their exact, stored-grid share is added to gpart afterwards -- one pair per chunk here
    void *rws = nullptr;
    size_t rws_bytes = 0;
    if (rescue && rescue->ws && Yt64) {
        const size_t head = sizeof(double) * (size_t)((g.P + 1) / 2 * 2);
        if (rescue->ws_bytes <= head) return SK_ERR_WORKSPACE;
        rws = (char *)rescue->ws + head;
        rws_bytes = rescue->ws_bytes - head;
        if (rescue->kfinal) {
            const int rc = launch_fused_screen(rescue->kfinal, scale, g.P, rescue->screen, (double *)rescue->ws, err, s);
            if (rc != SK_OK) return rc;
            prm.scale = (const double *)rescue->ws;
        }
    }
    int rc;
    if (y32 && g.dyadic == 0) return SK_ERR_UNSUPPORTED;
    if (y32) rc = g.dyadic == 1 ? launch_amb<1, 2, 16, true>(prm, pl, ws, ws_bytes, s) : launch_amb<2, 1, 16, true>(prm, pl, ws, ws_bytes, s);
    else if (g.dyadic == 0) rc = pl.fd == 8 ? launch_amb<0, 2, 8>(prm, pl, ws, ws_bytes, s) : launch_amb<0, 2, 16>(prm, pl, ws, ws_bytes, s);
    else if (g.dyadic == 1) rc = pl.fd == 8 ? launch_amb<1, 2, 8>(prm, pl, ws, ws_bytes, s) : launch_amb<1, 2, 16>(prm, pl, ws, ws_bytes, s);
    else rc = pl.fd == 8 ? launch_amb<2, 1, 8>(prm, pl, ws, ws_bytes, s) : launch_amb<2, 1, 16>(prm, pl, ws, ws_bytes, s);
    if (rc != SK_OK || !rws) return rc;
    ChunkSplit cs{};          // one pair per chunk, slot = pair
    cs.nr = 1; cs.nch = (int)(B > 0 ? B : 1); cs.cpr = cs.nch; cs.gpr = (int64_t)1 << 62; cs.size[0] = 1; cs.off[0] = 0;
    return launch_fused_rescue(1, Xr, Yt64, scale, err, rescue->tol, gpart, nullptr, A, B, Mrows, Ncp, D, g, (int)rows, pl.fd + 2, 0, inv_sigma, cs,
                               g.P, rws, rws_bytes, s, pl.fd, n0, 2 * pl.NUp, rescue->kfinal);
}

// LinearKernel on long paths: gpart [P][nb 64 RC][fd], FLIPPED coarse rows (row f = rows - 1 - p), per PAIR; summed over the pairs of
// an x_a and flipped back it is the T of sk_linear_adjoint_* (dL/dx[m] = s^2 (T[m-1] - T[m])).
int launch_adj_fused_linear_mb(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Ncp, int D, int fd, const Geom &g,
                               const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *err, void *ws,
                               size_t ws_bytes, const FusedRescue *rescue, hipStream_t s) {
    if (B < 0 || g.P != (B > 0 ? A * B : A)) return SK_ERR_UNSUPPORTED;
    const AmbPlan pl = amb_plan(g.Mc, g.Nc, g.dyadic, D, false, 0);
    if (!pl.ok || fd != pl.fd) return SK_ERR_UNSUPPORTED;
    if (Ncp < pl.NUp * 2 || (Ncp & 1) || Mrows < pl.nb * AMB_L * pl.RC) return SK_ERR_UNSUPPORTED;
    if (g.Nc > 2 * pl.NUp) return SK_ERR_UNSUPPORTED;
    const int64_t rows = (int64_t)pl.nb * AMB_L * pl.RC;
    if (gpart_doubles < (size_t)(g.P * rows * pl.fd)) return SK_ERR_WORKSPACE;
    AdjMbParams prm{};
    prm.Xr = dXr; prm.Yt = dYt; prm.edges = edges; prm.scale = scale; prm.Gpart = gpart; prm.N0 = nullptr; prm.err = err;
    prm.P = g.P; prm.B = B; prm.Mrows = Mrows; prm.Ncp = Ncp; prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = pl.NUp; prm.nb = pl.nb;
    prm.inv_sigma = 0.0;
    prm.naive = g.naive;
    void *rws = nullptr;
    size_t rws_bytes = 0;
    if (rescue && rescue->ws) {
        const size_t head = sizeof(double) * (size_t)((g.P + 1) / 2 * 2);
        if (rescue->ws_bytes <= head) return SK_ERR_WORKSPACE;
        rws = (char *)rescue->ws + head;
        rws_bytes = rescue->ws_bytes - head;
        if (rescue->kfinal) {
            const int rc = launch_fused_screen(rescue->kfinal, scale, g.P, rescue->screen, (double *)rescue->ws, err, s);
            if (rc != SK_OK) return rc;
            prm.scale = (const double *)rescue->ws;
        }
    }
    int rc;
    if (pl.fd == 8)
        rc = g.dyadic == 0 ? launch_amb<0, 4, 8, false, 0>(prm, pl, ws, ws_bytes, s)
           : g.dyadic == 1 ? launch_amb<1, 2, 8, false, 0>(prm, pl, ws, ws_bytes, s) : launch_amb<2, 1, 8, false, 0>(prm, pl, ws, ws_bytes, s);
    else
        rc = g.dyadic == 0 ? launch_amb<0, 4, 16, false, 0>(prm, pl, ws, ws_bytes, s)
           : g.dyadic == 1 ? launch_amb<1, 2, 16, false, 0>(prm, pl, ws, ws_bytes, s) : launch_amb<2, 1, 16, false, 0>(prm, pl, ws, ws_bytes, s);
    if (rc != SK_OK || !rws) return rc;
    ChunkSplit cs{};          // one pair per chunk, slot = pair
    cs.nr = 1; cs.nch = (int)(B > 0 ? B : 1); cs.cpr = cs.nch; cs.gpr = (int64_t)1 << 62; cs.size[0] = 1; cs.off[0] = 0;
    return launch_fused_rescue(0, dXr, dYt, scale, err, rescue->tol, gpart, nullptr, A, B, Mrows, Ncp, D, g, (int)rows, pl.fd, 0, 0.0, cs, g.P, rws,
                               rws_bytes, s, pl.fd, nullptr, 0, rescue->kfinal);
}

}  // namespace sk
