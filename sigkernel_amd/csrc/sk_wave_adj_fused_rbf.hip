// sk_wave_adj_fused_rbf.hip -- the adjoint solver with the RBF static kernel fused in, both ways (the RBF twin of
// sk_wave_adj_fused.hip):
//   * the increments of the reverse sweep are the 4-corner differences of nodes G[p][q] = exp(-|x_p - y_q|^2 / sigma) that
//     the kernel evaluates itself from the path points (the two LDS rings of sk_wave_fused.hip, filled back to front);
//   * the weights W = d k / d inc are pushed on the spot through the 4-corner difference and the exponential,
//         V[r][c] = w[r-1][c-1] + w[r][c] - w[r-1][c] - w[r][c-1]            (d L / d G[r][c],  w = s_ab W)
//         dL/dx_r = sum_c V[r][c] G[r][c] (-2/sigma) (x_r - y_c) = (-2/sigma) (x_r sum_c V G  -  sum_c V G y_c),
//     so a lane keeps 1 + D accumulators per node row (cs = sum V G, accd = sum V G y) over the pairs of its lane group and
//     neither the increments nor W nor G_static ever exist in HBM.  Replaces, for RBFKernel, sk_static_increments +
//     sk_solve_adj(EDGES_GIVEN) + sk_static_adjoint, i.e. sigkernel.py:419-502 (prep_backward) + :404-416.
//
// Why the flipped sweep makes this cheap.  In flipped coordinates lane `lam` owns the ORIGINAL coarse rows
// p_k = Mcp-1 - (lam RC + k) and walks the original columns from right to left.  It evaluates the TOP node row of each of
// its coarse rows (RC x 2 exps per macro-step); the node row under its first coarse row (p_0 + 1) is the last row of the lane
// ABOVE, which is one macro-step ahead and therefore has it already (one DPP pair, no lag); the third node column of a unit
// (2uo+2) is the first column of the unit swept one macro-step earlier (kept).  Likewise a node's V needs the cells to its
// lower right, which are the lane above's / the previous step's: V of node columns 2uo+1 and 2uo+2 is complete at the step
// that sweeps unit uo.  Node column 0 of a pair completes during the first macro-step of the next pair (whose own
// contribution there is masked padding), hence one extra macro-step at the end.  Node row 0 (which no lane has as a "row
// below") is accumulated by every lane for its last row and written by the bottom lane only.
//
// Decomposition, edges, self-check: as sk_wave_adj_fused.hip (PPG consecutive pairs of ONE x_a per lane group, partial
// sums stored per group and added by the host in a fixed order; terminal edges from sk_solve_fwd_rbf_edges_f64; the
// terminal ROW arrives through LDS chunks as in sk_wave_adj.hip).
// Scope: fp64, dyadic 1..2, path dim <= 8 (ND = 4 variants for dim <= 4), one band per pair with M <= L RC, N - 1 <= 2 NUp - 1;
// dyadic 0: default stencil, two coarse rows per lane (M <= 128).
#include "sk_wave_common.h"

namespace sk {
namespace {

#ifndef SK_ADJR_HOLD_Y
#define SK_ADJR_HOLD_Y 1
#endif
constexpr int RFD = 8;                 // dims carried by the staged arrays
constexpr int RY_SLAB = RFD * 128;
constexpr int RX_SLOTS = 2;
constexpr int YW = 6;                  // doubles per (pair, node column) of the second-argument sums: S0, 0, S1[0..4); 10 for dims 5..8
constexpr int yw_of(int nd) { return nd + 2; }

// One x window slab (per lane group, ring slot and lap): everything the 8 lanes that start a pair during the window need, in
// ONE run of 16-byte DMA pieces:
//   [0, 8 RC 64)            the x points of their node rows, position i = (lam & 7) RC + k <-> node row Mcp-1 - (lamj RC + i)
//   [XR_UP, +64)            the node row above position 0 (the last row of the lane above the window's first lane)
//   [XR_COL, +(4R+1) 16)    the pair's terminal COLUMN K[j][NN], j = MMp-(lamj+8)R-1 .. MMp-lamj R  (8R + 2 doubles)
//   [XR_SC, +16)            the pair's upstream gradient, as the aligned 16 bytes that hold scale[pair]
template <int RC, int R> struct XSlab {
    static constexpr int XR_UP = 8 * RC * 64;
    static constexpr int XR_COL = XR_UP + 64;
    static constexpr int NPCOL = 4 * R + 1;
    static constexpr int XR_SC = XR_COL + NPCOL * 16;
    static constexpr int NPIECES = (XR_SC + 16) / 16;
    static constexpr int BYTES = (XR_SC + 16 + 63) / 64 * 64;
};

struct AdjRbfParams {
    const double *Xr;      // [A][Mrows][8]  x_p (points), zero rows / dims beyond M / D
    const double *Yt;      // [B][8][Ncp]    y_q, dimension-major, zero-padded
    const double *edges;   // [P][NNp + MMp] strip layout (strip_geom)
    const double *scale;   // [P] upstream gradient per pair, nullable
    double *Gpart;         // [P / PPG][L*RC + 1][OUTW]  per node row: cs, 0, accd[0..ND)
    double *err;           // [P] zero-initialised: worst |Kf - 1| on the recomputed boundary
    double *Ypart;         // YSIDE: [P][2 NUp][YW] per pair and node column of y_b: S0 = sum_r V G, S1 = sum_r V G x_r, both
                           // WITHOUT the pair's upstream gradient (the caller weights the pairs when it folds them over a)
    int64_t P, B;
    int Mrows, Ncp, Mc, Nc, NUp, logL, PPG, n_steps;
    ChunkSplit cs;         // chunk sizes by wave age rank; PPG / n_steps are the equal split's
    double inv_sigma;
    WaveGroup wg;
    int naive;             // _naive_solver stencil: c_12 = 0 (a = 1 + g/2, b = 1 exactly; see sk_wave_fused_mb.hip)
};

template <int ND>
__device__ __forceinline__ void lds_read_ydims(d2_t (&v)[ND], unsigned a_even, unsigned a_odd);
template <>
__device__ __forceinline__ void lds_read_ydims<8>(d2_t (&v)[8], unsigned a_even, unsigned a_odd) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %8 offset:256\n\tds_read_b128 %3, %9 offset:256\n\t"
                 "ds_read_b128 %4, %8 offset:512\n\tds_read_b128 %5, %9 offset:512\n\tds_read_b128 %6, %8 offset:768\n\t"
                 "ds_read_b128 %7, %9 offset:768\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(a_even), "v"(a_odd) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_ydims<4>(d2_t (&v)[4], unsigned a_even, unsigned a_odd) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:256\n\tds_read_b128 %3, %5 offset:256\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a_even), "v"(a_odd) : "memory");
}
// ND consecutive doubles of an x row (64-byte rows), one wait
template <int ND>
__device__ __forceinline__ void lds_read_xpt(double (&x)[ND], unsigned a) {
    if constexpr (ND == 8) {
        lds_read_row1<8>(x, a);
    } else {
        lds_read_row1<4>(x, a);
    }
}
// The second-argument carry (YSIDE: S0 / S1[0..4) of node columns c1, c2 over the node rows of the lanes above) travels DOWN THE WAVE
// in registers: lane l + 1 sweeps at macro-step t + 1 the unit lane l swept at t, so one wave_shr:1 of the ten sums per macro-step (20 DPP
// moves, bound_ctrl zero for lane 0; the top lanes of the other lane groups are cleared under the exec mask) carries them along.  Until
// late in round 6 it went through five 65-slot LDS pieces (read with the y points, written back): 10 KB of LDS traffic per wave and
// macro-step and 5.2 KB of the CU's LDS per wave -- the piece size was what kept eight waves of C4's y-side launch on a CU (160,640 of
// 163,840 bytes).  Same-box A/B (profiles/r06_yside_dpp_ab.txt): C4 314.4-315.6 -> 313.8-314.3 ms, swapped RBF gradients of 512 x 64
// points 4.03 -> 3.87 (d = 0), 4.54 -> 4.32 ms (d = 1).
// the y points of a macro-step (four dims) AND the top lane's S terminal-row values, one wait instead of two
template <int S>
__device__ __forceinline__ void lds_read_ydims_trow(d2_t (&v)[4], double (&t)[S], unsigned a_even, unsigned a_odd, unsigned ta);
template <>
__device__ __forceinline__ void lds_read_ydims_trow<4>(d2_t (&v)[4], double (&t)[4], unsigned a_even, unsigned a_odd, unsigned ta) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %8 offset:256\n\tds_read_b128 %3, %9 offset:256\n\t"
                 "ds_read_b64 %4, %10\n\tds_read_b64 %5, %10 offset:8\n\tds_read_b64 %6, %10 offset:16\n\tds_read_b64 %7, %10 offset:24\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                 : "v"(a_even), "v"(a_odd), "v"(ta) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_ydims_trow<2>(d2_t (&v)[4], double (&t)[2], unsigned a_even, unsigned a_odd, unsigned ta) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %6 offset:256\n\tds_read_b128 %3, %7 offset:256\n\t"
                 "ds_read_b64 %4, %8\n\tds_read_b64 %5, %8 offset:8\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(t[0]), "=&v"(t[1])
                 : "v"(a_even), "v"(a_odd), "v"(ta) : "memory");
}

// Everything a lane reads when it starts a pair (four-dimension variants), issued together and handed over by ONE wait: its NX / 2 node
// rows (two 16-byte pieces each, 64-byte rows), NC pieces of the pair's terminal column and the upstream gradient.  Some lane starts a pair
// in EVERY macro-step, so the wave pays for this block every step; as four reads with a wait each it cost a lone wave four LDS round
// trips per step (profiles/r06_small_launch_pmc.txt: nothing hides them at one wave per SIMD).
template <int NX, int NC>
__device__ __forceinline__ void lds_read_pair_start(d2_t (&x)[NX], d2_t (&c)[NC], double &sv, unsigned xa, unsigned ca, unsigned sa);
template <>
__device__ __forceinline__ void lds_read_pair_start<4, 3>(d2_t (&x)[4], d2_t (&c)[3], double &sv, unsigned xa, unsigned ca, unsigned sa) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:80\n\t"
                 "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:16\n\tds_read_b128 %6, %9 offset:32\n\tds_read_b64 %7, %10\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(sv)
                 : "v"(xa), "v"(ca), "v"(sa) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_pair_start<4, 2>(d2_t (&x)[4], d2_t (&c)[2], double &sv, unsigned xa, unsigned ca, unsigned sa) {
    asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:64\n\tds_read_b128 %3, %7 offset:80\n\t"
                 "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:16\n\tds_read_b64 %6, %9\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(sv)
                 : "v"(xa), "v"(ca), "v"(sa) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_pair_start<2, 3>(d2_t (&x)[2], d2_t (&c)[3], double &sv, unsigned xa, unsigned ca, unsigned sa) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\t"
                 "ds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\tds_read_b128 %4, %7 offset:32\n\tds_read_b64 %5, %8\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(sv)
                 : "v"(xa), "v"(ca), "v"(sa) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_pair_start<2, 2>(d2_t (&x)[2], d2_t (&c)[2], double &sv, unsigned xa, unsigned ca, unsigned sa) {
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\t"
                 "ds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:16\n\tds_read_b64 %4, %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(sv)
                 : "v"(xa), "v"(ca), "v"(sa) : "memory");
}

// YS: 0 = first-argument sums; 1 = both (the triangular K_XX; dims <= 4); 2 = the SECOND-argument sums only (route FUSED_SWAP: all the
// swapped call needs -- no first-argument accumulators, no second read of the y points; the only second-argument form of dims 5..8)
// PAIRED: a paired batch (B == 0) with SEVERAL pairs per lane group (sk_wave_adj_fused.hip): a lane stores and clears its sums when a pair
// of its run is complete -- at the FIRST step of the next one, whose c2 terms still belong to the pair before
template <int DY, int RC, bool FULLWAVE, int ND, int YS, bool PAIRED = false>
__global__ __launch_bounds__(4 * WAVE) __attribute__((amdgpu_waves_per_eu(2))) void k_adj_fused_rbf(const AdjRbfParams prm) {
    constexpr bool YSIDE = YS != 0;
    static_assert(!PAIRED || YS == 0, "several pairs per lane group: first-argument sums only");
    static_assert(YS != 1 || ND == 4, "both sets of sums fit for paths of dim <= 4 only");
    constexpr int CW = 2;
    constexpr int R = RC << DY, S = CW << DY, r = 1 << DY;
    typedef XSlab<RC, R> XS;
    constexpr int XSLAB = XS::BYTES;
    constexpr int OUTW = ND + 2;
    constexpr int ECG = (4 * S + 1) * 16;    // terminal-row chunk of one lane group (sk_wave_adj.hip)
    constexpr int NPC = 4 * S + 1;
    static_assert(R == 4 || R == 2, "the column-edge reads below take R + 2 doubles out of aligned 16-byte pieces");
    // YONLY at ND = 8: the 9 (RC + 1) first-argument accumulators and the previous unit's y points make room for the 18 carried sums
    // and the node row above
    constexpr bool YONLY = YS == 2;
    constexpr int NCAR = 1 + ND;           // S0, S1[0..ND) per node column
    constexpr int YWK = yw_of(ND);
    extern __shared__ __attribute__((aligned(16))) char lds_block[];
    char *lds;
    const int64_t wave_id = wave_slot(prm.wg, lds_block, lds);
    if (wave_id < 0) return;
    const unsigned lds0 = lds_offset(lds);

    const int lane = threadIdx.x & (WAVE - 1);
    const int L = 1 << prm.logL, G = WAVE >> prm.logL;
    const int lam = lane & (L - 1), grp = lane >> prm.logL;
    const int NUp = prm.NUp;
    const int Mcp = L * RC;
    const int MMp = Mcp << DY, NNp = (NUp * CW) << DY;
    const int NSLAB = (L >> 3) + 2;
    const unsigned y_bytes = (unsigned)(NSLAB * RY_SLAB);
    const unsigned x_base0 = (unsigned)G * y_bytes;
    const double sc = 1.0 / (double)(1 << (2 * DY));
    const double c_half = 0.5 * sc, c_12 = prm.naive ? 0.0 : sc * sc / 12.0;

    // ---- consumer state (flipped coordinates; one band per pair) ---------------------------------------------------------
    int u, ps;
    {
        ps = floor_div(-lam, NUp);
        u = -lam - ps * NUp;
    }
    int yslab, ypar;
    {
        const int s0 = floor_div(-lam, 8);
        yslab = ((s0 % NSLAB) + NSLAB) % NSLAB;
        ypar = (s0 + grp) & 1;
    }
    const int lam7 = lam & 7;
    // all pairs of the group share one a; how many they are depends on the wave's age rank (ChunkSplit, sk_wave_common.h)
    int64_t pair0, gslot;
    int PPG;
    chunk_share(prm.cs, wave_id * G + grp, prm.B > 0 ? prm.P / prm.B : prm.P, prm.B, prm.P, pair0, gslot, PPG);
    // the lane group's OWN pairs (what is summed); the wave sweeps as many as its longest group has (one rank per wave; uneven chunks
    // differ by one pair, sk_wave_adj_fused.hip)
    const int ppg_own = PPG;
    PPG = __builtin_amdgcn_readfirstlane(PPG);
    if (prm.cs.uneven)
        for (int gq = 1; gq < G; ++gq) PPG = max(PPG, __builtin_amdgcn_readlane(ppg_own, gq << prm.logL));
    const int n_steps = PPG * NUp + (L - 1) + 1;   // + 1: node column 0 of the last pair completes one step later
    auto group_first = [&](int g) -> int64_t { return readlane64(pair0, g << prm.logL); };
    const bool is_top = lam == 0;
    const bool is_bot = lam == L - 1;
    const unsigned my_y = lds0 + (unsigned)grp * y_bytes;
    const int JMAX = (L + NUp - 1) / NUp;
    const unsigned my_slab = lds0 + x_base0 + (unsigned)((grp * RX_SLOTS * JMAX) * XSLAB + (lam / NUp) * XSLAB);
    const unsigned my_x = my_slab + (unsigned)(lam7 * RC * 64);
    const unsigned my_xup = my_slab + (unsigned)(lam7 == 0 ? XS::XR_UP : lam7 * RC * 64 - 64);   // the node row above this lane's first
    const unsigned my_col = my_slab + (unsigned)(XS::XR_COL + (7 - lam7) * R * 8);                // doubles (7-lam7)R .. +5 of the column piece
    const unsigned my_sc = my_slab + (unsigned)XS::XR_SC;
    const unsigned ec_base = lds0 + x_base0 + (unsigned)(G * RX_SLOTS * JMAX * XSLAB);   // terminal-row chunks behind the rings
    const unsigned ec_slot = (unsigned)(G * ECG);
    bool row_ok[RC];   // this lane's coarse rows that exist (p_k < Mc)
#pragma unroll
    for (int k = 0; k < RC; ++k) row_ok[k] = Mcp - 1 - (lam * RC + k) < prm.Mc;

    // ---- producers: the rings of sk_wave_fused.hip (points), filled in FLIPPED order --------------------------------------
    auto split_b = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p % (uint32_t)prm.B);   // (32-bit: the launcher refuses P >= 2^31 - 2^20, and B <= P)
    };
    auto split_a = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p / (uint32_t)prm.B);
    };
    int y_pi = 0, y_u0 = 0, y_slot = 0, y_par = 0;
    auto issue_y = [&]() {
        for (int g = 0; g < G; ++g) {
            int64_t p = group_first(g) + y_pi;
            if (y_pi >= PPG || p >= prm.P) p = 0;
            const int64_t b = split_b(p);
            const int krow = (lane >> 3) ^ ((y_par + g) & 1);
            const int uo = NUp - 1 - (y_u0 + (lane & 7));      // flipped unit -> original unit (node columns 2uo, 2uo+1)
            const double *src = prm.Yt + ((b * RFD + krow) * (int64_t)prm.Ncp + (int64_t)uo * 2);
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(lds + g * y_bytes + y_slot * RY_SLAB), 16, 0, 0);
        }
        y_slot = y_slot + 1 == NSLAB ? 0 : y_slot + 1;
        y_par ^= 1;
        y_u0 += 8;
        if (y_u0 == NUp) { y_u0 = 0; y_pi += 1; }
    };
    // x window slabs (XSlab): the points of the 8 lanes' node rows, the row above them, the pair's terminal column for their
    // fine rows and its upstream gradient -- one run of 16-byte pieces per lane group and lap
    const int E = NNp + MMp;
    int x_q0 = 0, x_lam0 = 0, x_slot = 0;
    auto issue_x = [&]() {
        for (int j = 0; j < JMAX; ++j) {
            const int lamj = x_lam0 + j * NUp, pi = x_q0 - j;
            if (lamj >= L) break;
            for (int g = 0; g < G; ++g) {
                // out of the group's range: something valid -- and of the group's OWN a: the lanes reload their x points at every
                // pair start, and the second-argument sums of a pair's node column 0 are completed one macro-step into the next
                const int64_t gf = group_first(g);
                int64_t p = gf + pi;
                if (pi < 0 || pi >= PPG || p >= prm.P) p = gf < prm.P ? gf : 0;
                const int64_t a = split_a(p);
                char *dst = lds + x_base0 + ((g * RX_SLOTS + x_slot) * JMAX + j) * XSLAB;
                const double *xa = prm.Xr + a * prm.Mrows * RFD;
                const double *ecol = prm.edges + p * E + (NNp - 2 + MMp - (lamj + 8) * R);
                const double *scp = prm.scale ? reinterpret_cast<const double *>(reinterpret_cast<uintptr_t>(prm.scale + p) & ~(uintptr_t)15) : prm.Xr;   // the aligned 16 bytes that hold scale[p]
#pragma unroll
                for (int c = 0; c < (XS::NPIECES + 63) / 64; ++c) {
                    const int idx = c * 64 + lane;
                    if (idx < XS::NPIECES) {
                        const double *src;
                        if (idx < 32 * RC + 4) {
                            const int i = idx < 32 * RC ? (idx >> 2) : -1;
                            src = xa + (Mcp - 1 - (lamj * RC + i)) * RFD + (idx & 3) * 2;
                        } else if (idx < 32 * RC + 4 + XS::NPCOL) {
                            src = ecol + 2 * (idx - (32 * RC + 4));
                        } else {
                            src = scp;
                        }
                        __builtin_amdgcn_global_load_lds(src, (lds_void *)(dst + c * 1024), 16, 0, 0);
                    }
                }
            }
        }
        x_slot = x_slot + 1 == RX_SLOTS ? 0 : x_slot + 1;
        x_lam0 += 8;
        if (x_lam0 == NUp) { x_lam0 = 0; x_q0 += 1; }
    };
    // terminal ROW of the pair the top lanes are in, one window ahead (sk_wave_adj.hip: issue_edge_chunk)
    int ec_u0 = 0, ec_ps = 0, ec_fill = 0;
    auto issue_edge_chunk = [&]() {
        for (int c = 0; c * WAVE < G * NPC; ++c) {
            const int idx = c * WAVE + lane, g = idx / NPC, i = idx - g * NPC;
            int64_t pr = gather64(pair0, (g < G ? g : 0) << prm.logL) + ec_ps;   // (g differs per lane here)
            pr = (ec_ps >= PPG || pr >= prm.P) ? 0 : pr;
            const int k = NNp - (ec_u0 + LINE_UNITS) * S - 2 + 2 * i;
            if (g < G && k >= 0)
                __builtin_amdgcn_global_load_lds(prm.edges + pr * E + k,
                                                 (lds_void *)(lds + x_base0 + G * RX_SLOTS * JMAX * XSLAB + ec_fill * (G * ECG) + c * 1024), 16, 0, 0);
        }
        ec_fill ^= 1;
        ec_u0 += LINE_UNITS;
        if (ec_u0 == NUp) { ec_u0 = 0; ec_ps += 1; }
    };

    double xr[RC][ND], xup[YSIDE ? ND : 1];
#pragma unroll
    for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int j = 0; j < ND; ++j) xr[k][j] = 0.0;
#pragma unroll
    for (int j = 0; j < (YSIDE ? ND : 1); ++j) xup[j] = 0.0;
    // accumulators per node row r_k = p_k + 1 (k < RC) and, [RC], node row p_{RC-1} (node row 0 on the bottom lane)
    double cs[YONLY ? 1 : RC + 1], accd[YONLY ? 1 : RC + 1][ND];
#pragma unroll
    for (int k = 0; k < (YONLY ? 1 : RC + 1); ++k) {
        cs[k] = 0.0;
#pragma unroll
        for (int j = 0; j < ND; ++j) accd[k][j] = 0.0;
    }
    // node history: this lane's rows at node column 2uo+2 (the previous step's first column), the same of the row above, what
    // the lane below takes from this lane (its last row's two columns of the previous step), and y at column 2uo+2
    double GownP[RC], GabvP = 0.0, lastOwn[2] = {0.0, 0.0}, yP[ND];
    double wkP[RC], wupP = 0.0, lastW[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < RC; ++k) { GownP[k] = 0.0; wkP[k] = 0.0; }
#pragma unroll
    for (int j = 0; j < ND; ++j) yP[j] = 0.0;
    double leftR[R], botR[S], cornerR = 1.0;
    double leftF[R], botF[S], cornerF = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = 1.0; }
#pragma unroll
    for (int i = 0; i < S; ++i) { botR[i] = 1.0; botF[i] = 1.0; }
    ExpCoef expc;   // polynomial coefficients in VGPRs: as SGPR pairs they spill the scalar state (v_readlane in the loop)
    expc.init();
    double chk_val = 0.0;
    int64_t chk_pair = -1;
    // upstream gradient of the pair being swept: sx for what the step adds at node column c1, sx_d (its value one step ago) at
    // c2 -- at the first step of a pair the c2 terms still belong to the pair before.  The weights themselves stay unscaled (the
    // second-argument sums are weighted by the caller) and are SELECTED to zero outside the group's pairs.
    double sx = 0.0, sx_d = 0.0;
    int valid = 0, valid_d = 0;      // (valid_d, PAIRED: whether the pair BEFORE the one being swept is one of the group's)
    d2_t car[YSIDE ? NCAR : 1];    // YSIDE: S0 / S1[0..ND) of node columns (c1, c2), summed over the node rows of the lanes above
#pragma unroll
    for (int i = 0; i < (YSIDE ? NCAR : 1); ++i) car[i] = d2_t{0.0, 0.0};
    double *yp_cur = nullptr, *yp_prev = nullptr;   // YSIDE: the Ypart blocks of this lane's pair and of the one before (null: none)
    // per-lane constants of the pair-start block below (some lane starts a pair in EVERY macro-step, so the wave pays for that block
    // every step: 32-bit compares and one multiply-add instead of 64-bit pair arithmetic): how many of the group's pairs exist,
    // which half of its aligned 16 bytes holds scale[pair0], the Ypart block of pair0
    int ps_end;
    {
        const int64_t rem = prm.P - pair0;
        ps_end = rem <= 0 ? 0 : (rem < (int64_t)ppg_own ? (int)rem : ppg_own);
    }
    const unsigned sc_par0 = (unsigned)(((reinterpret_cast<uintptr_t>(prm.scale) >> 3) ^ (uintptr_t)pair0) & 1u);
    double *const yp_base = YSIDE ? prm.Ypart + pair0 * (int64_t)(2 * NUp * YWK) : nullptr;
    asm volatile("" : "+v"(ps_end));

    {   // lanes ahead of their first pair read slabs no DMA has written yet: make those finite
        const int total = (int)(G * y_bytes + G * RX_SLOTS * JMAX * XSLAB + 2 * G * ECG);
        const d2_t z = {0.0, 0.0};
        for (int o = lane * 16; o < total; o += WAVE * 16) lds_write_b128(lds0 + (unsigned)o, z);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    issue_y();
    issue_x();
    issue_edge_chunk();

    for (int t0 = 0; t0 < n_steps; t0 += 8) {
        // window of 8 macro-steps: what it consumes was issued a window ago; what the next one consumes is issued now
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue_y();
        issue_x();
        issue_edge_chunk();
        const unsigned x_rd = (unsigned)(((t0 >> 3) % RX_SLOTS) * JMAX * XSLAB);
        const unsigned ec_rd = ec_base + (unsigned)(((t0 >> 3) & 1) * ec_slot + grp * ECG);
        const int t_end = t0 + 8 < n_steps ? t0 + 8 : n_steps;
        for (int t = t0; t < t_end; ++t) {
        // the top lane's terminal-row values of this macro-step (no wait: complete at the y read's lgkmcnt(0) below)
        // (left in flight only in the variants without spills -- dyadic 2, dims <= 4: C4's --; the others spill 16-127 registers and
        // read it blocking where it is needed: tools/check_async_hazards.py, scan_pressure)
        constexpr bool TPEND = DY == 2 && ND == 4 && !(YSIDE && !FULLWAVE);   // (that one: 16 VGPRs short since the carry lives in registers)
        constexpr bool HOLD_Y = SK_ADJR_HOLD_Y && ND == 4 && !YSIDE;
        double trow_p[S], trow[S];
        if constexpr (TPEND) {
#pragma unroll
            for (int i = 0; i < S; ++i) async_begin(trow_p[i]);
            lds_read_f64_run<S>(trow_p, ec_rd + (unsigned)(((7 - (t & 7)) * S + 1) * 8));
        }

        if (chk_pair >= 0) {
            atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
            chk_pair = -1;
        }
        const int uo = NUp - 1 - u;    // original unit: node columns c0 = 2uo, c1 = 2uo + 1 (c2 = 2uo + 2 is last step's c0)

        // -- start of a (flipped) pair: boundaries, terminal column, upstream gradient, this lane's x points
        if (u == 0) {
            asm volatile("");
            valid = (unsigned)ps < (unsigned)ps_end ? 1 : 0;
            const unsigned xa = my_x + x_rd;
            double col[R + 2];
            double sv_rd = 1.0;
            const unsigned sv_at = my_sc + x_rd + (((sc_par0 ^ (unsigned)ps) & 1u) << 3);
            if constexpr (ND == 4) {     // one asm, one wait (lds_read_pair_start)
                d2_t xq[2 * RC], cq[R == 4 ? 3 : 2];
                lds_read_pair_start<2 * RC, (R == 4 ? 3 : 2)>(xq, cq, sv_rd, xa, my_col + x_rd, sv_at);
#pragma unroll
                for (int k = 0; k < RC; ++k) { xr[k][0] = xq[2 * k][0]; xr[k][1] = xq[2 * k][1]; xr[k][2] = xq[2 * k + 1][0]; xr[k][3] = xq[2 * k + 1][1]; }
#pragma unroll
                for (int i = 0; i < (R == 4 ? 3 : 2); ++i) { col[2 * i] = cq[i][0]; col[2 * i + 1] = cq[i][1]; }
            } else {
#pragma unroll
                for (int k = 0; k < RC; ++k) lds_read_xpt<ND>(xr[k], xa + k * 64u);
                const unsigned ca_ = my_col + x_rd;
                if constexpr (R == 4) {
                    d2_t c3[3];
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(c3[0]), "=&v"(c3[1]), "=&v"(c3[2]) : "v"(ca_) : "memory");
                    col[0] = c3[0][0]; col[1] = c3[0][1]; col[2] = c3[1][0]; col[3] = c3[1][1]; col[4] = c3[2][0]; col[5] = c3[2][1];
                } else {      // two fine rows per lane (dyadic 0, two coarse rows)
                    d2_t c2[2];
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(c2[0]), "=&v"(c2[1]) : "v"(ca_) : "memory");
                    col[0] = c2[0][0]; col[1] = c2[0][1]; col[2] = c2[1][0]; col[3] = c2[1][1];
                }
                if (prm.scale) sv_rd = lds_read_f64(sv_at);
            }
            if constexpr (YSIDE) {
                lds_read_xpt<ND>(xup, my_xup + x_rd);
                yp_prev = yp_cur;
                yp_cur = valid ? yp_base + (uint64_t)(unsigned)ps * (uint64_t)(unsigned)(2 * NUp * YWK) : nullptr;
            }
            // col[1 + m] = K[MMp - lam R - R + m][NN], m = 0..R: the lane's fine rows bottom to top; K[0][NN] = 1 is not stored
            cornerR = 1.0;
            cornerF = col[1 + R];
#pragma unroll
            for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = col[R - i]; }
            if (lam * R + R == MMp) leftF[R - 1] = 1.0;
            const double sv = prm.scale ? sv_rd : 1.0;
            if (sv != sv) valid = 0;      // NaN: a pair the rescue's screen took out of the sweep (sk_adj_fused_rescue.hip)
            sx = valid ? sv : 0.0;
            if constexpr (YSIDE) { if (!valid) yp_cur = nullptr; }
        }

        // -- y points of the unit's two node columns
        d2_t yv[ND];
        const unsigned ya = my_y + (unsigned)(yslab * RY_SLAB + ((u & 7) << 4));
        if constexpr (!TPEND && ND == 4 && S <= 4) {
            lds_read_ydims_trow<S>(yv, trow, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7), ec_rd + (unsigned)(((7 - (t & 7)) * S + 1) * 8));
        } else {
            lds_read_ydims<ND>(yv, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
            if constexpr (TPEND) lds_take<S>(trow, trow_p);
            else lds_read_f64_block<S>(trow, ec_rd + (unsigned)(((7 - (t & 7)) * S + 1) * 8));
        }

        // -- top rows of the two states
        double topR[S], topF[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            double tf = trow[S - 1 - i];
            if (i == S - 1 && u == NUp - 1) tf = 1.0;     // K[MM][0] = 1 is not stored
            if (FULLWAVE) {   // lane 0 keeps the `old` operand: the boundary (1 for the reverse state, the terminal row for K)
                topR[i] = dpp_shr1_one(botR[i]);
                topF[i] = dpp_shr1(botF[i], tf);
            } else {
                const double shR = dpp_shr1(botR[i], 1.0);
                const double shF = dpp_shr1(botF[i], 1.0);
                topR[i] = is_top ? 1.0 : shR;
                topF[i] = is_top ? tf : shF;
            }
        }
        // what the lane above evaluated / weighted one macro-step ago, for this unit's two columns (garbage for a top lane:
        // its first coarse row is padding and masked)
        double Gabv[2], wup0[2];
        Gabv[0] = dpp_shr1_zero(lastOwn[0]);
        Gabv[1] = dpp_shr1_zero(lastOwn[1]);
        wup0[0] = dpp_shr1_zero(lastW[0]);
        wup0[1] = dpp_shr1_zero(lastW[1]);
        if (is_top) { wup0[0] = 0.0; wup0[1] = 0.0; }     // nothing above the first lane of a group contributes

        // -- nodes of this lane's rows at the two columns
        double Gown[RC][2];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                double d2 = 0.0;
#pragma unroll
                for (int j = 0; j < ND; ++j) {
                    const double df = xr[k][j] - yv[j][q];
                    d2 = fma(df, df, d2);
                }
                Gown[k][q] = exp_nonpos(fma(-d2, prm.inv_sigma, d2 * 0.0), expc);
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

        // -- weights of the cells (original columns c0, c1), WITHOUT the pair's upstream gradient; zero outside the pair, in
        //    padding rows / columns and outside the group's pairs (SELECTED: leftovers may hold anything, NaN included)
        double wk[RC][2];
        {
            const bool live = valid != 0;
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                wk[k][0] = (live && row_ok[k] && c0_ok) ? acc[k][1] * sc : 0.0;
                wk[k][1] = (live && row_ok[k] && c1_ok) ? acc[k][0] * sc : 0.0;
            }
        }
        // -- contraction: node rows r_k = p_k + 1 at node columns c1 (this unit's second) and c2 (the previous unit's first).
        //    The y points are read from the ring a second time: holding them across the sweep costs 4 ND VGPRs
        // (HOLD_Y: the variants with registers to spare keep them instead -- one LDS round trip less per macro-step)
        if constexpr (YSIDE) {
            if constexpr (!YONLY) {      // (the second-argument sums alone need no y point)
                asm volatile("" ::: "memory");
                lds_read_ydims<ND>(yv, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
            }
#pragma unroll
            for (int i = 0; i < NCAR; ++i) { car[i][0] = dpp_shr1_zero(car[i][0]); car[i][1] = dpp_shr1_zero(car[i][1]); }
            if (!FULLWAVE && is_top) {
                asm volatile("");
#pragma unroll
                for (int i = 0; i < NCAR; ++i) { car[i][0] = 0.0; car[i][1] = 0.0; }
            }
        } else if constexpr (!HOLD_Y) {
            asm volatile("" ::: "memory");
            lds_read_ydims<ND>(yv, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
        }
        double cbA[PAIRED ? RC + 1 : 1], cbB[PAIRED ? RC + 1 : 1];     // PAIRED: V G of the node rows r_k (k < RC) and of the bottom lane's node row 0 at columns c1, c2
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const double u0 = k == 0 ? wup0[0] : wk[(k + RC - 1) % RC][0];     // cells of coarse row p_k + 1
            const double u1 = k == 0 ? wup0[1] : wk[(k + RC - 1) % RC][1];
            const double u2 = k == 0 ? wupP : wkP[(k + RC - 1) % RC];
            const double g1 = k == 0 ? Gabv[1] : Gown[(k + RC - 1) % RC][1];   // G[r_k][c1]
            const double g2 = k == 0 ? GabvP : GownP[(k + RC - 1) % RC];       // G[r_k][c2]
            const double V1 = ((wk[k][0] + u1) - wk[k][1]) - u0;
            const double V2 = ((wk[k][1] + u2) - wkP[k]) - u1;
            const double cb1 = V1 * g1, cb2 = V2 * g2;
            if constexpr (PAIRED) { cbA[k] = cb1; cbB[k] = cb2; }
            if constexpr (!YONLY && !PAIRED) {
                const double cv1 = cb1 * sx, cv2 = cb2 * sx_d;
                cs[k] += cv1 + cv2;
#pragma unroll
                for (int j = 0; j < ND; ++j) accd[k][j] = fma(cv1, yv[j][1], fma(cv2, yP[j], accd[k][j]));
            }
            if constexpr (YSIDE) {     // x of node row r_k: the lane above's last row (k = 0) or this lane's row k - 1
                car[0][0] += cb1; car[0][1] += cb2;
#pragma unroll
                for (int j = 0; j < ND; ++j) {
                    const double xj = k == 0 ? xup[j] : xr[(k + RC - 1) % RC][j];
                    car[1 + j][0] = fma(cb1, xj, car[1 + j][0]);
                    car[1 + j][1] = fma(cb2, xj, car[1 + j][1]);
                }
            }
        }
        {   // node row p_{RC-1} from its own cells only (V[0][c] = w[0][c] - w[0][c-1]): node row 0 on the bottom lane
            const double V1 = wk[RC - 1][1] - wk[RC - 1][0];
            const double V2 = wkP[RC - 1] - wk[RC - 1][1];
            const double cb1 = V1 * Gown[RC - 1][1], cb2 = V2 * GownP[RC - 1];
            if constexpr (PAIRED) { cbA[RC] = cb1; cbB[RC] = cb2; }
            if constexpr (!YONLY && !PAIRED) {
                const double cv1 = cb1 * sx, cv2 = cb2 * sx_d;
                if (is_bot) {
                    asm volatile("");
                    cs[RC] += cv1 + cv2;
#pragma unroll
                    for (int j = 0; j < ND; ++j) accd[RC][j] = fma(cv1, yv[j][1], fma(cv2, yP[j], accd[RC][j]));
                }
            }
            if constexpr (YSIDE) {
                // what this lane hands down (next macro-step's wave_shr): the sums over the node rows r_k so far
                if (is_bot) {   // the bottom lane completes them with node row 0 and stores the two columns of its pair
                    asm volatile("");
                    car[0][0] += cb1; car[0][1] += cb2;
#pragma unroll
                    for (int j = 0; j < ND; ++j) {
                        car[1 + j][0] = fma(cb1, xr[RC - 1][j], car[1 + j][0]);
                        car[1 + j][1] = fma(cb2, xr[RC - 1][j], car[1 + j][1]);
                    }
                    // node column c1 = 2uo + 1 of this pair; c2 = 2uo + 2, which at the pair's first step (uo = NUp - 1) is
                    // node column 0 of the pair BEFORE
                    double *d1 = yp_cur ? yp_cur + (2 * uo + 1) * YWK : nullptr;
                    double *d2p = u == 0 ? yp_prev : (yp_cur ? yp_cur + (2 * uo + 2) * YWK : nullptr);
                    if (d1) {
                        *reinterpret_cast<d2_t *>(d1) = d2_t{car[0][0], 0.0};
#pragma unroll
                        for (int j = 0; j < ND; j += 2) *reinterpret_cast<d2_t *>(d1 + 2 + j) = d2_t{car[1 + j][0], car[2 + j][0]};
                    }
                    if (d2p) {
                        *reinterpret_cast<d2_t *>(d2p) = d2_t{car[0][1], 0.0};
#pragma unroll
                        for (int j = 0; j < ND; j += 2) *reinterpret_cast<d2_t *>(d2p + 2 + j) = d2_t{car[1 + j][1], car[2 + j][1]};
                    }
                }
            }
        }
        // -- PAIRED: the first-argument sums, weighted by the upstream gradient of the pair a column belongs to (c2 at a pair's first step: the
        //    pair before -- which that step completes, stores and clears)
        if constexpr (PAIRED) {
            auto add = [&](double w1, double w2) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    const double cv1 = cbA[k] * w1, cv2 = cbB[k] * w2;
                    cs[k] += cv1 + cv2;
#pragma unroll
                    for (int j = 0; j < ND; ++j) accd[k][j] = fma(cv1, yv[j][1], fma(cv2, yP[j], accd[k][j]));
                }
                if (is_bot) {
                    asm volatile("");
                    const double cv1 = cbA[RC] * w1, cv2 = cbB[RC] * w2;
                    cs[RC] += cv1 + cv2;
#pragma unroll
                    for (int j = 0; j < ND; ++j) accd[RC][j] = fma(cv1, yv[j][1], fma(cv2, yP[j], accd[RC][j]));
                }
            };
            {
                if (u == 0) {
                    asm volatile("");
                    add(0.0, sx_d);       // the c2 terms complete the pair before: out it goes (slot = pair), the sums start over
                    if (valid_d) {
                        double *base = prm.Gpart + (pair0 + ps - 1) * (int64_t)(Mcp + 1) * OUTW;
#pragma unroll
                        for (int k = 0; k <= RC; ++k) {
                            if (k == RC && !is_bot) break;
                            const int row = k == RC ? 0 : Mcp - lam * RC - k;
                            double *dst = base + (int64_t)row * OUTW;
                            *reinterpret_cast<d2_t *>(dst) = d2_t{cs[k], 0.0};
#pragma unroll
                            for (int j = 0; j < ND; j += 2) *reinterpret_cast<d2_t *>(dst + 2 + j) = d2_t{accd[k][j], accd[k][j + 1]};
                        }
                    }
#pragma unroll
                    for (int k = 0; k <= RC; ++k) {
                        cs[k] = 0.0;
#pragma unroll
                        for (int j = 0; j < ND; ++j) accd[k][j] = 0.0;
                    }
                    add(sx, 0.0);
                } else {
                    add(sx, sx_d);
                }
            }
        }
        // -- histories for the next macro-step (and for the lane below, which reads lastOwn / lastW at its top)
        wupP = wup0[0];
        GabvP = Gabv[0];
#pragma unroll
        for (int k = 0; k < RC; ++k) { wkP[k] = wk[k][0]; GownP[k] = Gown[k][0]; }
        if constexpr (!YONLY) {
#pragma unroll
            for (int j = 0; j < ND; ++j) yP[j] = yv[j][0];
        }
        lastOwn[0] = Gown[RC - 1][0]; lastOwn[1] = Gown[RC - 1][1];
        lastW[0] = wk[RC - 1][0]; lastW[1] = wk[RC - 1][1];
        sx_d = sx;
        if constexpr (PAIRED) valid_d = valid;

        // -- self-check on the last flipped unit (see sk_wave_adj.hip)
        if (u == NUp - 1 && prm.err && valid) {
            double e = 0.0;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) e = fmax(e, fabs(leftF[rr] - 1.0));
            chk_val = e;
            chk_pair = pair0 + ps;
        }

        // -- close the step
        u += 1;
        if (u == NUp) { u = 0; ps += 1; }
        if (((t + 1) & 7) == lam7) {
            yslab = yslab + 1 == NSLAB ? 0 : yslab + 1;
            ypar ^= 1;
        }
        }
    }
    if (chk_pair >= 0)
        atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));

    // ---- the group's partial sums: Gpart[group][node row][OUTW], node row r_k = Mcp - lam RC - k; node row 0 from the bottom lane
    if constexpr (!YONLY && !PAIRED) {
        if (pair0 < prm.P) {
            double *base = prm.Gpart + gslot * (int64_t)(Mcp + 1) * OUTW;
#pragma unroll
            for (int k = 0; k <= RC; ++k) {
                if (k == RC && lam != L - 1) break;
                const int row = k == RC ? 0 : Mcp - lam * RC - k;
                double *dst = base + (int64_t)row * OUTW;
                *reinterpret_cast<d2_t *>(dst) = d2_t{cs[k], 0.0};
#pragma unroll
                for (int j = 0; j < ND; j += 2) *reinterpret_cast<d2_t *>(dst + 2 + j) = d2_t{accd[k][j], accd[k][j + 1]};
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DY, int RC, bool FULLWAVE, int ND, int YS, bool PAIRED = false>
int launch_adjr(const AdjRbfParams &prm, size_t lds_block, hipStream_t s) {
    auto kern = k_adj_fused_rbf<DY, RC, FULLWAVE, ND, YS, PAIRED>;
    if (lds_block > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
    SK_LAUNCH(kern, dim3(wave_group_blocks(prm.wg)), dim3(WAVE * prm.wg.wpb), lds_block, s, prm);
    return check_launch();
}

}  // namespace

// gpart viewed as [A][B / PPG][*rows_out][*outw_out] and summed over the chunk axis gives, per node row r < M of x_a,
// cs = that[a][r][0] and accd = that[a][r][2 .. 2 + D): dL/dx_a[r] = (-2 / sigma) (x_a[r] cs - accd).  gpart == nullptr: query.
// ypart (nullable; Gram, D <= 4): [A B][*ycols_out][6] per pair and node column c of y_b: S0 = that[..][0], S1 = that[..][2 .. 2 + D),
// without the upstream gradient: d k(x_a, y_b) / d y_b[c] = (-2 / sigma) (y_b[c] S0 - S1).
namespace {
int launch_adj_fused_rbf_rows(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                              double inv_sigma, const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *err,
                              double *ypart, int ys, int *ppg_out, int *rows_out, int *outw_out, int *ycols_out, int64_t *rows_per_launch,
                              int64_t force_nch, const FusedRescue *rescue, const double *scale_orig, void *rescue_ws, size_t rescue_ws_bytes,
                              hipStream_t s) {
    // ys: 0 first-argument sums, 1 both, 2 the second-argument sums only (the kernel's YS)
    const int DY = g.dyadic;
    if (DY < 0 || DY > 2 || B < 0 || D < 1 || D > RFD || g.P != (B > 0 ? A * B : A)) return SK_ERR_UNSUPPORTED;
    const Strip st = strip_geom(rbf_edge_geom(g), 8);   // the layout of the edges; the sweep uses the same lanes and units
    if (!st.ok || st.nb != 1) return SK_ERR_UNSUPPORTED;
    // dyadic 0: the strip kernels give a lane four coarse rows, which this kernel's accumulators do not fit (95-128 VGPRs spilled:
    // measured no faster than the streaming route); it sweeps TWO rows per lane with twice the lanes -- the same padded rows, so the
    // edge layout is the strip kernels' own (pairs of up to 128 points; the default stencil: what the forward that keeps these edges
    // is built for; dim 5..8: 18 VGPRs spilled, like the dyadic-2 variants of that width)
    if (DY == 0 && (g.naive || st.logL > 5)) return SK_ERR_UNSUPPORTED;
    // dyadic 1 with 8 staged dims: two coarse rows of 8 dims per lane spill 84-113 VGPRs (and lose to the multi-band kernel); ONE row
    // per lane on twice the lanes fits (238 VGPRs) -- pairs of up to 64 points, the strip layout as it is.  (The two-row form and the
    // dyadic-2 variant of that width spilled 84-280 bytes and no route of sk_route_query reached them: removed in round 5,
    // profiles/r05_variants.txt; such calls take the multi-band adjoint or stream.)
    if (DY == 2 && D > 4) return SK_ERR_UNSUPPORTED;
    // ... and so do the second-argument sums at dyadic 1: with two rows per lane that variant spilled 168-176 bytes and ran 2.5x
    // slower per pair than the plain one (the triangle it serves lost to all pairs: 42 against 27 ms on 1024 paths of 64 points,
    // profiles/r05_yside_ab.txt); one row per lane fits -- pairs of up to 64 points; longer paths take all pairs (sigkernel.py)
    // (the second-argument sums ALONE fit with two rows per lane: 174-181 VGPRs at one row)
    const bool half_rows = DY == 0 || (DY == 1 && (D > 4 || ys == 1));
    if (half_rows && st.logL > 5) return SK_ERR_UNSUPPORTED;
    const int RC = half_rows ? st.RC / 2 : st.RC, NUp = st.NUp, logL = half_rows ? st.logL + 1 : st.logL, L = 1 << logL, G = WAVE / L;
    if (g.Mc + 1 > L * RC) return SK_ERR_UNSUPPORTED;          // the node rows must fit the lanes (the last lane-row is padding)
    if (g.Nc > 2 * NUp - 1) return SK_ERR_UNSUPPORTED;         // node column 2 NUp must be padding
    if (Ncp < NUp * 2 || (Ncp & 1) || Mrows < L * RC + 1) return SK_ERR_UNSUPPORTED;   // (+ 1: the node row above the first lane's)
    const int ND = D <= 4 ? 4 : 8;
    if (ys == 1 && ND != 4) return SK_ERR_UNSUPPORTED;   // (dims 5..8: the second-argument sums INSTEAD of the first-argument ones)
    if (ys && B <= 0) return SK_ERR_UNSUPPORTED;
    const int JMAX = (L + NUp - 1) / NUp;
    const int S = 2 << DY;
    const int xslab = DY == 0 ? XSlab<2, 2>::BYTES : DY == 1 ? (half_rows ? XSlab<1, 2>::BYTES : XSlab<2, 4>::BYTES) : XSlab<1, 4>::BYTES;
    const size_t lds_bytes = (size_t)G * (((L >> 3) + 2) * RY_SLAB + RX_SLOTS * JMAX * xslab) + (size_t)2 * G * (4 * S + 1) * 16;
    if (lds_bytes > 160 * 1024) return SK_ERR_UNSUPPORTED;

    const int n_cu = device_cu_count();
    const int wpc = knobs().adjr_wpc > 0 ? knobs().adjr_wpc : 8;
    const int64_t max_groups = (int64_t)n_cu * wpc * G;
    // pairs per lane group: see pick_chunk (paired batches: every pair has its own x, one pair per lane group)
    int64_t PPG = pick_chunk(A, B, max_groups);
    // paired batches of more pairs than resident lane groups (dims <= 4): several consecutive pairs per lane group (PAIRED)
    int64_t ppp = 1;
    if (B <= 0 && ys == 0 && ND == 4) {
        ppp = g.P / max_groups;
        ppp = ppp < 1 ? 1 : (ppp > 64 ? 64 : ppp);
        PPG = ppp;
    }
    if (force_nch > 0) PPG = (B + force_nch - 1) / force_nch;
    else if (rows_per_launch) {
        // Shares by wave age rank (ChunkSplit) need a launch that fills the chip exactly, with the chunks of an a a multiple of
        // the ranks: when the equal split does not give that, the caller sweeps the rows in several such launches.
        *rows_per_launch = 0;
        const int wpb = wave_group(lds_bytes, max_groups / G, knobs().adjr_wpb).wpb;
        const int64_t gpr = (int64_t)n_cu * wpb * G;
        const int64_t nr = gpr > 0 && max_groups % gpr == 0 ? max_groups / gpr : 0;
        const int64_t nch = chunks_of(B, PPG);
        if (B > 0 && nr >= 2 && !(A * nch == max_groups && nch % nr == 0 && B % nch == 0)) {
            // several exactly-filling launches of max_groups / m rows with m chunks per a, shares by wave age rank (~0.9 of the equal
            // split's time each), against the one launch above: rounds x chunk length; the best m, the smallest among equals
            const int64_t single_rounds = (A * nch + max_groups - 1) / max_groups;
            double best = (double)single_rounds * (double)PPG;
            for (int64_t m = nr; m <= B && m <= max_groups; m += nr)
                if (m >= nch && B % m == 0 && max_groups % m == 0 && B / m >= 4 * nr) {
                    const int64_t rpl = max_groups / m, launches = (A + rpl - 1) / rpl;
                    const double t = 0.9 * (double)launches * (double)(B / m);
                    if (A >= rpl && t < best * 0.97) { best = t; *rows_per_launch = rpl; PPG = B / m; }
                }
        }
    }
    if (PPG > 0x3fffffff / NUp || g.P >= 0x7ff00000LL) return SK_ERR_UNSUPPORTED;   // (pair indices are divided in 32 bits inside the kernel)
    const int64_t groups = B > 0 ? A * chunks_of(B, PPG) : (g.P + ppp - 1) / ppp;
    const int OUTW = ND + 2;
    if (ppg_out) *ppg_out = (int)PPG;
    if (rows_out) *rows_out = L * RC + 1;
    if (outw_out) *outw_out = OUTW;
    if (ycols_out) *ycols_out = 2 * NUp;
    if (!gpart && !ypart) return SK_OK;
    if (gpart && gpart_doubles < (size_t)(B > 0 ? groups : g.P) * (L * RC + 1) * OUTW) return SK_ERR_WORKSPACE;   // (paired: a slot per pair)
    const int64_t waves = (groups + G - 1) / G;

    AdjRbfParams prm;
    prm.Xr = Xr; prm.Yt = Yt; prm.edges = edges; prm.scale = scale; prm.Gpart = gpart; prm.err = err; prm.Ypart = ypart;
    prm.P = g.P; prm.B = B; prm.Mrows = Mrows; prm.Ncp = Ncp; prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = NUp; prm.logL = logL;
    prm.naive = g.naive;
    prm.PPG = (int)PPG;
    prm.inv_sigma = inv_sigma;
    prm.n_steps = (int)(PPG * NUp + (L - 1)) + 1;    // + 1: node column 0 of the last pair completes one step later
    prm.wg = wave_group(lds_bytes, waves, knobs().adjr_wpb);
    prm.cs = chunk_split(A, B, PPG, max_groups, G, prm.wg.wpb, n_cu, knobs().adjr_rank_w);
    prm.cs.ppp = (int)ppp;
    const size_t lds_block = wave_group_lds(prm.wg);
    const bool full = logL == 6;
    int rc;
    const bool yonly = ys == 2;
    if (ppp > 1) {
        if (DY == 0) rc = full ? launch_adjr<0, 2, true, 4, 0, true>(prm, lds_block, s) : launch_adjr<0, 2, false, 4, 0, true>(prm, lds_block, s);
        else if (DY == 1) rc = full ? launch_adjr<1, 2, true, 4, 0, true>(prm, lds_block, s) : launch_adjr<1, 2, false, 4, 0, true>(prm, lds_block, s);
        else rc = full ? launch_adjr<2, 1, true, 4, 0, true>(prm, lds_block, s) : launch_adjr<2, 1, false, 4, 0, true>(prm, lds_block, s);
    } else if (DY == 0) {
        if (yonly && ND == 8) rc = full ? launch_adjr<0, 2, true, 8, 2>(prm, lds_block, s) : launch_adjr<0, 2, false, 8, 2>(prm, lds_block, s);
        else if (yonly) rc = full ? launch_adjr<0, 2, true, 4, 2>(prm, lds_block, s) : launch_adjr<0, 2, false, 4, 2>(prm, lds_block, s);
        else if (ypart) rc = full ? launch_adjr<0, 2, true, 4, 1>(prm, lds_block, s) : launch_adjr<0, 2, false, 4, 1>(prm, lds_block, s);
        else if (ND == 8) rc = full ? launch_adjr<0, 2, true, 8, 0>(prm, lds_block, s) : launch_adjr<0, 2, false, 8, 0>(prm, lds_block, s);
        else rc = full ? launch_adjr<0, 2, true, 4, 0>(prm, lds_block, s) : launch_adjr<0, 2, false, 4, 0>(prm, lds_block, s);
    } else if (ypart) {
        if (DY == 1 && ND == 8) rc = full ? launch_adjr<1, 1, true, 8, 2>(prm, lds_block, s) : launch_adjr<1, 1, false, 8, 2>(prm, lds_block, s);
        else if (DY == 1 && yonly) rc = full ? launch_adjr<1, 2, true, 4, 2>(prm, lds_block, s) : launch_adjr<1, 2, false, 4, 2>(prm, lds_block, s);
        else if (DY == 1) rc = full ? launch_adjr<1, 1, true, 4, 1>(prm, lds_block, s) : launch_adjr<1, 1, false, 4, 1>(prm, lds_block, s);
        else if (yonly) rc = full ? launch_adjr<2, 1, true, 4, 2>(prm, lds_block, s) : launch_adjr<2, 1, false, 4, 2>(prm, lds_block, s);
        else rc = full ? launch_adjr<2, 1, true, 4, 1>(prm, lds_block, s) : launch_adjr<2, 1, false, 4, 1>(prm, lds_block, s);
    } else if (DY == 1) {
        if (ND == 4) rc = full ? launch_adjr<1, 2, true, 4, 0>(prm, lds_block, s) : launch_adjr<1, 2, false, 4, 0>(prm, lds_block, s);
        else rc = full ? launch_adjr<1, 1, true, 8, 0>(prm, lds_block, s) : launch_adjr<1, 1, false, 8, 0>(prm, lds_block, s);
    } else {
        rc = full ? launch_adjr<2, 1, true, 4, 0>(prm, lds_block, s) : launch_adjr<2, 1, false, 4, 0>(prm, lds_block, s);
    }
    if (rc != SK_OK || !rescue || !rescue_ws) return rc;
    ChunkSplit rcs = prm.cs;      // (paired: every pair has its slot whatever the lane groups swept -- the rescue walks pairs)
    rcs.ppp = 1;
    if (B <= 0) rcs.size[0] = 1;
    return launch_fused_rescue(1, Xr, Yt, scale_orig, err, rescue->tol, gpart, ypart, A, B, Mrows, Ncp, D, g, L * RC + 1, OUTW, 2 * NUp, inv_sigma,
                               rcs, B > 0 ? groups : g.P, rescue_ws, rescue_ws_bytes, s, 8, nullptr, 0, rescue->kfinal);
}
}  // namespace

int launch_adj_fused_rbf(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                         double inv_sigma, const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *err,
                         double *ypart, size_t ypart_doubles, int want_yside, int *ppg_out, int *rows_out, int *outw_out, int *ycols_out,
                         const FusedRescue *rescue, hipStream_t s) {
    int ppg = 0, rows = 0, outw = 0, ycols = 0;
    int64_t per_launch = 0;
    const bool yside = want_yside || ypart;
    // the form of the sweep: both sets of sums (dims <= 4) unless the caller hands over no gpart; a pure size query (neither array)
    // answers for the form that exists -- the sizes it returns for the second-argument sums (*ycols_out) are the same in both
    int ys = !yside ? 0 : ((D > 4 || (ypart && !gpart)) ? 2 : 1);
    int rc = launch_adj_fused_rbf_rows(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, edges, scale, nullptr, 0, err, nullptr, ys, &ppg, &rows, &outw,
                                       yside ? &ycols : nullptr, &per_launch, 0, nullptr, nullptr, nullptr, 0, s);
    if (rc == SK_ERR_UNSUPPORTED && ys == 1 && !gpart && !ypart) {
        ys = 2;
        rc = launch_adj_fused_rbf_rows(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, edges, scale, nullptr, 0, err, nullptr, ys, &ppg, &rows, &outw,
                                       &ycols, &per_launch, 0, nullptr, nullptr, nullptr, 0, s);
    }
    if (rc != SK_OK) return rc;
    if (ppg_out) *ppg_out = ppg;
    if (rows_out) *rows_out = rows;
    if (outw_out) *outw_out = outw;
    if (ycols_out) *ycols_out = ycols;
    if (!gpart && !ypart) return SK_OK;      // (gpart NULL with ypart: the second-argument sums only)
    const int yw = yw_of(D <= 4 ? 4 : 8);
    if (ypart && ypart_doubles < (size_t)g.P * ycols * yw) return SK_ERR_WORKSPACE;
    // device-side rescue (sk_adj_fused_rescue.hip): the workspace starts with the swept upstream gradient (screened pairs NaN)
    const double *sweep_scale = scale;
    void *rws = nullptr;
    size_t rws_bytes = 0;
    if (rescue && rescue->ws) {
        const size_t head = sizeof(double) * (size_t)((g.P + 1) / 2 * 2);
        if (rescue->ws_bytes <= head) return SK_ERR_WORKSPACE;
        rws = (char *)rescue->ws + head;
        rws_bytes = rescue->ws_bytes - head;
        if (rescue->kfinal) {
            rc = launch_fused_screen(rescue->kfinal, scale, g.P, rescue->screen, (double *)rescue->ws, err, s);
            if (rc != SK_OK) return rc;
            sweep_scale = (const double *)rescue->ws;
        }
    }
    if (per_launch <= 0 || B <= 0)
        return launch_adj_fused_rbf_rows(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, edges, sweep_scale, gpart, gpart_doubles, err, ypart, ys, nullptr,
                                         nullptr, nullptr, nullptr, nullptr, B > 0 ? chunks_of(B, ppg) : 0, rescue, scale, rws, rws_bytes, s);
    // several launches of per_launch rows each, all with the same chunks per a (so that gpart keeps one layout)
    const int64_t nch = chunks_of(B, ppg);
    const int64_t slot = (int64_t)rows * outw;
    if (gpart && gpart_doubles < (size_t)(A * nch * slot)) return SK_ERR_WORKSPACE;
    const Strip st = strip_geom(rbf_edge_geom(g), 8);
    const int64_t Epair = (int64_t)st.NUp * (2 << g.dyadic) + (int64_t)(1 << st.logL) * st.RC * (1 << g.dyadic);   // edge doubles per pair
    for (int64_t a0 = 0; a0 < A; a0 += per_launch) {
        const int64_t An = A - a0 < per_launch ? A - a0 : per_launch;
        Geom gs = g;
        gs.P = An * B;
        rc = launch_adj_fused_rbf_rows(Xr + a0 * Mrows * RFD, Yt, An, B, Mrows, Ncp, D, gs, inv_sigma, edges + a0 * B * Epair,
                                       sweep_scale ? sweep_scale + a0 * B : nullptr, gpart ? gpart + a0 * nch * slot : nullptr, (size_t)(An * nch * slot),
                                       err ? err + a0 * B : nullptr, ypart ? ypart + a0 * B * (int64_t)ycols * yw : nullptr, ys, nullptr, nullptr,
                                       nullptr, nullptr, nullptr, nch, rescue, scale ? scale + a0 * B : nullptr, rws, rws_bytes, s);
        if (rc != SK_OK) return rc;
    }
    return SK_OK;
}

}  // namespace sk
