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
// Scope: fp64, dyadic 1..2, path dim <= 8 (ND = 4 variants for dim <= 4), one band per pair with M <= L RC, N - 1 <= 2 NUp - 1.
#include "sk_wave_common.h"

namespace sk {
namespace {

constexpr int RFD = 8;                 // dims carried by the staged arrays
constexpr int RY_SLAB = RFD * 128;
constexpr int RX_SLOTS = 2;

struct AdjRbfParams {
    const double *Xr;      // [A][Mrows][8]  x_p (points), zero rows / dims beyond M / D
    const double *Yt;      // [B][8][Ncp]    y_q, dimension-major, zero-padded
    const double *edges;   // [P][NNp + MMp] strip layout (strip_geom)
    const double *scale;   // [P] upstream gradient per pair, nullable
    double *Gpart;         // [P / PPG][L*RC + 1][OUTW]  per node row: cs, 0, accd[0..ND)
    double *err;           // [P] zero-initialised: worst |Kf - 1| on the recomputed boundary
    int64_t P, B;
    int Mrows, Ncp, Mc, Nc, NUp, logL, PPG, n_steps;
    ChunkSplit cs;         // chunk sizes by wave age rank; PPG / n_steps are the equal split's
    double inv_sigma;
    WaveGroup wg;
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

template <int DY, int RC, bool FULLWAVE, int ND>
__global__ __launch_bounds__(4 * WAVE) __attribute__((amdgpu_waves_per_eu(2))) void k_adj_fused_rbf(const AdjRbfParams prm) {
    constexpr int CW = 2;
    constexpr int R = RC << DY, S = CW << DY, r = 1 << DY;
    constexpr int XSLAB = RC * 512;
    constexpr int OUTW = ND + 2;
    constexpr int ECG = (4 * S + 1) * 16;    // terminal-row chunk of one lane group (sk_wave_adj.hip)
    constexpr int NPC = 4 * S + 1;
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
    const int MM = prm.Mc << DY, MMp = Mcp << DY, NNp = (NUp * CW) << DY;
    const int NSLAB = (L >> 3) + 2;
    const unsigned y_bytes = (unsigned)(NSLAB * RY_SLAB);
    const unsigned x_base0 = (unsigned)G * y_bytes;
    const double sc = 1.0 / (double)(1 << (2 * DY));
    const double c_half = 0.5 * sc, c_12 = sc * sc / 12.0;

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
    PPG = __builtin_amdgcn_readfirstlane(PPG);   // (one rank per wave)
    const int n_steps = PPG * NUp + (L - 1) + 1;   // + 1: node column 0 of the last pair completes one step later
    auto group_first = [&](int g) -> int64_t { return readlane64(pair0, g << prm.logL); };
    const bool is_top = lam == 0;
    const unsigned my_y = lds0 + (unsigned)grp * y_bytes;
    const int JMAX = (L + NUp - 1) / NUp;
    const unsigned my_x = lds0 + x_base0 + (unsigned)((grp * RX_SLOTS * JMAX) * XSLAB + (lam / NUp) * XSLAB) +
                          (unsigned)((lam & 7) * RC * 64);
    const unsigned ec_base = lds0 + x_base0 + (unsigned)(G * RX_SLOTS * JMAX * XSLAB);   // terminal-row chunks behind the rings
    const unsigned ec_slot = (unsigned)(G * ECG);
    bool row_ok[RC];   // this lane's coarse rows that exist (p_k < Mc)
#pragma unroll
    for (int k = 0; k < RC; ++k) row_ok[k] = Mcp - 1 - (lam * RC + k) < prm.Mc;

    // ---- producers: the rings of sk_wave_fused.hip (points), filled in FLIPPED order --------------------------------------
    const bool small = prm.P <= 0x7fffffffLL && prm.B <= 0x7fffffffLL;
    auto split_b = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return small ? (int64_t)((uint32_t)p % (uint32_t)prm.B) : p % prm.B;
    };
    auto split_a = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return small ? (int64_t)((uint32_t)p / (uint32_t)prm.B) : p / prm.B;
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
    // x slabs: LDS position i = (lam & 7) * RC + k holds the x POINT of node row p = Mcp - 1 - (lamj*RC + i)
    int x_q0 = 0, x_lam0 = 0, x_slot = 0;
    auto issue_x = [&]() {
        for (int j = 0; j < JMAX; ++j) {
            const int lamj = x_lam0 + j * NUp, pi = x_q0 - j;
            if (lamj >= L) break;
            for (int g = 0; g < G; ++g) {
                int64_t p = group_first(g) + pi;
                if (pi < 0 || pi >= PPG || p >= prm.P) p = 0;
                const int64_t a = split_a(p);
                char *dst = lds + x_base0 + ((g * RX_SLOTS + x_slot) * JMAX + j) * XSLAB;
#pragma unroll
                for (int c = 0; c < (XSLAB + 1023) / 1024; ++c)
                    if (c * 1024 + lane * 16 < XSLAB) {
                        const int i = c * 16 + (lane >> 2);
                        const int row = Mcp - 1 - (lamj * RC + i);
                        const double *src = prm.Xr + (a * prm.Mrows + row) * RFD + (lane & 3) * 2;
                        __builtin_amdgcn_global_load_lds(src, (lds_void *)(dst + c * 1024), 16, 0, 0);
                    }
            }
        }
        x_slot = x_slot + 1 == RX_SLOTS ? 0 : x_slot + 1;
        x_lam0 += 8;
        if (x_lam0 == NUp) { x_lam0 = 0; x_q0 += 1; }
    };
    // terminal ROW of the pair the top lanes are in, one window ahead (sk_wave_adj.hip: issue_edge_chunk)
    const int E = NNp + MMp;
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

    // ---- terminal COLUMN and the upstream gradient of the coming pair, one macro-step ahead (sk_wave_adj.hip) ------------
    auto prefetch_edges = [&](int nu, int nps, double (&pcol)[R + 1], double &pscale) {
        if (nu == 0) {
            int64_t pr = pair0 + nps;
            pr = pr < 0 ? 0 : (pr >= prm.P ? prm.P - 1 : pr);
            const double *e = prm.edges + pr * E;
            const int i0 = lam * RC * r;
            const double *q = e + (NNp - 1);
#pragma unroll
            for (int i = 0; i < R; ++i) load_async(pcol[i], q + min(MM, MMp - (i0 + i)));
            load_async(pcol[R], q + max(min(MM, MMp - (i0 + R)), 1));
            if (prm.scale) load_async(pscale, prm.scale + pr);
        }
    };
    auto fix_edges = [&](int nu, double (&pcol)[R + 1]) {
        if (nu == 0 && lam * RC * r + R == MMp) pcol[R] = 1.0;
    };

    double xr[RC][ND];
#pragma unroll
    for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int j = 0; j < ND; ++j) xr[k][j] = 0.0;
    // accumulators per node row r_k = p_k + 1 (k < RC) and, [RC], node row p_{RC-1} (node row 0 on the bottom lane)
    double cs[RC + 1], accd[RC + 1][ND];
#pragma unroll
    for (int k = 0; k <= RC; ++k) {
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
    double s_pair = 0.0;    // upstream gradient of the pair being swept (0 outside the group's pairs)
    double ncol[R + 1], nscale = 0.0;
#pragma unroll
    for (int i = 0; i <= R; ++i) ncol[i] = 1.0;

    {   // lanes ahead of their first pair read slabs no DMA has written yet: make those finite
        const int total = (int)(G * y_bytes + G * RX_SLOTS * JMAX * XSLAB + 2 * G * ECG);
        const d2_t z = {0.0, 0.0};
        for (int o = lane * 16; o < total; o += WAVE * 16) lds_write_b128(lds0 + (unsigned)o, z);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    issue_y();
    issue_x();
    issue_edge_chunk();
    {
        double pcol[R + 1], pscale[1], tsc[1];
#pragma unroll
        for (int i = 0; i <= R; ++i) async_begin(pcol[i]);
        async_begin(pscale[0]);
        prefetch_edges(u, ps, pcol, pscale[0]);
        async_wait<0>(ncol, pcol);
        async_wait<0>(tsc, pscale);
        fix_edges(u, ncol);
        nscale = (u == 0 && ps >= 0 && ps < PPG) ? (prm.scale ? tsc[0] : 1.0) : 0.0;
    }
    issue_y();
    issue_x();

    for (int t = 0; t < n_steps; ++t) {
        // the top lane's terminal-row values of this macro-step (no wait: complete at the y read's lgkmcnt(0) below)
        double trow_p[S], trow[S];
#pragma unroll
        for (int i = 0; i < S; ++i) async_begin(trow_p[i]);
        lds_read_f64_run<S>(trow_p, ec_base + (unsigned)(((t >> 3) & 1) * ec_slot + grp * ECG + ((7 - (t & 7)) * S + 1) * 8));

        if (chk_pair >= 0) {
            atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
            chk_pair = -1;
        }
        int nu = u + 1, nps = ps;
        if (nu == NUp) { nu = 0; nps += 1; }
        const int uo = NUp - 1 - u;    // original unit: node columns c0 = 2uo, c1 = 2uo + 1 (c2 = 2uo + 2 is last step's c0)

        // -- start of a (flipped) pair: boundaries, upstream gradient, this lane's x points
        if (u == 0) {
            cornerR = 1.0;
            cornerF = ncol[0];
#pragma unroll
            for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = ncol[i + 1]; }
            s_pair = nscale;
            const unsigned xa = my_x + (unsigned)(((t >> 3) % RX_SLOTS) * JMAX * XSLAB);
#pragma unroll
            for (int k = 0; k < RC; ++k) lds_read_xpt<ND>(xr[k], xa + k * 64u);
        }

        // -- y points of the unit's two node columns
        d2_t yv[ND];
        {
            const unsigned ya = my_y + (unsigned)(yslab * RY_SLAB + ((u & 7) << 4));
            lds_read_ydims<ND>(yv, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
        }
        lds_take<S>(trow, trow_p);
        if ((t & 7) == 0) issue_edge_chunk();

        // -- top rows of the two states
        double topR[S], topF[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            double tf = trow[S - 1 - i];
            if (i == S - 1 && u == NUp - 1) tf = 1.0;     // K[MM][0] = 1 is not stored
            if (FULLWAVE) {   // lane 0 keeps the `old` operand: the boundary (1 for the reverse state, the terminal row for K)
                topR[i] = dpp_shr1(botR[i], 1.0);
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
        Gabv[0] = dpp_shr1(lastOwn[0], 0.0);
        Gabv[1] = dpp_shr1(lastOwn[1], 0.0);
        wup0[0] = dpp_shr1(lastW[0], 0.0);
        wup0[1] = dpp_shr1(lastW[1], 0.0);
        if (is_top) { wup0[0] = 0.0; wup0[1] = 0.0; }     // nothing above the first lane of a group contributes

        // -- next step's column edges (asynchronous)
        double pcol[R + 1], pscale[1];
#pragma unroll
        for (int i = 0; i <= R; ++i) async_begin(pcol[i]);
        async_begin(pscale[0]);
        prefetch_edges(nu, nps, pcol, pscale[0]);

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

        // -- weights of the cells (original columns c0, c1), scaled by the pair's upstream gradient; zero outside the pair,
        //    in padding rows / columns and outside the group's pairs (SELECTED: leftovers may hold anything, NaN included)
        double wk[RC][2];
        {
            const bool live = s_pair != 0.0;
            const double wsc = sc * s_pair;
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                wk[k][0] = (live && row_ok[k] && c0_ok) ? acc[k][1] * wsc : 0.0;
                wk[k][1] = (live && row_ok[k] && c1_ok) ? acc[k][0] * wsc : 0.0;
            }
        }
        // -- contraction: node rows r_k = p_k + 1 at node columns c1 (this unit's second) and c2 (the previous unit's first).
        //    The y points are read from the ring a second time: holding them across the sweep costs 4 ND VGPRs
        {
            const unsigned ya = my_y + (unsigned)(yslab * RY_SLAB + ((u & 7) << 4));
            asm volatile("" ::: "memory");
            lds_read_ydims<ND>(yv, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
        }
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const double u0 = k == 0 ? wup0[0] : wk[(k + RC - 1) % RC][0];     // cells of coarse row p_k + 1
            const double u1 = k == 0 ? wup0[1] : wk[(k + RC - 1) % RC][1];
            const double u2 = k == 0 ? wupP : wkP[(k + RC - 1) % RC];
            const double g1 = k == 0 ? Gabv[1] : Gown[(k + RC - 1) % RC][1];   // G[r_k][c1]
            const double g2 = k == 0 ? GabvP : GownP[(k + RC - 1) % RC];       // G[r_k][c2]
            const double V1 = ((wk[k][0] + u1) - wk[k][1]) - u0;
            const double V2 = ((wk[k][1] + u2) - wkP[k]) - u1;
            const double cv1 = V1 * g1, cv2 = V2 * g2;
            cs[k] += cv1 + cv2;
#pragma unroll
            for (int j = 0; j < ND; ++j) accd[k][j] = fma(cv1, yv[j][1], fma(cv2, yP[j], accd[k][j]));
        }
        {   // node row p_{RC-1} from its own cells only (V[0][c] = w[0][c] - w[0][c-1]): node row 0 on the bottom lane
            const double V1 = wk[RC - 1][1] - wk[RC - 1][0];
            const double V2 = wkP[RC - 1] - wk[RC - 1][1];
            const double cv1 = V1 * Gown[RC - 1][1], cv2 = V2 * GownP[RC - 1];
            cs[RC] += cv1 + cv2;
#pragma unroll
            for (int j = 0; j < ND; ++j) accd[RC][j] = fma(cv1, yv[j][1], fma(cv2, yP[j], accd[RC][j]));
        }
        // -- histories for the next macro-step (and for the lane below, which reads lastOwn / lastW at its top)
        wupP = wup0[0];
        GabvP = Gabv[0];
#pragma unroll
        for (int k = 0; k < RC; ++k) { wkP[k] = wk[k][0]; GownP[k] = Gown[k][0]; }
#pragma unroll
        for (int j = 0; j < ND; ++j) yP[j] = yv[j][0];
        lastOwn[0] = Gown[RC - 1][0]; lastOwn[1] = Gown[RC - 1][1];
        lastW[0] = wk[RC - 1][0]; lastW[1] = wk[RC - 1][1];

        // -- self-check on the last flipped unit (see sk_wave_adj.hip)
        if (u == NUp - 1 && prm.err && ps >= 0 && ps < PPG && pair0 + ps < prm.P) {
            double e = 0.0;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) e = fmax(e, fabs(leftF[rr] - 1.0));
            chk_val = e;
            chk_pair = pair0 + ps;
        }

        // -- close the step
        {
            double tsc[1];
            async_wait<0>(ncol, pcol);
            async_wait<0>(tsc, pscale);
            fix_edges(nu, ncol);
            if (nu == 0) nscale = (nps >= 0 && nps < PPG) ? (prm.scale ? tsc[0] : 1.0) : 0.0;
        }
        u = nu;
        ps = nps;
        if (((t + 1) & 7) == lam7) {
            yslab = yslab + 1 == NSLAB ? 0 : yslab + 1;
            ypar ^= 1;
        }
        if (((t + 1) & 7) == 0) {
            issue_y();
            issue_x();
        }
    }
    if (chk_pair >= 0)
        atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));

    // ---- the group's partial sums: Gpart[group][node row][OUTW], node row r_k = Mcp - lam RC - k; node row 0 from the bottom lane
    {
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

template <int DY, int RC, bool FULLWAVE, int ND>
int launch_adjr(const AdjRbfParams &prm, size_t lds_block, hipStream_t s) {
    auto kern = k_adj_fused_rbf<DY, RC, FULLWAVE, ND>;
    if (lds_block > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
    hipLaunchKernelGGL(kern, dim3(wave_group_blocks(prm.wg)), dim3(WAVE * prm.wg.wpb), lds_block, s, prm);
    return check_launch();
}

}  // namespace

// gpart viewed as [A][B / PPG][*rows_out][*outw_out] and summed over the chunk axis gives, per node row r < M of x_a,
// cs = that[a][r][0] and accd = that[a][r][2 .. 2 + D): dL/dx_a[r] = (-2 / sigma) (x_a[r] cs - accd).  gpart == nullptr: query.
namespace {
int launch_adj_fused_rbf_rows(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                              double inv_sigma, const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *err,
                              int *ppg_out, int *rows_out, int *outw_out, int64_t *rows_per_launch, int64_t force_nch, hipStream_t s) {
    const int DY = g.dyadic;
    if (DY < 1 || DY > 2 || B < 0 || g.naive || D < 1 || D > RFD || g.P != (B > 0 ? A * B : A)) return SK_ERR_UNSUPPORTED;
    const Strip st = strip_geom(g, 8);   // the layout of the edges; the sweep uses the same lanes and units
    if (!st.ok || st.nb != 1) return SK_ERR_UNSUPPORTED;
    const int RC = st.RC, NUp = st.NUp, logL = st.logL, L = 1 << logL, G = WAVE / L;
    if (g.Mc + 1 > L * RC) return SK_ERR_UNSUPPORTED;          // the node rows must fit the lanes (the last lane-row is padding)
    if (g.Nc > 2 * NUp - 1) return SK_ERR_UNSUPPORTED;         // node column 2 NUp must be padding
    if (Ncp < NUp * 2 || (Ncp & 1) || Mrows < L * RC) return SK_ERR_UNSUPPORTED;
    const int ND = D <= 4 ? 4 : 8;
    if (ND == 8 && DY == 1 && !env_int("SK_ADJR_ALL", 0)) return SK_ERR_UNSUPPORTED;   // two coarse rows of 8 dims per lane: 77 VGPRs spilled
    const int JMAX = (L + NUp - 1) / NUp;
    const int S = 2 << DY;
    const size_t lds_bytes = (size_t)G * (((L >> 3) + 2) * RY_SLAB + RX_SLOTS * JMAX * RC * 512) + (size_t)2 * G * (4 * S + 1) * 16;
    if (lds_bytes > 160 * 1024) return SK_ERR_UNSUPPORTED;

    const int wpc = env_int("SK_ADJR_WPC", 8);
    const int64_t max_groups = 256LL * wpc * G;
    int64_t PPG = B > 0 ? B : 1;
    for (int64_t d = 1; d <= B; ++d)
        if (B % d == 0 && A * (B / d) <= max_groups) { PPG = d; break; }
    if (force_nch > 0) PPG = B / force_nch;
    else if (rows_per_launch) {
        // Shares by wave age rank (ChunkSplit) need a launch that fills the chip exactly, with the chunks of an a a multiple of
        // the ranks: when the equal split does not give that, the caller sweeps the rows in several such launches.
        *rows_per_launch = 0;
        const int wpb = wave_group(lds_bytes, max_groups / G, "SK_ADJR_WPB").wpb;
        const int64_t gpr = (int64_t)device_cu_count() * wpb * G;
        const int64_t nr = gpr > 0 && max_groups % gpr == 0 ? max_groups / gpr : 0;
        const int64_t nch = B > 0 ? B / PPG : 1;
        if (B > 0 && nr >= 2 && !(A * nch == max_groups && nch % nr == 0))
            for (int64_t m = nr; m <= B && m <= max_groups; m += nr)
                if (m >= nch && B % m == 0 && max_groups % m == 0 && B / m >= 4 * nr) {
                    if (A >= max_groups / m) { *rows_per_launch = max_groups / m; PPG = B / m; }
                    break;
                }
    }
    if (PPG > 0x3fffffff / NUp) return SK_ERR_UNSUPPORTED;
    const int64_t groups = g.P / PPG;
    const int OUTW = ND + 2;
    if (ppg_out) *ppg_out = (int)PPG;
    if (rows_out) *rows_out = L * RC + 1;
    if (outw_out) *outw_out = OUTW;
    if (!gpart) return SK_OK;
    if (gpart_doubles < (size_t)groups * (L * RC + 1) * OUTW) return SK_ERR_WORKSPACE;
    const int64_t waves = (groups + G - 1) / G;

    AdjRbfParams prm;
    prm.Xr = Xr; prm.Yt = Yt; prm.edges = edges; prm.scale = scale; prm.Gpart = gpart; prm.err = err;
    prm.P = g.P; prm.B = B; prm.Mrows = Mrows; prm.Ncp = Ncp; prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = NUp; prm.logL = logL;
    prm.PPG = (int)PPG;
    prm.inv_sigma = inv_sigma;
    prm.n_steps = (int)(PPG * NUp + (L - 1)) + 1;    // + 1: node column 0 of the last pair completes one step later
    prm.wg = wave_group(lds_bytes, waves, "SK_ADJR_WPB");
    prm.cs = chunk_split(A, B, PPG, max_groups, G, prm.wg.wpb, device_cu_count(), "SK_ADJR_RANK_W");
    const size_t lds_block = wave_group_lds(prm.wg);
    const bool full = logL == 6;
    if (DY == 1) {
        if (ND == 4) return full ? launch_adjr<1, 2, true, 4>(prm, lds_block, s) : launch_adjr<1, 2, false, 4>(prm, lds_block, s);
        return full ? launch_adjr<1, 2, true, 8>(prm, lds_block, s) : launch_adjr<1, 2, false, 8>(prm, lds_block, s);
    }
    if (ND == 4) return full ? launch_adjr<2, 1, true, 4>(prm, lds_block, s) : launch_adjr<2, 1, false, 4>(prm, lds_block, s);
    return full ? launch_adjr<2, 1, true, 8>(prm, lds_block, s) : launch_adjr<2, 1, false, 8>(prm, lds_block, s);
}
}  // namespace

int launch_adj_fused_rbf(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                         double inv_sigma, const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *err,
                         int *ppg_out, int *rows_out, int *outw_out, hipStream_t s) {
    int ppg = 0, rows = 0, outw = 0;
    int64_t per_launch = 0;
    int rc = launch_adj_fused_rbf_rows(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, edges, scale, nullptr, 0, err, &ppg, &rows, &outw, &per_launch, 0, s);
    if (rc != SK_OK) return rc;
    if (ppg_out) *ppg_out = ppg;
    if (rows_out) *rows_out = rows;
    if (outw_out) *outw_out = outw;
    if (!gpart) return SK_OK;
    if (per_launch <= 0 || B <= 0)
        return launch_adj_fused_rbf_rows(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, edges, scale, gpart, gpart_doubles, err, nullptr, nullptr, nullptr,
                                         nullptr, B > 0 ? B / ppg : 0, s);
    // several launches of per_launch rows each, all with the same chunks per a (so that gpart keeps one layout)
    const int64_t nch = B / ppg;
    const int64_t slot = (int64_t)rows * outw;
    if (gpart_doubles < (size_t)(A * nch * slot)) return SK_ERR_WORKSPACE;
    const Strip st = strip_geom(g, 8);
    const int64_t Epair = (int64_t)st.NUp * (2 << g.dyadic) + (int64_t)(1 << st.logL) * st.RC * (1 << g.dyadic);   // edge doubles per pair
    for (int64_t a0 = 0; a0 < A; a0 += per_launch) {
        const int64_t An = A - a0 < per_launch ? A - a0 : per_launch;
        Geom gs = g;
        gs.P = An * B;
        rc = launch_adj_fused_rbf_rows(Xr + a0 * Mrows * RFD, Yt, An, B, Mrows, Ncp, D, gs, inv_sigma, edges + a0 * B * Epair,
                                       scale ? scale + a0 * B : nullptr, gpart + a0 * nch * slot, (size_t)(An * nch * slot),
                                       err ? err + a0 * B : nullptr, nullptr, nullptr, nullptr, nullptr, nch, s);
        if (rc != SK_OK) return rc;
    }
    return SK_OK;
}

}  // namespace sk
