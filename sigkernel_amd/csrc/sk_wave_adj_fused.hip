// sk_wave_adj_fused.hip -- the adjoint solver with the LINEAR static kernel fused in, both ways:
//   * the increments of the reverse sweep are formed from the path differences inside the kernel (the two LDS rings of
//     sk_wave_fused.hip, filled back to front), so no increment matrix is read;
//   * the weights W = d k / d inc are contracted on the spot with the y differences they belong to,
//         T[a][p][:] = sum_b s_ab sum_q W[a,b,p,q] (y_b[q+1] - y_b[q])          (what sk_linear_adjoint_* forms from W),
//     accumulated in registers over the pairs of one lane group, so no W matrix is written or read either.
// What remains in HBM is the paths, the terminal edges the (fused) forward kept, and (chunks x A x M x 8) partial sums.
// The two-state sweep itself (reverse PDE + backward recompute of K from the edges, self-check) is the one of
// sk_wave_adj.hip; see there for the mathematics and for the edge prefetch.
//
// Decomposition: a lane group sweeps PPG consecutive pairs (a, b0 .. b0 + PPG - 1) of ONE path x_a -- the launcher splits the B pairs
// of an a into as many chunks as the resident lane groups take (pick_chunk; lengths differ by one where that number does not divide
// B) -- so its registers hold a partial sum over b for that a; it is stored (plain stores, no atomics: the result does not depend on
// scheduling) to Tpart[group][flipped row][8], and the host adds the ceil(B / PPG) chunks of an a.
// Paired batches (B = 0) run with PPG = 1: one lane group per pair.
// Scope: fp64, path dim <= 8, one band per pair (dyadic 0: up to 128 increment rows), dyadic <= 2, default scheme.
#include "sk_wave_common.h"

namespace sk {
namespace {

constexpr int FD = 8;
constexpr int Y_SLAB_PITCH = FD * 128;
constexpr int X_SLOTS = 2;

struct AdjFusedParams {
    const double *dXr;     // [A][Mrows][8]  s^2 (x[p+1]-x[p]), zero rows / dims beyond Mc / D
    const double *dYt;     // [B][8][Ncp]    y[q+1]-y[q], dimension-major, zero columns / dims beyond Nc / D
    const double *edges;   // [P][NNp + MMp] strip layout (strip_geom)
    const double *scale;   // [P] upstream gradient per pair, nullable
    double *Tpart;         // [P / PPG][L*RC][8]  partial sums, flipped coarse rows
    double *Ypart;         // YSIDE: [P][2 NUp][8] per pair and increment column q of y_b: sum_p W[p][q] s^2 (x[p+1]-x[p]), WITHOUT the
                           // upstream gradient (the caller weights and adds the pairs of a second path)
    double *err;           // [P] zero-initialised: worst |Kf - 1| on the recomputed boundary
    int64_t P, B;          // B > 0: Gram, pair p = (p / B, p % B); B == 0: paired, pair p = (p, p) and PPG = 1
    int Mrows, Ncp, Mc, Nc, NUp, logL, PPG, n_steps;
    ChunkSplit cs;         // chunk sizes by wave age rank; PPG / n_steps are the equal split's
    int E;                 // doubles per pair in `edges`
    WaveGroup wg;
    int naive;             // _naive_solver stencil: c_12 = 0 (a = 1 + g/2, b = 1 exactly; see sk_wave_fused_mb.hip)
};

__device__ __forceinline__ void lds_read_dims8(d2_t (&v)[8], unsigned a_even, unsigned a_odd) {
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %9\n\t"
                 "ds_read_b128 %2, %8 offset:256\n\t"
                 "ds_read_b128 %3, %9 offset:256\n\t"
                 "ds_read_b128 %4, %8 offset:512\n\t"
                 "ds_read_b128 %5, %9 offset:512\n\t"
                 "ds_read_b128 %6, %8 offset:768\n\t"
                 "ds_read_b128 %7, %9 offset:768\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(a_even), "v"(a_odd)
                 : "memory");
}
// RC consecutive 64-byte rows, one wait
template <int NB128>
__device__ __forceinline__ void lds_read_run(d2_t (&v)[NB128], unsigned a);
template <>
__device__ __forceinline__ void lds_read_run<4>(d2_t (&v)[4], unsigned a) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a) : "memory");
}

template <>
__device__ __forceinline__ void lds_read_run<8>(d2_t (&v)[8], unsigned a) {      // two consecutive 64-byte rows, one wait
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\t"
                 "ds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(a) : "memory");
}

// RC = coarse rows per lane: the forward kernels' choice at dyadic 1, 2; at dyadic 0 two instead of their four (register budget)
// YSIDE: the SECOND-argument sums instead of the first-argument ones (route FUSED_SWAP: long first paths against short second ones are
// swept as (y, x)): per unit the lanes hand the running sum over their rows DOWN THE WAVE -- lane l + 1 sweeps at macro-step t + 1 the
// unit lane l swept at t, so one wave_shr:1 of the 16 sums per macro-step (32 DPP moves) carries them along; a group's top lane starts
// from zero, its bottom lane stores the unit's sums of its pair.  (First built with the carry in LDS, eight 65-slot pieces read with
// the y differences and written back: 24 KB of LDS traffic per wave and macro-step against 8 -- the dyadic-0 sweep, four cells per
// step, ran 1.87x the time of the first-argument form on the same grid, profiles/r06_asym.txt; the DPP form costs issue slots only.)
// PAIRED: a paired batch (B == 0) with SEVERAL pairs per lane group: every pair has its own x, so a lane stores and clears its sums when
// it finishes a pair (one pair per lane group, the form until round 6, pays the skew's fill and the wave's prologue for every pair:
// 262 144 pairs of 64 points 68 ns per pair with a gradient against 20 for a Gram pair, profiles/r06_aspect.txt)
template <int DY, int RC, bool FULLWAVE, bool YSIDE, bool PAIRED = false>
__global__ __launch_bounds__(4 * WAVE) void k_adj_fused_linear(const AdjFusedParams prm) {
    constexpr int CW = 2;
    constexpr int R = RC << DY, S = CW << DY, r = 1 << DY;
    constexpr int XSLAB = RC * 512;
    constexpr int NPC = 4 * S + 1, ECG = NPC * 16;   // terminal-row chunk of one lane group (sk_wave_adj.hip)
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
    const unsigned y_bytes = (unsigned)(NSLAB * Y_SLAB_PITCH);
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
    // differ by one pair: a shorter group's last pair position is swept unweighted)
    const int ppg_own = PPG;
    PPG = __builtin_amdgcn_readfirstlane(PPG);
    if (prm.cs.uneven)
        for (int gq = 1; gq < G; ++gq) PPG = max(PPG, __builtin_amdgcn_readlane(ppg_own, gq << prm.logL));
    const int n_steps = PPG * NUp + (L - 1);
    auto group_first = [&](int g) -> int64_t { return readlane64(pair0, g << prm.logL); };
    const bool is_top = lam == 0;
    const unsigned my_y = lds0 + (unsigned)grp * y_bytes;
    const int JMAX = (L + NUp - 1) / NUp;
    const unsigned my_x = lds0 + x_base0 + (unsigned)((grp * X_SLOTS * JMAX) * XSLAB + (lam / NUp) * XSLAB) +
                          (unsigned)((lam & 7) * RC * 64);
    const unsigned ec_off = x_base0 + (unsigned)(G * X_SLOTS * JMAX * XSLAB);   // terminal-row chunks behind the rings
    const unsigned ec_slot = (unsigned)(G * ECG);
    const bool is_bot = lam == L - 1;

    // ---- producers: the rings of sk_wave_fused.hip, filled in FLIPPED order ------------------------------------------------
    // y slab s = flipped units [8s, 8s+8) of the group's stream; flipped unit u' of a pair is original unit NUp-1-u' (its two
    // columns stay in original order inside the 16-byte unit, as in the increment matrix the unfused kernel reads)
    auto split_b = [&](int64_t p) -> int64_t {   // B == 0: paired, pair p = (x_p, y_p)
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
            const int uo = NUp - 1 - (y_u0 + (lane & 7));
            const double *src = prm.dYt + ((b * FD + krow) * (int64_t)prm.Ncp + (int64_t)uo * 2);
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(lds + g * y_bytes + y_slot * Y_SLAB_PITCH), 16, 0, 0);
        }
        y_slot = y_slot + 1 == NSLAB ? 0 : y_slot + 1;
        y_par ^= 1;
        y_u0 += 8;
        if (y_u0 == NUp) { y_u0 = 0; y_pi += 1; }
    };
    // x slabs: lanes lamj .. lamj+7 start pair pi during the window; LDS position i = (lam & 7) * RC + k holds the flipped
    // coarse row lamj*RC + i, i.e. original row Mcp - 1 - (lamj*RC + i): each DMA lane fetches its 16-byte piece from there
    int x_q0 = 0, x_lam0 = 0, x_slot = 0;
    auto issue_x = [&]() {
        for (int j = 0; j < JMAX; ++j) {
            const int lamj = x_lam0 + j * NUp, pi = x_q0 - j;
            if (lamj >= L) break;
            for (int g = 0; g < G; ++g) {
                int64_t p = group_first(g) + pi;
                if (pi < 0 || pi >= PPG || p >= prm.P) p = 0;
                const int64_t a = split_a(p);
                char *dst = lds + x_base0 + ((g * X_SLOTS + x_slot) * JMAX + j) * XSLAB;
#pragma unroll
                for (int c = 0; c < (XSLAB + 1023) / 1024; ++c)
                    if (c * 1024 + lane * 16 < XSLAB) {
                        const int i = c * 16 + (lane >> 2);                 // row position inside the slab
                        const int row = Mcp - 1 - (lamj * RC + i);          // original coarse row (>= Mc: zero padding)
                        const double *src = prm.dXr + (a * prm.Mrows + row) * FD + (lane & 3) * 2;
                        __builtin_amdgcn_global_load_lds(src, (lds_void *)(dst + c * 1024), 16, 0, 0);
                    }
            }
        }
        x_slot = x_slot + 1 == X_SLOTS ? 0 : x_slot + 1;
        x_lam0 += 8;
        if (x_lam0 == NUp) { x_lam0 = 0; x_q0 += 1; }
    };

    // ---- terminal edges and the upstream gradient of the coming pair, one macro-step ahead (sk_wave_adj.hip) ---------------
    // pair stride of the edges: the layout of the kernel that wrote them (at dyadic 0 its padded row count differs from ours)
    const int E = DY == 0 ? prm.E : NNp + MMp;
    // the terminal ROW reaches the top lanes through LDS chunks fetched once per window of 8 macro-steps (sk_wave_adj.hip:
    // issue_edge_chunk); the terminal COLUMN and the upstream gradient are loaded into registers one macro-step ahead
    int ec_u0 = 0, ec_ps = 0, ec_fill = 0;
    auto issue_edge_chunk = [&]() {
        for (int c = 0; c * WAVE < G * NPC; ++c) {
            const int idx = c * WAVE + lane, g = idx / NPC, i = idx - g * NPC;
            int64_t pr = gather64(pair0, (g < G ? g : 0) << prm.logL) + ec_ps;   // (g differs per lane here)
            pr = (ec_ps >= PPG || pr >= prm.P) ? 0 : pr;
            const int k = NNp - (ec_u0 + LINE_UNITS) * S - 2 + 2 * i;
            if (g < G && k >= 0)
                __builtin_amdgcn_global_load_lds(prm.edges + pr * E + k, (lds_void *)(lds + ec_off + ec_fill * (G * ECG) + c * 1024), 16, 0, 0);
        }
        ec_fill ^= 1;
        ec_u0 += LINE_UNITS;
        if (ec_u0 == NUp) { ec_u0 = 0; ec_ps += 1; }
    };
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

    // YSIDE: the upstream gradient only says which pairs are swept (NaN: screened out; the sums stay unweighted): 1 for a pair of the
    // group that exists, NaN kept, 0 outside
    auto pair_scale = [&](int ps_, double sv) -> double {
        if (ps_ < 0 || ps_ >= ppg_own) return 0.0;
        const double v = prm.scale ? sv : 1.0;
        if constexpr (YSIDE) return pair0 + ps_ < prm.P ? (v != v ? v : 1.0) : 0.0;
        return v;
    };
    double dxr[RC][FD], tacc[YSIDE ? 1 : RC][FD];
#pragma unroll
    for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int j = 0; j < FD; ++j) { dxr[k][j] = 0.0; if (!YSIDE || k == 0) tacc[YSIDE ? 0 : k][j] = 0.0; }
    double car[YSIDE ? CW : 1][FD];     // YSIDE: the second-argument sums of the unit being swept, over the rows of the lanes above
#pragma unroll
    for (int q = 0; q < (YSIDE ? CW : 1); ++q)
#pragma unroll
        for (int j = 0; j < FD; ++j) car[q][j] = 0.0;
    double ktopR[S];
#pragma unroll
    for (int i = 0; i < S; ++i) ktopR[i] = 1.0;
    double leftR[R], botR[S], cornerR = 1.0;
    double leftF[R], botF[S], cornerF = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = 1.0; }
#pragma unroll
    for (int i = 0; i < S; ++i) { botR[i] = 1.0; botF[i] = 1.0; }
    double chk_val = 0.0;
    int64_t chk_pair = -1;
    double s_pair = 0.0;    // upstream gradient of the pair being swept (0 outside the group's pairs)
    double ncol[R + 1], nscale = 0.0;
#pragma unroll
    for (int i = 0; i <= R; ++i) ncol[i] = 1.0;

    {   // lanes ahead of their first pair read slabs no DMA has written yet: make those finite (see the contraction below)
        const int total = (int)(G * y_bytes + G * X_SLOTS * JMAX * XSLAB + 2 * G * ECG);
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
        nscale = u == 0 ? pair_scale(ps, tsc[0]) : 0.0;
    }
    issue_y();
    issue_x();

    for (int t = 0; t < n_steps; ++t) {
        // the top lane's terminal-row values of this macro-step (no wait: complete at the y read's lgkmcnt(0) below)
        double trow_p[S], trow[S];
#pragma unroll
        for (int i = 0; i < S; ++i) async_begin(trow_p[i]);
        lds_read_f64_run<S>(trow_p, lds0 + ec_off + (unsigned)(((t >> 3) & 1) * ec_slot + grp * ECG + ((7 - (t & 7)) * S + 1) * 8));
        if (chk_pair >= 0) {
            atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
            chk_pair = -1;
        }
        int nu = u + 1, nps = ps;
        if (nu == NUp) { nu = 0; nps += 1; }

        // -- start of a (flipped) pair: boundaries, upstream gradient, this lane's x rows
        if (u == 0) {
            cornerR = 1.0;
            cornerF = ncol[0];
#pragma unroll
            for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = ncol[i + 1]; }
            s_pair = nscale;
            const unsigned xa = my_x + (unsigned)(((t >> 3) % X_SLOTS) * JMAX * XSLAB);
            // (some lane starts a pair in every macro-step: the rows of a lane in ONE asm with one wait, profiles/r06_small_launch_pmc.txt)
            {
                d2_t xv[4 * RC];
                lds_read_run<4 * RC>(xv, xa);
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { dxr[k][2 * j] = xv[4 * k + j][0]; dxr[k][2 * j + 1] = xv[4 * k + j][1]; }
            }
        }

        // -- y differences of the unit (original column order inside the unit)
        d2_t dyv[FD];
        {
            const unsigned ya = my_y + (unsigned)(yslab * Y_SLAB_PITCH + ((u & 7) << 4));
            lds_read_dims8(dyv, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
        }
        lds_take<S>(trow, trow_p);
        if (__builtin_expect((t & 7) == 0, 0)) issue_edge_chunk();

        // -- top rows
        double topR[S], topF[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            double tf = trow[S - 1 - i];
            if (i == S - 1 && u == NUp - 1) tf = 1.0;     // K[MM][0] = 1 is not stored
            if (FULLWAVE) {
                ktopR[i] = dpp_shr1(botR[i], ktopR[i]);
                topR[i] = ktopR[i];
                topF[i] = dpp_shr1(botF[i], tf);
            } else {
                const double shR = dpp_shr1(botR[i], 1.0);
                const double shF = dpp_shr1(botF[i], 1.0);
                topR[i] = is_top ? 1.0 : shR;
                topF[i] = is_top ? tf : shF;
            }
        }

        // -- next step's edge values (asynchronous)
        double pcol[R + 1], pscale[1];
#pragma unroll
        for (int i = 0; i <= R; ++i) async_begin(pcol[i]);
        async_begin(pscale[0]);
        prefetch_edges(nu, nps, pcol, pscale[0]);

        // -- increments (original column order q = 0, 1 of the unit) and coefficients
        double ginc[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                double g = 0.0;
#pragma unroll
                for (int j = 0; j < FD; ++j) g = fma(dxr[k][j], dyv[j][q], g);
                ginc[k][q] = g;
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

        // -- W of the RC x 2 coarse cells, contracted with the y differences of their columns.  Lanes outside their group's
        //    pairs sweep leftovers whose weights may be anything, NaN included: their w is SELECTED to zero, and the y values
        //    they multiply are finite because the rings were zero-filled before the first DMA (0 * NaN would poison the sum)
        {
            const bool live = s_pair != 0.0 && s_pair == s_pair;   // (NaN: a pair the rescue's screen took out of the sweep)
            const double wsc = sc * s_pair;
            double wq[RC][CW];
#pragma unroll
            for (int k = 0; k < RC; ++k) { wq[k][0] = live ? acc[k][1] * wsc : 0.0; wq[k][1] = live ? acc[k][0] * wsc : 0.0; }   // original columns 0, 1
            if constexpr (!YSIDE) {
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int j = 0; j < FD; ++j) tacc[k][j] = fma(wq[k][0], dyv[j][0], fma(wq[k][1], dyv[j][1], tacc[k][j]));
            } else {
                // second argument: column q of the unit gets sum_k w[k][q] dxr[k][:], on top of what the lanes above summed for it one
                // macro-step ago: the sums move one lane down, a group's top lane starts from zero
#pragma unroll
                for (int q = 0; q < CW; ++q)
#pragma unroll
                    for (int j = 0; j < FD; ++j) car[q][j] = dpp_shr1_zero(car[q][j]);
                if (!FULLWAVE && is_top) {
                    asm volatile("");      // a real branch under the exec mask: sixteen moves, not thirty-two selects
#pragma unroll
                    for (int q = 0; q < CW; ++q)
#pragma unroll
                        for (int j = 0; j < FD; ++j) car[q][j] = 0.0;
                }
#pragma unroll
                for (int q = 0; q < CW; ++q)
#pragma unroll
                    for (int k = 0; k < RC; ++k)
#pragma unroll
                        for (int j = 0; j < FD; ++j) car[q][j] = fma(wq[k][q], dxr[k][j], car[q][j]);
                if (is_bot && live) {
                    const int uo = NUp - 1 - u;
                    double *yp = prm.Ypart + ((pair0 + ps) * (int64_t)(2 * NUp) + 2 * uo) * FD;
#pragma unroll
                    for (int q = 0; q < CW; ++q)
#pragma unroll
                        for (int j = 0; j < FD; j += 2) *reinterpret_cast<d2_t *>(yp + q * FD + j) = d2_t{car[q][j], car[q][j + 1]};
                }
            }
        }

        // -- PAIRED: this lane's rows of the pair are complete with its last flipped unit: out they go (slot = pair), the sums start over
        if constexpr (PAIRED) {
            if (u == NUp - 1) {
                if (ps >= 0 && ps < ppg_own && pair0 + ps < prm.P) {
                    double *dst = prm.Tpart + ((pair0 + ps) * Mcp + (int64_t)lam * RC) * FD;
#pragma unroll
                    for (int k = 0; k < RC; ++k)
#pragma unroll
                        for (int j = 0; j < FD; j += 2) *reinterpret_cast<d2_t *>(dst + k * FD + j) = d2_t{tacc[k][j], tacc[k][j + 1]};
                }
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int j = 0; j < FD; ++j) tacc[k][j] = 0.0;
            }
        }

        // -- self-check on the last flipped unit (see sk_wave_adj.hip)
        if (u == NUp - 1 && prm.err && ps >= 0 && ps < ppg_own && pair0 + ps < prm.P && s_pair == s_pair) {
            double e = 0.0;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) e = fmax(e, fabs(leftF[rr] - 1.0));
            chk_val = e;
            chk_pair = pair0 + ps;
        }

        // -- close the step: the edge values (and any ring piece issued in the previous step) have landed
        {
            double tsc[1];
            async_wait<0>(ncol, pcol);
            async_wait<0>(tsc, pscale);
            fix_edges(nu, ncol);
            if (nu == 0) nscale = pair_scale(nps, tsc[0]);
        }

        // -- advance; the next slab / window is requested right after the wait, so it has a whole macro-step before the
        //    next closing wait asks for it
        u = nu;
        ps = nps;
        if (((t + 1) & 7) == lam7) {
            yslab = yslab + 1 == NSLAB ? 0 : yslab + 1;
            ypar ^= 1;
        }
        if (__builtin_expect(((t + 1) & 7) == 0, 0)) {
            issue_y();
            issue_x();
        }
    }
    if (chk_pair >= 0)
        atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));

    // ---- the group's partial sums: Tpart[group][flipped coarse row][8] ------------------------------------------------------
    {
        if (!YSIDE && !PAIRED && pair0 < prm.P) {
            double *dst = prm.Tpart + (gslot * Mcp + (int64_t)lam * RC) * FD;
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int j = 0; j < FD; j += 2) {
                    d2_t v = {tacc[k][j], tacc[k][j + 1]};
                    *reinterpret_cast<d2_t *>(dst + k * FD + j) = v;
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DY, int RC, bool FULLWAVE, bool YSIDE, bool PAIRED = false>
int launch_adjf(const AdjFusedParams &prm, size_t lds_block, hipStream_t s) {
    auto kern = k_adj_fused_linear<DY, RC, FULLWAVE, YSIDE, PAIRED>;
    if (lds_block > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
    SK_LAUNCH(kern, dim3(wave_group_blocks(prm.wg)), dim3(WAVE * prm.wg.wpb), lds_block, s, prm);
    return check_launch();
}

}  // namespace

// Rows of Tpart = (P / PPG) * L * RC; *ppg_out / *rows_out tell the caller how to fold it: Tpart viewed as
// [A][B / PPG][L*RC][8], summed over the chunks, rows flipped (coarse row p = L*RC - 1 - r).  tpart == nullptr: query only.
namespace {
int launch_adj_fused_linear_rows(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Ncp, const Geom &g,
                                 const double *edges, const double *scale, double *tpart, size_t tpart_doubles, double *err,
                                 double *ypart, bool yside, int *ppg_out, int *rows_out, int *ycols_out, int64_t *rows_per_launch,
                                 int64_t *epair, int64_t force_nch, const FusedRescue *rescue, const double *scale_orig, void *rescue_ws,
                                 size_t rescue_ws_bytes, hipStream_t s) {
    const int DY = g.dyadic;
    if (DY > 2 || B < 0 || g.P != (B > 0 ? A * B : A)) return SK_ERR_UNSUPPORTED;
    const Strip st = strip_geom(g, 8);   // the layout of the edges
    if (!st.ok || st.nb != 1) return SK_ERR_UNSUPPORTED;
    // dyadic 0: the strip kernels give a lane four coarse rows; with the two 4 x 8 register arrays of this kernel that is
    // 316 VGPRs, so it sweeps two rows per lane (pairs of up to 128 increments rows) and only shares the edge layout
    const int RC = DY == 0 ? 2 : st.RC, NUp = st.NUp;
    int logL = 3;
    while (logL < 6 && (RC << logL) < g.Mc) ++logL;
    const int L = 1 << logL, G = WAVE / L;
    if (L * RC < g.Mc) return SK_ERR_UNSUPPORTED;
    if (Ncp < NUp * 2 || (Ncp & 1) || Mrows < L * RC) return SK_ERR_UNSUPPORTED;
    const int JMAX = (L + NUp - 1) / NUp;
    const int S = 2 << DY;
    const size_t lds_bytes = (size_t)G * (((L >> 3) + 2) * Y_SLAB_PITCH + X_SLOTS * JMAX * RC * 512) + (size_t)2 * G * (4 * S + 1) * 16;
    if (lds_bytes > 160 * 1024) return SK_ERR_UNSUPPORTED;
    if (yside && B <= 0) return SK_ERR_UNSUPPORTED;   // (paired batches have no use for it: both arguments fit or neither does)

    // the resident lane groups: 8 waves per CU (the kernel holds ~230 VGPRs, two waves per SIMD)
    const int wpc = knobs().adjf_wpc > 0 ? knobs().adjf_wpc : 8;
    const int64_t max_groups = (int64_t)device_cu_count() * wpc * G;
    // pairs per lane group: see pick_chunk (paired batches: every pair has its own x, one pair per lane group)
    int64_t PPG = pick_chunk(A, B, max_groups);
    // paired batches of more pairs than resident lane groups: several consecutive pairs per lane group (the kernel stores and clears its
    // sums at every pair end: PAIRED), up to 64 -- the skew's fill and the wave's prologue are then paid once per group, not per pair
    int64_t ppp = 1;
    if (B <= 0 && !yside) {
        ppp = g.P / max_groups;
        ppp = ppp < 1 ? 1 : (ppp > 64 ? 64 : ppp);
        PPG = ppp;
    }
    if (epair) *epair = st.NNp + st.MMp;
    if (force_nch > 0) PPG = (B + force_nch - 1) / force_nch;
    else if (rows_per_launch) {   // see launch_adj_fused_rbf_rows (sk_wave_adj_fused_rbf.hip)
        *rows_per_launch = 0;
        const int wpb = wave_group(lds_bytes, max_groups / G, knobs().adjf_wpb).wpb;
        const int64_t gpr = (int64_t)device_cu_count() * wpb * G;
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
    if (ppg_out) *ppg_out = (int)PPG;
    if (rows_out) *rows_out = L * RC;
    if (ycols_out) *ycols_out = 2 * NUp;
    if (yside ? !ypart : !tpart) return SK_OK;
    if (!yside && tpart_doubles < (size_t)(B > 0 ? groups : g.P) * L * RC * FD) return SK_ERR_WORKSPACE;   // (paired: a slot per pair)
    const int64_t waves = (groups + G - 1) / G;

    AdjFusedParams prm;
    prm.dXr = dXr; prm.dYt = dYt; prm.edges = edges; prm.scale = scale; prm.Tpart = tpart; prm.err = err; prm.Ypart = ypart;
    prm.P = g.P; prm.B = B; prm.Mrows = Mrows; prm.Ncp = Ncp; prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = NUp; prm.logL = logL;
    prm.PPG = (int)PPG;
    prm.E = st.NNp + st.MMp;
    prm.naive = g.naive;
    prm.n_steps = (int)(PPG * NUp + (L - 1));
    prm.wg = wave_group(lds_bytes, waves, knobs().adjf_wpb);
    prm.cs = chunk_split(A, B, PPG, max_groups, G, prm.wg.wpb, device_cu_count(), knobs().adjf_rank_w);
    prm.cs.ppp = (int)ppp;
    const size_t lds_block = wave_group_lds(prm.wg);
    const bool full = logL == 6;
    int rc;
    if (yside)
        switch (DY) {
            case 0: rc = full ? launch_adjf<0, 2, true, true>(prm, lds_block, s) : launch_adjf<0, 2, false, true>(prm, lds_block, s); break;
            case 1: rc = full ? launch_adjf<1, 2, true, true>(prm, lds_block, s) : launch_adjf<1, 2, false, true>(prm, lds_block, s); break;
            default: rc = full ? launch_adjf<2, 1, true, true>(prm, lds_block, s) : launch_adjf<2, 1, false, true>(prm, lds_block, s); break;
        }
    else if (ppp > 1)
        switch (DY) {
            case 0: rc = full ? launch_adjf<0, 2, true, false, true>(prm, lds_block, s) : launch_adjf<0, 2, false, false, true>(prm, lds_block, s); break;
            case 1: rc = full ? launch_adjf<1, 2, true, false, true>(prm, lds_block, s) : launch_adjf<1, 2, false, false, true>(prm, lds_block, s); break;
            default: rc = full ? launch_adjf<2, 1, true, false, true>(prm, lds_block, s) : launch_adjf<2, 1, false, false, true>(prm, lds_block, s); break;
        }
    else
        switch (DY) {
            case 0: rc = full ? launch_adjf<0, 2, true, false>(prm, lds_block, s) : launch_adjf<0, 2, false, false>(prm, lds_block, s); break;
            case 1: rc = full ? launch_adjf<1, 2, true, false>(prm, lds_block, s) : launch_adjf<1, 2, false, false>(prm, lds_block, s); break;
            default: rc = full ? launch_adjf<2, 1, true, false>(prm, lds_block, s) : launch_adjf<2, 1, false, false>(prm, lds_block, s); break;
        }
    if (rc != SK_OK || !rescue || !rescue_ws) return rc;
    // (second-argument sums: the rescue writes a rescued pair's block of ypart and has no partial sums to patch)
    ChunkSplit rcs = prm.cs;      // (paired: every pair has its slot whatever the lane groups swept -- the rescue walks pairs)
    rcs.ppp = 1;
    if (B <= 0) rcs.size[0] = 1;
    return launch_fused_rescue(0, dXr, dYt, scale_orig, err, rescue->tol, yside ? nullptr : tpart, ypart, A, B, Mrows, Ncp, 8, g, L * RC, FD,
                               2 * NUp, 0.0, rcs, B > 0 ? groups : g.P, rescue_ws, rescue_ws_bytes, s, 8, nullptr, 0, rescue->kfinal);
}
}  // namespace

// ypart / ycols_out (either non-null: Gram only): the SECOND-argument sums instead, [A B][*ycols_out][8] per pair and increment column
// of y_b, without the upstream gradient; tpart is not touched then.  Query (sizes only): tpart == ypart == nullptr.
int launch_adj_fused_linear(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Ncp, const Geom &g,
                            const double *edges, const double *scale, double *tpart, size_t tpart_doubles, double *err, double *ypart,
                            size_t ypart_doubles, int *ppg_out, int *rows_out, int *ycols_out, const FusedRescue *rescue, hipStream_t s) {
    int ppg = 0, rows = 0, ycols = 0;
    int64_t per_launch = 0, epair = 0;
    const bool yside = ypart != nullptr || ycols_out != nullptr;
    int rc = launch_adj_fused_linear_rows(dXr, dYt, A, B, Mrows, Ncp, g, edges, scale, nullptr, 0, err, nullptr, yside, &ppg, &rows, &ycols,
                                          &per_launch, &epair, 0, nullptr, nullptr, nullptr, 0, s);
    if (rc != SK_OK) return rc;
    if (ppg_out) *ppg_out = ppg;
    if (rows_out) *rows_out = rows;
    if (ycols_out) *ycols_out = ycols;
    if (yside ? !ypart : !tpart) return SK_OK;
    if (ypart && ypart_doubles < (size_t)g.P * ycols * FD) return SK_ERR_WORKSPACE;
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
        return launch_adj_fused_linear_rows(dXr, dYt, A, B, Mrows, Ncp, g, edges, sweep_scale, tpart, tpart_doubles, err, ypart, yside, nullptr,
                                            nullptr, nullptr, nullptr, nullptr, B > 0 ? chunks_of(B, ppg) : 0, rescue, scale, rws, rws_bytes, s);
    // several launches of per_launch rows each, all with the same chunks per a (so that tpart keeps one layout)
    const int64_t nch = chunks_of(B, ppg), slot = (int64_t)rows * FD;
    if (!yside && tpart_doubles < (size_t)(A * nch * slot)) return SK_ERR_WORKSPACE;
    for (int64_t a0 = 0; a0 < A; a0 += per_launch) {
        const int64_t An = A - a0 < per_launch ? A - a0 : per_launch;
        Geom gs = g;
        gs.P = An * B;
        rc = launch_adj_fused_linear_rows(dXr + a0 * Mrows * FD, dYt, An, B, Mrows, Ncp, gs, edges + a0 * B * epair,
                                          sweep_scale ? sweep_scale + a0 * B : nullptr, yside ? nullptr : tpart + a0 * nch * slot,
                                          (size_t)(An * nch * slot), err ? err + a0 * B : nullptr,
                                          ypart ? ypart + a0 * B * (int64_t)ycols * FD : nullptr, yside, nullptr, nullptr, nullptr, nullptr,
                                          nullptr, nch, rescue, scale ? scale + a0 * B : nullptr, rws, rws_bytes, s);
        if (rc != SK_OK) return rc;
    }
    return SK_OK;
}

}  // namespace sk
