// sk_wave_adj.hip -- the fast adjoint solver.
//
// What it computes (reference: prep_backward, sigkernel.py:438-470, and _SigKernel.backward :282-311):
//   W[a][b] = 4^-d * sum over the fine cells (i,j) of coarse cell (a,b) of  K[i][j] * Krev[MM-1-i][NN-1-j]
// with K the forward solution and Krev the solution on the doubly flipped increments.
//
// How.  In flipped coordinates (i' = MM-1-i, j' = NN-1-j) the product is (K at the cell's far corner, seen
// from the flipped origin) x (Krev at the cell's near corner).  The kernel makes ONE sweep over the flipped grid,
// with exactly the machinery of the forward kernel (sk_wave.hip: skewed row strips in registers, DPP neighbour
// exchange, persistent pipelining over bands and pairs, just-in-time whole-line LDS-DMA of the increments --
// here fetched back to front), carrying two states per node:
//   Kr : the reverse PDE, ordinary recurrence                    Kr11 = a (Kr10 + Kr01) - b Kr00
//   Kf : the forward solution RECOMPUTED backwards from its terminal row and column (the `edges` the forward
//        kernel emitted, (MM+NN+2) doubles per pair) by solving the same stencil for the far corner:
//                                                                Kf11 = (a (Kf10 + Kf01) - Kf00) / b
// and accumulates Kf11 * Kr00 per coarse cell.  So the forward grid is never stored: forward + adjoint read the
// increments twice and write W once -- the algorithmic 3 (M-1)(N-1) s bytes of SURVEY 8(d) -- where storing
// K would cost 16 B per CELL.  The backward recurrence is as stable as the forward one while K stays O(1..1e3)
// (error ~ 1e-16 K_max^2); the kernel checks itself: the recomputed K must come out as 1 on the j = 0 boundary,
// the worst deviation per pair goes to `err`, and the host re-solves flagged pairs with the stored-grid kernel
// (sk_simple.hip).
//
// The sweep runs on the grid padded to whole bands and whole 128-byte lines; padding columns must hold zero
// increments (sk_increments_* writes them), padding rows are masked to zero here.  Zero increments propagate
// K unchanged, so the terminal edges are simply clamped.
#include "sk_wave_common.h"

namespace sk {
namespace {

struct AdjParams {
    const void *inc;       // [P, Mc, ld]
    const double *edges;   // [P, NNp + MMp]: K[MM][1..NNp], K[1..MMp][NN] (the forward strip kernel's layout)
    void *W;               // [P, Mc, ldw]
    double *err;           // [P] (zero-initialised by the caller) worst |Kf - 1| on the recomputed j=0 boundary
    int64_t P;
    int64_t ldb, ldwb;     // row strides in bytes
    int Mc, Nc;
    int NUp, nb, logL, PPG, n_steps, naive;
    WaveGroup wg;      // workgroups of independent waves (sk_wave_common.h)
    RankSplit rs;      // pairs per wave by age rank (sk_wave_common.h); PPG / n_steps are the largest share's
};

// Prefetch distance of the increment lines, in macro-steps.  Memory operations of a macro-step are issued at its top in
// the order [W line stores] [self-check atomic] [edge loads for the next step] [increment lines for step t + PF], and the
// step closes with s_waitcnt vmcnt((PF-1)*RC): loads and stores share vmcnt on gfx9, so "at most the (PF-1)*RC newest
// operations in flight" means everything but the newest increment lines has completed.  Ring = 8 (being consumed) + 1
// (being written out as W) + PF (in flight) slots.  Measured at d = 1 (131072 pairs of 127x127, forward + adjoint):
// PF = 1 with 8 waves/CU (20 KB of LDS per wave) 13.3 ms, PF = 2 with 7 waves/CU (22 KB) 14.4 ms, PF = 2 with 4 waves/CU
// 14.8 ms, PF = 1 with 4 waves/CU 17.4 ms: occupancy buys more than prefetch depth.
// Ring = 8 slots being consumed + PF in flight: the finished W line of a slot is read at the top of the very step that re-fetches
// the slot (the read is waited for before the fetch is issued), so no spare slot is needed.  PF = 1 where the ring is what
// bounds the resident waves (dyadic 0, 1: 36 / 18 KB + the edge chunks = 4 / 8 waves per CU), 2 at dyadic 2 (10 KB).
constexpr int adj_pf(int dy) { return dy >= 2 ? 2 : 1; }
// the pair's terminal row K[MM][.] reaches the top lanes through LDS: per window of 8 macro-steps one chunk of 8 S + 2 doubles
// per lane group (the 8 S values the window consumes, widened to 16-byte alignment), two slots
constexpr int adj_chunk_bytes(int S) { return (4 * S + 1) * 16; }

__device__ __forceinline__ void store_unit(double *dst, double a, double b) {
    d2_t v = {a, b};
    *reinterpret_cast<d2_t *>(dst) = v;
}
__device__ __forceinline__ void store_unit(float *dst, double a, double b, double c, double d) {
    f4_t v = {(float)a, (float)b, (float)c, (float)d};
    *reinterpret_cast<f4_t *>(dst) = v;
}

template <typename T, int DY, bool NAIVE, bool MULTIBAND, bool FULLWAVE>
__global__ __launch_bounds__(4 * WAVE) void k_adj_wave(const AdjParams prm) {
    constexpr int PF = adj_pf(DY);
    constexpr int CW = Unit<T>::CW;
    typedef typename Unit<T>::vec vec_t;
    constexpr int RC = Tile<DY>::RC, R = Tile<DY>::R, S = CW << DY, r = 1 << DY;
    // ring slots: a slot is fetched PF steps ahead, consumed for 8 steps while its units are overwritten in place by
    // the W units of the same positions, and written out as whole lines on the 9th step
    constexpr int NSLOT = LINE_UNITS + PF;
    constexpr int SLOT_BYTES = RC * 1024;
    extern __shared__ __attribute__((aligned(16))) char lds_block[];
    char *lds;
    const int64_t wave_id = wave_slot(prm.wg, lds_block, lds);
    if (wave_id < 0) return;
    const unsigned lds0 = lds_offset(lds);

    const int lane = threadIdx.x & (WAVE - 1);
    const int L = 1 << prm.logL, G = WAVE >> prm.logL;
    const int lam = lane & (L - 1), grp = lane >> prm.logL;
    const int NUp = prm.NUp, nb = prm.nb, NLp = NUp / LINE_UNITS;
    const int Mcp = nb * L * RC;                       // padded coarse rows
    const int MM = prm.Mc << DY, MMp = Mcp << DY, NNp = (NUp * CW) << DY;
    const double sc = 1.0 / (double)(1 << (2 * DY));
    const double c_half = 0.5 * sc, c_12 = sc * sc / 12.0;

    // ---- consumer state (flipped coordinates) ----------------------------------------------------------
    int u, band, ps;
    {
        const int sig = floor_div(-lam, NUp);
        u = -lam - sig * NUp;
        ps = floor_div(sig, nb);
        band = sig - ps * nb;
    }
    int PPG;               // this wave's pairs per lane group (by age rank, sk_wave_common.h), its first pair, the end of its rank
    int64_t first_pair, P_end;
    rank_share(prm.rs, wave_id, G, prm.P, PPG, first_pair, P_end);
    const int n_steps = PPG * nb * NUp + (L - 1);
    const int64_t pair0 = first_pair + (int64_t)grp * PPG;
    const bool is_top = lam == 0, is_bot = lam == L - 1;
    int slot = (((-(u & 7)) % NSLOT) + NSLOT) % NSLOT;
    const unsigned rd_lane = lds0 + (unsigned)(lane >> 3) * 128u;
    // LDS map: [increment ring][edge-row chunks: 2 slots x G groups][MULTIBAND: Kr and Kf band boundary rows]
    constexpr int ECG = adj_chunk_bytes(S);                 // one group's chunk
    const unsigned ec_base = lds0 + NSLOT * SLOT_BYTES;
    const unsigned ec_slot = (unsigned)(G * ECG);           // one slot (all groups)
    const unsigned bnd_r = ec_base + 2 * ec_slot + (unsigned)(grp * 2 * NUp * S) * 8u;
    const unsigned bnd_f = bnd_r + (unsigned)(NUp * S) * 8u;

    // ---- producer: increments, whole lines, back to front (see sk_wave.hip for the forward-order twin) ----
    const int64_t pair_bytes = (int64_t)prm.Mc * prm.ldb;
    int64_t span = (P_end - first_pair) * pair_bytes;
    const int64_t wave_span = (int64_t)G * PPG * pair_bytes;
    span = span < wave_span ? span : wave_span;
    if (span < 0) span = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(static_cast<const char *>(prm.inc) + first_pair * pair_bytes), 0, (int)span, 0x00020000);
    const int ldb = (int)prm.ldb;
    const int delta_band = NLp * 128 - L * RC * ldb;                                   // next (higher) band
    const int delta_pair = NLp * 128 + (nb - 1) * L * RC * ldb + (int)pair_bytes;   // bottom band of the next pair
    int st_m, st_band;
    unsigned st_off;
    {
        const int ip = (lane >> 3) & ((L >> 3) - 1);
        const int gc = (lane >> 3) >> (prm.logL - 3);
        const int v0 = -ip * LINE_UNITS;
        const int sg = floor_div(v0, NUp);
        st_m = (v0 - sg * NUp) / LINE_UNITS;
        const int ps0 = floor_div(sg, nb);
        st_band = sg - ps0 * nb;
        st_off = (unsigned)((gc * PPG + ps0) * (int)pair_bytes +
                            (Mcp - 1 - (st_band * L + ip * LINE_UNITS) * RC) * ldb + (NLp - 1 - st_m) * 128 +
                            (7 - (lane & 7)) * 16);
    }
    int fj = 0, fslot = 0;
    auto issue_fetch = [&]() {
#pragma unroll
        for (int k = 0; k < RC; ++k)   // aux = 2, streaming: the line is read exactly once, keep L2 for the W lines being written
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + fslot * SLOT_BYTES + k * 1024), 16,
                                                     st_off - (unsigned)((fj * RC + k) * ldb), 0, 0, 2);
        fslot = fslot + 1 == NSLOT ? 0 : fslot + 1;
        fj += 1;
        if (fj == LINE_UNITS) {
            fj = 0;
            st_m += 1;
            st_off -= 128;
            if (st_m == NLp) {
                st_m = 0;
                const bool last = st_band == nb - 1;
                st_off += last ? delta_pair : delta_band;
                st_band = last ? 0 : st_band + 1;
            }
        }
    };

    // ---- W write-out cursor: the same walk as the fetch cursor, 8 + PF steps later, on W's strides ------------
    const int ldwb = (int)prm.ldwb;
    const int64_t pairw_bytes = (int64_t)prm.Mc * prm.ldwb;
    const int wdelta_band = NLp * 128 - L * RC * ldwb;
    const int wdelta_pair = NLp * 128 + (nb - 1) * L * RC * ldwb + (int)pairw_bytes;
    int wt_m, wt_band, wt_ps, wt_row;   // line, band, pair-in-group, first flipped coarse row (class 0, k 0)
    unsigned wt_off;
    const int wt_gc = (lane >> 3) >> (prm.logL - 3);
    {
        const int ip = (lane >> 3) & ((L >> 3) - 1);
        const int v0 = -ip * LINE_UNITS;
        const int sg = floor_div(v0, NUp);
        wt_m = (v0 - sg * NUp) / LINE_UNITS;
        wt_ps = floor_div(sg, nb);
        wt_band = sg - wt_ps * nb;
        wt_row = (wt_band * L + ip * LINE_UNITS) * RC;
        wt_off = (unsigned)((wt_gc * PPG + wt_ps) * (int)pairw_bytes + (Mcp - 1 - wt_row) * ldwb +
                            (NLp - 1 - wt_m) * 128 + (7 - (lane & 7)) * 16);
    }
    char *const w_wave = static_cast<char *>(prm.W) + first_pair * pairw_bytes;
    int wj = 0, wslot = 0;
    // write the finished line of class wj (8 lanes x RC rows per instruction, 128 contiguous bytes per row)
    // wv: the finished line, read from lds0 + wslot * SLOT_BYTES + lane * 16 by the caller
    auto store_lines = [&](const vec_t (&wv)[RC]) {
        const bool pair_ok = wt_ps >= 0 && wt_ps < PPG && first_pair + (int64_t)wt_gc * PPG + wt_ps < P_end;
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const int orow = Mcp - 1 - (wt_row + wj * RC + k);
            if (pair_ok && orow < prm.Mc)
                *reinterpret_cast<vec_t *>(w_wave + (wt_off - (unsigned)((wj * RC + k) * ldwb))) = wv[k];
        }
        wslot = wslot + 1 == NSLOT ? 0 : wslot + 1;
        wj += 1;
        if (wj == LINE_UNITS) {
            wj = 0;
            wt_m += 1;
            wt_off -= 128;
            if (wt_m == NLp) {
                wt_m = 0;
                const bool last = wt_band == nb - 1;
                wt_off += last ? wdelta_pair : wdelta_band;
                wt_band = last ? 0 : wt_band + 1;
                wt_ps += last ? 1 : 0;
                wt_row = (wt_band * L + ((lane >> 3) & ((L >> 3) - 1)) * LINE_UNITS) * RC;
            }
        }
    };

    // ---- terminal edges of the forward solution (what the backward recompute of K starts from) -----------------------
    // K[MM][.] feeds the group's top lane, S values per macro-step; K[.][NN] feeds every lane at the start of its row
    // unit, R + 1 values.  Both are fetched from global memory (L2-resident: the forward kernel has just written them)
    // straight into registers ONE macro-step ahead -- asynchronously at the top of the step, waited for at its end --
    // so no LDS is spent on them.  (The first version staged whole pairs of edges in an LDS ring: 12 KB per wave, which
    // held the kernel at one wave per SIMD.)
    const int E = NNp + MMp;
    auto prefetch_edges = [&](int nu, int nband, int nps, double (&pcol)[R + 1]) {
        if (nu == 0) {
            int64_t pr = pair0 + nps;
            pr = pr < 0 ? 0 : (pr >= prm.P ? prm.P - 1 : pr);     // not-yet-started / finished lanes: any valid pair, value unused
            const double *e = prm.edges + pr * E;
            // Kf[i'][0] = K[min(MM, MMp - i')][NN], i' = i0 .. i0 + R (corner first), stored at [NNp + row - 1]; rows past
            // MM of the padded strip hold garbage, hence the min; only i' = MMp asks for K[0][NN] = 1 (fixed up after the wait)
            const int i0 = (nband * L + lam) * RC * r;
            const double *q = e + (NNp - 1);
#pragma unroll
            for (int i = 0; i < R; ++i) load_async(pcol[i], q + min(MM, MMp - (i0 + i)));
            load_async(pcol[R], q + max(min(MM, MMp - (i0 + R)), 1));
        }
    };
    // after the wait: the position whose value is the (unstored) boundary 1
    auto fix_edges = [&](int nu, int nband, double (&pcol)[R + 1]) {
        if (nu == 0 && (nband * L + lam) * RC * r + R == MMp) pcol[R] = 1.0;
    };
    // The terminal ROW K[MM][.] of the pair feeds the group's top lane, S values per macro-step: Kf[0][j'] = K[MM][NNp - j'],
    // j' = u S + i + 1, stored at index k = NNp - u S - i - 2 of the pair's edge block.  The 8 S values of a window of 8
    // macro-steps are contiguous (descending k); once per window one LDS-DMA brings the aligned span
    // [NNp - (u0 + 8) S - 2, NNp - u0 S - 1] (u0 = the window's first unit; 4 S + 1 sixteen-byte pieces) of every lane
    // group, one window ahead, into the slot the previous window has finished with -- no global load and no address arithmetic per
    // macro-step (the first version loaded the S values every step, one macro-step ahead: 16 % of the kernel).  The one
    // value outside the block (k = -1 = K[MM][0] = 1, last unit of a row) falls in a masked piece and is fixed up.
    constexpr int NPC = 4 * S + 1;   // pieces per group
    int ec_u0 = 0, ec_band = 0, ec_ps = 0, ec_fill = 0;   // window to fetch next: first unit, band, pair-in-group, slot
    auto issue_edge_chunk = [&]() {
        for (int c = 0; c * WAVE < G * NPC; ++c) {
            const int idx = c * WAVE + lane, g = idx / NPC, i = idx - g * NPC;
            int64_t pr = first_pair + (int64_t)(g < G ? g : 0) * PPG + ec_ps;
            pr = (pr < 0 || pr >= P_end) ? 0 : pr;
            const int k = NNp - (ec_u0 + LINE_UNITS) * S - 2 + 2 * i;
            if (g < G && k >= 0)
                __builtin_amdgcn_global_load_lds(prm.edges + pr * E + k, (lds_void *)(lds + NSLOT * SLOT_BYTES + ec_fill * (G * ECG) + c * 1024),
                                                 16, 0, 0);
        }
        ec_fill ^= 1;
        ec_u0 += LINE_UNITS;
        if (ec_u0 == NUp) {
            ec_u0 = 0;
            ec_band += 1;
            if (ec_band == nb) { ec_band = 0; ec_ps += 1; }
        }
    };

    double ktopR[S];
#pragma unroll
    for (int i = 0; i < S; ++i) ktopR[i] = 1.0;
    double leftR[R], botR[S], cornerR = 1.0;
    double leftF[R], botF[S], cornerF = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = 1.0; }
#pragma unroll
    for (int i = 0; i < S; ++i) { botR[i] = 1.0; botF[i] = 1.0; }

    double chk_val = 0.0;      // pending self-check result (see the end of the macro-step)
    int64_t chk_pair = -1;
    // column-edge values of the coming macro-step: requested at the top of a step (after this step's values have been
    // consumed), complete at its end
    double ncol[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) ncol[i] = 1.0;
    {
        double pcol[R + 1];
#pragma unroll
        for (int i = 0; i <= R; ++i) async_begin(pcol[i]);
        issue_edge_chunk();     // window 0 (older than the line fetches below: complete after the counted wait)
        prefetch_edges(u, band, ps, pcol);
#pragma unroll
        for (int f = 0; f < PF; ++f) issue_fetch();
        async_wait<(PF - 1) * RC>(ncol, pcol);   // the line of macro-step 0 and the edge values
        fix_edges(u, band, ncol);
    }

    for (int t = 0; t < n_steps; ++t) {
        vec_t gv[RC];
        const unsigned my_unit = rd_lane + (unsigned)(slot * SLOT_BYTES + ((u & 7) << 4));
        // the top lane's S terminal-row values of this macro-step, from the window's chunk (lane 0's unit is t modulo NUp; every
        // lane of the group reads the same address): issued without a wait, complete at the increments' lgkmcnt(0) below
        double trow_p[S], trow[S];
#pragma unroll
        for (int i = 0; i < S; ++i) async_begin(trow_p[i]);
        lds_read_f64_run<S>(trow_p, ec_base + (unsigned)(((t >> 3) & 1) * ec_slot + grp * ECG + ((7 - (t & 7)) * S + 1) * 8));
        {   // the increments (their line arrived before the previous step's closing wait) and the W line whose last
            // unit was written in the previous macro-step: one LDS round trip for both
            vec_t wv[RC];
            lds_read_rows_pair(gv, my_unit, wv, lds0 + (unsigned)(wslot * SLOT_BYTES + lane * 16));
            if (t >= LINE_UNITS) store_lines(wv);
        }
        lds_take<S>(trow, trow_p);
        if ((t & 7) == 0) issue_edge_chunk();   // the NEXT window's chunk, into the slot the previous window has finished with
        if (chk_pair >= 0) {
            atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
            chk_pair = -1;
        }

        // -- state of this lane one macro-step ahead, and its edge values (asynchronous, see prefetch_edges)
        int nu = u + 1, nband = band, nps = ps;
        if (nu == NUp) {
            nu = 0;
            nband += 1;
            if (nband == nb) { nband = 0; nps += 1; }
        }
        const int prow0 = (band * L + lam) * RC;          // first flipped coarse row of this lane

        // -- row-unit start: left boundaries.  Kr[i'][0] = 1;  Kf[i'][0] = K[min(MM, MMp - i')][NN]
        if (u == 0) {
            cornerR = 1.0;
            cornerF = ncol[0];
#pragma unroll
            for (int i = 0; i < R; ++i) { leftR[i] = 1.0; leftF[i] = ncol[i + 1]; }
        }

        // -- top rows: from the lane above, or (top lane) the band boundary / the pair's terminal row
        double topR[S], topF[S];
        {
            // what a top lane sees: Kr[0][j'] = 1 and Kf[0][j'] = the prefetched K[MM][.] values, or the band boundary
            double tbR[S], tbF[S];
#pragma unroll
            for (int i = 0; i < S; ++i) { tbR[i] = 1.0; tbF[i] = trow[S - 1 - i]; }
            if (u == NUp - 1) tbF[S - 1] = 1.0;     // K[MM][0] = 1 is not stored (only a top lane's value is used)
            if (MULTIBAND) {
                if (is_top && band > 0) {
                    // both boundary rows in one LDS round trip (a top lane exists in every wave, every macro-step)
                    lds_read_2rows<S>(tbR, tbF, bnd_r + (unsigned)(u * S) * 8u, bnd_f + (unsigned)(u * S) * 8u);
                }
            }
#pragma unroll
            for (int i = 0; i < S; ++i) {
                if (FULLWAVE) {   // lane 0 is the only top lane: wave_shr leaves its `old` operand in place there
                    if (MULTIBAND) {
                        topR[i] = dpp_shr1(botR[i], tbR[i]);
                    } else {   // lane 0 keeps the 1.0 of the persistent `old` register (see sk_wave.hip)
                        ktopR[i] = dpp_shr1(botR[i], ktopR[i]);
                        topR[i] = ktopR[i];
                    }
                    topF[i] = dpp_shr1(botF[i], tbF[i]);
                } else {
                    const double shR = dpp_shr1(botR[i], 1.0);
                    const double shF = dpp_shr1(botF[i], 1.0);
                    topR[i] = is_top ? tbR[i] : shR;
                    topF[i] = is_top ? tbF[i] : shF;
                }
            }
        }

        // -- request the next step's edge values (asynchronously, into temporaries: see async_wait), then the increment
        //    lines of macro-step t + PF, the newest operations in flight
        double pcol[R + 1];
#pragma unroll
        for (int i = 0; i <= R; ++i) async_begin(pcol[i]);
        prefetch_edges(nu, nband, nps, pcol);
        issue_fetch();

        // -- coefficients per coarse cell: a, b for Kr;  a/b, 1/b for the backward recompute of K
        double ca[RC][CW], cb[RC][CW], ca2[RC][CW], cib[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const bool row_ok = Mcp - 1 - (prow0 + k) < prm.Mc;   // padding rows carry zero increments
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                double g = vec_get<vec_t>(gv[k], CW - 1 - q);      // flipped column order inside the unit
                g = row_ok ? g : 0.0;
                if (NAIVE) {
                    ca[k][q] = fma(g, c_half, 1.0);
                    cb[k][q] = 1.0;
                    ca2[k][q] = ca[k][q];
                    cib[k][q] = 1.0;
                } else {
                    const double g2 = g * g;
                    ca[k][q] = fma(g2, c_12, fma(g, c_half, 1.0));
                    cb[k][q] = fma(g2, -c_12, 1.0);
                    cib[k][q] = fast_rcp(cb[k][q]);
                    ca2[k][q] = ca[k][q] * cib[k][q];
                }
            }
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
                double vR, vF;
                if (NAIVE) {
                    vR = fma(aboveR, a, fma(lR, a, -diagR));
                    vF = fma(aboveF, a, fma(lF, a, -diagF));
                } else {
                    vR = fma(aboveR, a, fma(lR, a, -(diagR * b)));
                    vF = fma(aboveF, a2, fma(lF, a2, -(diagF * ib)));
                }
                acc[k][q] = fma(vF, diagR, acc[k][q]);   // K[i][j] * Krev[i'][j'] for this fine cell
                diagR = lR; aboveR = vR; leftR[rr] = vR;
                diagF = lF; aboveF = vF; leftF[rr] = vF;
            }
            botR[cc] = aboveR;
            botF[cc] = aboveF;
        }
        cornerR = topR[S - 1];
        cornerF = topF[S - 1];

        if (MULTIBAND) {
            if (is_bot) {
#pragma unroll
                for (int i = 0; i < S; i += 2) {
                    d2_t vr = {botR[i], botR[i + 1]}, vf = {botF[i], botF[i + 1]};
                    lds_write_b128(bnd_r + (unsigned)(u * S + i) * 8u, vr);
                    lds_write_b128(bnd_f + (unsigned)(u * S + i) * 8u, vf);
                }
            }
        }

        // -- W: one 16-byte unit per coarse row, original column order, written over the increment unit just consumed;
        //    store_lines() sends the line to HBM once all 8 units are in
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            vec_t wu;
            if constexpr (CW == 2) { wu[0] = acc[k][1] * sc; wu[1] = acc[k][0] * sc; }
            else { wu[0] = (float)(acc[k][3] * sc); wu[1] = (float)(acc[k][2] * sc); wu[2] = (float)(acc[k][1] * sc); wu[3] = (float)(acc[k][0] * sc); }
            lds_write_b128(my_unit + k * 1024u, wu);
        }
        // -- self-check on the last flipped unit: the recomputed K on the j = 0 boundary must be 1.  The worst deviation is
        //    held in a register and sent at the top of the next macro-step together with the other memory operations: an
        //    atomic issued here would sit, with its whole latency, in front of the closing vmcnt(0).
        if (u == NUp - 1 && prm.err && ps >= 0 && ps < PPG && pair0 + ps < P_end) {
            double e = 0.0;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) e = fmax(e, fabs(leftF[rr] - 1.0));
            chk_val = e;
            chk_pair = pair0 + ps;
        }

        // -- close the step: everything requested from global memory in it (next line, W lines, edge values) has landed
        // -- close the step.  Lanes that requested nothing get garbage here and never read it: nrow matters to a top lane
        //    (which requests every step), ncol to a lane in the step right after its request
        async_wait<(PF - 1) * RC>(ncol, pcol);
        fix_edges(nu, nband, ncol);

        // -- advance
        if ((nu & 7) == 0) {
            slot += LINE_UNITS;
            if (slot >= NSLOT) slot -= NSLOT;
        }
        u = nu; band = nband; ps = nps;
    }
    for (int t = 0; t < LINE_UNITS; ++t) {   // the lines completed in the last 8 macro-steps
        vec_t wv[RC];
        lds_read_rows<63>(wv, lds0 + (unsigned)(wslot * SLOT_BYTES + lane * 16));
        store_lines(wv);
    }
    if (chk_pair >= 0)
        atomicMax(reinterpret_cast<unsigned long long *>(prm.err + chk_pair), (unsigned long long)__double_as_longlong(chk_val));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, int DY, bool NAIVE, bool MULTIBAND, bool FULLWAVE>
int launch_adj_one(const AdjParams &prm, int blocks, size_t lds_bytes, hipStream_t s) {
    auto kern = k_adj_wave<T, DY, NAIVE, MULTIBAND, FULLWAVE>;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    SK_LAUNCH(kern, dim3(blocks), dim3(WAVE * prm.wg.wpb), lds_bytes, s, prm);
    return check_launch();
}

template <typename T, int DY>
int launch_adj_dy(const AdjParams &prm, bool multiband, int blocks, size_t lds_bytes, hipStream_t s) {
    const bool full = prm.logL == 6;
    if (prm.naive) {
        if (multiband) return full ? launch_adj_one<T, DY, true, true, true>(prm, blocks, lds_bytes, s)
                                   : launch_adj_one<T, DY, true, true, false>(prm, blocks, lds_bytes, s);
        return full ? launch_adj_one<T, DY, true, false, true>(prm, blocks, lds_bytes, s)
                    : launch_adj_one<T, DY, true, false, false>(prm, blocks, lds_bytes, s);
    }
    if (multiband) return full ? launch_adj_one<T, DY, false, true, true>(prm, blocks, lds_bytes, s)
                               : launch_adj_one<T, DY, false, true, false>(prm, blocks, lds_bytes, s);
    return full ? launch_adj_one<T, DY, false, false, true>(prm, blocks, lds_bytes, s)
                : launch_adj_one<T, DY, false, false, false>(prm, blocks, lds_bytes, s);
}

}  // namespace

// SK_ERR_UNSUPPORTED when the shape / layout is not covered (the caller falls back to the stored-grid kernel).
// Requirements: dyadic 0..2 (d = 3 would spill), increment rows padded with ZEROS to whole
// 128-byte lines (ld*sizeof(T) % 128 == 0 and ld >= the padded width), W with the same row stride rule.
template <typename T>
int launch_adj_wave(const T *inc_c, int64_t ld, const Geom &g, const double *edges, T *W, int64_t ldw, double *err,
                    hipStream_t s) {
    constexpr int CW = Unit<T>::CW;
    const int DY = g.dyadic;
    // d = 3 (and d = 2 with 4-column fp32 units) needs > 256 VGPRs for the two states: stored-grid kernel
    if (DY < 0 || DY > (sizeof(T) == 8 ? 2 : 1)) return SK_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(inc_c) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return SK_ERR_UNSUPPORTED;
    const Strip st = strip_geom(g, (int)sizeof(T));   // the same decomposition as the forward kernel (it wrote `edges`)
    if (!st.ok) return SK_ERR_UNSUPPORTED;
    const int NUp = st.NUp, RC = st.RC, logL = st.logL, nb = st.nb, L = 1 << logL;
    if ((ld * sizeof(T)) % 128 || ld < (int64_t)NUp * CW) return SK_ERR_UNSUPPORTED;
    if ((ldw * sizeof(T)) % 16 || ldw < (int64_t)NUp * CW) return SK_ERR_UNSUPPORTED;
    const int S = CW << DY;
    const int G = WAVE / L;
    const bool multiband = nb > 1;

    size_t lds_bytes = (size_t)(LINE_UNITS + adj_pf(DY)) * RC * 1024 + (size_t)2 * G * adj_chunk_bytes(S);
    if (multiband) lds_bytes += (size_t)G * 2 * NUp * S * sizeof(double);
    if (lds_bytes > 160 * 1024) return SK_ERR_UNSUPPORTED;

    int waves_per_cu = (int)((160 * 1024) / lds_bytes);
    const int wpc_env = knobs().adj_wpc;
    // d = 2 (10 KB per wave): the VGPRs allow 12 resident waves per CU; launching 16 let the fourth workgroup of a CU start when
    // the first ended (13.1 ms vs 13.6 at 8), 12 with shares by age rank is better still (C4 tiles: 9.06 -> 8.44 ms)
    const int wpc_cap = wpc_env > 0 ? 16 : (DY == 2 ? 12 : 8);
    if (waves_per_cu > wpc_cap) waves_per_cu = wpc_cap;
    // this kernel waits on memory every macro-step, so every resident wave the LDS ring and the registers allow is taken (8 at
    // d = 1, 12 at d = 2); SK_ADJ_WPC overrides
    if (wpc_env > 0) waves_per_cu = waves_per_cu < wpc_env ? waves_per_cu : wpc_env;
    else if (waves_per_cu > 4) waves_per_cu &= ~3;   // whole four-wave workgroups: the same number of waves on every SIMD
    if (waves_per_cu < 1) waves_per_cu = 1;
    const int64_t max_waves = (int64_t)device_cu_count() * waves_per_cu;
    int64_t waves = (g.P + G - 1) / G;
    if (waves > max_waves) waves = max_waves;
    const int64_t pair_bytes = (int64_t)g.Mc * ld * (int64_t)sizeof(T);
    if (pair_bytes > (1LL << 29)) return SK_ERR_UNSUPPORTED;
    // shares by wave age rank when the launch fills the chip with four-wave workgroups and the largest share's span fits
    // the 32-bit buffer offsets; otherwise equal shares
    // (this kernel waits on HBM as much as on the vector unit: measured optimum 58 / 42 at d = 1, against 66 / 34 for the fused kernels)
    static constexpr double shares[5][4] = {{1, 0, 0, 0}, {1, 0, 0, 0}, {0.58, 0.42, 0, 0}, {0.53, 0.30, 0.17, 0}, {0.25, 0.25, 0.25, 0.25}};   // (three ranks: d = 2, bound by the vector unit like the fused kernels; four: not measured, equal)
    WaveGroup wg = wave_group(lds_bytes, waves, knobs().adj_wpb);
    RankSplit rs = rank_split(g.P, G, waves, max_waves, wg.wpb, device_cu_count(), knobs().adj_rank_w, shares);
    if (rs.nranks > 1 && ((int64_t)rs.cnt[0] * G + 1) * pair_bytes >= (1LL << 31)) rs = rank_split(g.P, G, waves, -1, wg.wpb, device_cu_count(), knobs().adj_rank_w);
    int64_t PPG = rs.cnt[0];
    if (rs.nranks == 1) {
        waves = (g.P + PPG * G - 1) / (PPG * G);
        if ((PPG * G + 1) * pair_bytes >= (1LL << 31)) {
            PPG = ((1LL << 31) - 1) / (G * pair_bytes) - 1;
            if (PPG < 1) return SK_ERR_UNSUPPORTED;
            waves = (g.P + PPG * G - 1) / (PPG * G);
        }
        wg = wave_group(lds_bytes, waves, knobs().adj_wpb);
        rs = rank_split(g.P, G, waves, -1, wg.wpb, device_cu_count(), knobs().adj_rank_w);
        rs.cnt[0] = (int)PPG;
        rs.base[1] = PPG * waves * G;
    }
    if (PPG > 0x3fffffff / (nb * NUp)) return SK_ERR_UNSUPPORTED;

    AdjParams prm;
    prm.inc = inc_c; prm.edges = edges; prm.W = W; prm.err = err; prm.P = g.P;
    prm.ldb = ld * (int64_t)sizeof(T); prm.ldwb = ldw * (int64_t)sizeof(T);
    prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = NUp; prm.nb = nb; prm.logL = logL; prm.PPG = (int)PPG;
    prm.n_steps = (int)(PPG * nb * NUp + (L - 1));
    prm.naive = g.naive;

    prm.wg = wg;
    prm.rs = rs;
    const int blocks = wave_group_blocks(prm.wg);
    const size_t lds_block = wave_group_lds(prm.wg);
    if (DY == 0) return launch_adj_dy<T, 0>(prm, multiband, blocks, lds_block, s);
    if (DY == 1) return launch_adj_dy<T, 1>(prm, multiband, blocks, lds_block, s);
    if constexpr (sizeof(T) == 8) return launch_adj_dy<T, 2>(prm, multiband, blocks, lds_block, s);
    return SK_ERR_UNSUPPORTED;
}

template int launch_adj_wave<double>(const double *, int64_t, const Geom &, const double *, double *, int64_t, double *,
                                     hipStream_t);
template int launch_adj_wave<float>(const float *, int64_t, const Geom &, const double *, float *, int64_t, double *,
                                    hipStream_t);

}  // namespace sk
