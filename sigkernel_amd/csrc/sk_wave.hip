// sk_wave.hip -- the fast forward solver: skewed row-strip wavefront sweep, one 64-lane wavefront per
// 64/L pairs, PDE state in registers, increments streamed HBM -> LDS by LDS-DMA, one 128-byte line per row
// at a time.
//
// Mapping.  A lane owns R = RC<<DY consecutive fine rows of a "band" of L*R fine rows
// and walks along the columns in macro-steps of S = CW<<DY fine columns (CW coarse columns = one 16-byte
// unit of the increment row).  Lane l runs l macro-steps behind lane l-1, so the bottom row of lane l-1's
// block is exactly what lane l needs next: it arrives by one DPP wave_shr:1 per 32-bit half, no LDS round
// trip.  Bands of one pair, and then the next pair of the lane group, follow each other without draining
// the skew (the top lane starts the next band while the bottom lanes finish the previous one), so the
// pipeline fills once per kernel, not once per pair.
//
// Increments: the coarse matrix is read exactly once, in whole 128-byte lines.  The skew staggers the lanes
// over the 8 units of a line: at macro-step t exactly the 8 lanes with lane % 8 == t % 8 start a new line.
// So every macro-step the wave issues RC LDS-DMA instructions (buffer_load_dwordx4 ... lds, bounds-checked)
// that fetch, PF macro-steps ahead, the next line of those 8 lanes' rows: 8 DMA lanes per row, 8 rows per
// instruction, landing as [row k][lane/8][128 B] in slot (t+PF) % (8+PF) of an LDS ring.  A line is touched
// by exactly one fetch (measured: fetching 64-byte halves in separate passes costs 1.4-1.9x HBM traffic once
// the resident lines outgrow the XCD's 4 MiB L2, because L2 fills are always whole 128-byte lines), the
// ring holds (8+PF)/8 lines per row instead of 2, and the consumers' ds_read_b128 are conflict-free
// without any swizzle because the skew itself spreads a 16-lane group over 16 distinct 16-byte slots.
//
// Replaces: sigkernel_cuda / sigkernel_Gram_cuda (reference cuda_backend.py:6-49, :121-160), whose
// thread-per-row sweep re-reads the solution grid from global memory every anti-diagonal.
#include "sk_wave_common.h"

namespace sk {
namespace {

struct WaveParams {
    const void *inc;   // [P, Mc, ld] coarse increments
    void *out;         // [P] final values
    int64_t P;
    int64_t ldb;       // row stride in bytes
    int Mc, Nc;
    int NUp;           // 16-byte units per row, padded to whole lines
    int nb;            // bands per pair
    int logL;          // lanes per pair group = 1 << logL (>= 8)
    int PPG;           // pairs per lane group
    int n_steps;       // macro-steps each wave sweeps (incl. drain)
    int u_f, lam_f, sel_f;  // where K[MM][NN] lives: unit, lane-in-group, k_f*CW + cw_f inside the block
    int naive;
    double *edges;     // nullable [P, NNp + MMp]: K[MM][1..NNp] then K[1..MMp][NN] (EDGES variant; padded strip sizes)
    int k_f;           // coarse row inside the lane's block that holds the pair's last row
    WaveGroup wg;      // workgroups of independent waves (sk_wave_common.h)
    RankSplit rs;      // pairs per wave by age rank (sk_wave_common.h); PPG / n_steps are the largest share's
};

// ------------------------------------------------------------------------------------------------
template <typename T, int DY, bool NAIVE, bool MULTIBAND, bool FULLWAVE, bool EDGES, int PF>
__global__ __launch_bounds__(4 * WAVE) void k_fwd_wave(const WaveParams prm) {
    constexpr int CW = Unit<T>::CW;
    typedef typename Unit<T>::vec vec_t;
    constexpr int RC = Tile<DY>::RC, R = Tile<DY>::R, S = CW << DY, r = 1 << DY;
    constexpr int NSLOT = LINE_UNITS + PF;    // ring slots; one slot = the next line of 8 lanes' rows
    constexpr int SLOT_BYTES = RC * 1024;     // [k][lane/8][128 B]
    extern __shared__ __attribute__((aligned(16))) char lds_block[];
    char *lds;
    const int64_t wave_id = wave_slot(prm.wg, lds_block, lds);
    if (wave_id < 0) return;
    const unsigned lds0 = lds_offset(lds);

    const int lane = threadIdx.x & (WAVE - 1);
    const int L = 1 << prm.logL, G = WAVE >> prm.logL;
    const int lam = lane & (L - 1);
    const int NUp = prm.NUp, nb = prm.nb, NLp = NUp / LINE_UNITS;
    const double sc = 1.0 / (double)(1 << (2 * DY));  // 4^-d
    const double c_half = 0.5 * sc, c_12 = sc * sc / 12.0;

    // ---- consumer state: virtual unit v = t - lam, row unit sig = floor(v / NUp), u = v mod NUp ----------
    int u, band, ps;   // ps = pair index inside the lane group
    {
        const int sig = floor_div(-lam, NUp);
        u = -lam - sig * NUp;
        ps = floor_div(sig, nb);
        band = sig - ps * nb;
    }
    const int my_uf = lam == prm.lam_f ? prm.u_f : -1;   // the unit at which this lane holds K[MM][NN] (if ever)
    int PPG;               // this wave's pairs per lane group (by age rank, sk_wave_common.h), its first pair, the end of its rank
    int64_t first_pair, P_end;
    rank_share(prm.rs, wave_id, G, prm.P, PPG, first_pair, P_end);
    const int n_steps = PPG * nb * NUp + (L - 1);
    const int64_t pair0 = first_pair + (int64_t)(lane >> prm.logL) * PPG;
    const bool is_top = lam == 0, is_bot = lam == L - 1;
    // ring slot of the line this lane is reading: the line it started at macro-step ts sits in slot ts % NSLOT
    int slot = (((-(u & 7)) % NSLOT) + NSLOT) % NSLOT;
    const unsigned rd_lane = lds0 + (unsigned)(lane >> 3) * 128u;
    // MULTIBAND: bottom row of the previous band, [G][NUp*S] doubles behind the ring
    const unsigned my_bnd = lds0 + NSLOT * SLOT_BYTES + (unsigned)((lane >> prm.logL) * NUp * S) * 8u;
    // EDGES: the terminal row and column of every pair (what the adjoint kernel's backward recompute of K starts
    // from) go straight from registers to global memory, 32..64 bytes per lane and macro-step, in the padded layout
    // [K[MM][1..NNp]] [K[1..MMp][NN]].  The values are held for one macro-step and stored right after the next
    // step's DMA wait, so that the stores have a whole step to be acknowledged before the following wait (loads and
    // stores share vmcnt on gfx9: a store still in flight at the wait costs a step of prefetch distance).
    const int EP = EDGES ? (NUp * S + nb * L * R) : 0;   // doubles per pair
    // The column values are this lane's `left` state and, when the pair's last coarse row is the last row of the lane's
    // block, the row values are its `bot` state: both survive untouched until the next step's sweep, so they are stored
    // from there.  Only when the last row sits higher in the block (k_f < RC-1) is a copy held in `erow`.
    double erow[S];
    int erow_at = -1, ecol_at = -1;                       // element offsets inside the pair's block, -1 = nothing to store
    int64_t e_pair = 0;
    const bool row_in_bot = prm.k_f == RC - 1;             // uniform

    // ---- producer (DMA) state ---------------------------------------------------------------------------
    // Fetch step f = 8q + j serves the 8 consumer lanes lc = j + 8*(lane/8); all of them are about to start
    // line number n = q - i' (i' = (lane/8) mod (L/8)) of their virtual row stream, so one (band, line)
    // cursor per DMA lane covers all 8 classes j; only the row offset j*RC*ldb differs, and that is uniform.
    // Addresses are 32-bit offsets into a per-wave buffer resource (base = first pair of this wave): rows past
    // the end of a pair, pairs past P and the not-yet-started lanes of the pipeline fall outside num_records
    // (or into a neighbouring pair) and the bounds-checked buffer load returns without touching memory.
    const int64_t pair_bytes = (int64_t)prm.Mc * prm.ldb;
    int64_t span = (P_end - first_pair) * pair_bytes;
    const int64_t wave_span = (int64_t)G * PPG * pair_bytes;
    span = span < wave_span ? span : wave_span;
    if (span < 0) span = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(static_cast<const char *>(prm.inc) + first_pair * pair_bytes), 0, (int)span, 0x00020000);
    const int ldb = (int)prm.ldb;
    const int delta_band = L * RC * ldb - NLp * 128;                                  // next band of the same pair
    const int delta_pair = (int)pair_bytes - (nb - 1) * L * RC * ldb - NLp * 128;    // first band of the next pair
    int st_m, st_band;
    unsigned st_off;
    {
        const int ip = (lane >> 3) & ((L >> 3) - 1);   // i'
        const int gc = (lane >> 3) >> (prm.logL - 3);    // lane group of the consumers this DMA lane serves
        const int v0 = -ip * LINE_UNITS;
        const int sg = floor_div(v0, NUp);
        st_m = (v0 - sg * NUp) / LINE_UNITS;
        const int ps0 = floor_div(sg, nb);
        st_band = sg - ps0 * nb;
        st_off = (unsigned)((gc * PPG + ps0) * (int)pair_bytes + (st_band * L + ip * LINE_UNITS) * RC * ldb +
                            st_m * 128 + (lane & 7) * 16);
    }
    int fj = 0, fslot = 0;   // class and ring slot of the next fetch step (uniform)

    auto issue_fetch = [&]() {
#pragma unroll
        for (int k = 0; k < RC; ++k)   // aux = 2, non-temporal: the line is read exactly once
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + fslot * SLOT_BYTES + k * 1024), 16,
                                                     st_off + (unsigned)((fj * RC + k) * ldb), 0, 0, 2);
        fslot = fslot + 1 == NSLOT ? 0 : fslot + 1;
        fj += 1;
        if (fj == LINE_UNITS) {   // all 8 classes have their line number n: move the cursor to n + 1
            fj = 0;
            st_m += 1;
            st_off += 128;
            if (st_m == NLp) {
                st_m = 0;
                const bool last = st_band == nb - 1;
                st_off += last ? delta_pair : delta_band;
                st_band = last ? 0 : st_band + 1;
            }
        }
    };

    double left[R], bot[S], ktop[S], corner = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) left[i] = 1.0;
#pragma unroll
    for (int i = 0; i < S; ++i) { bot[i] = 1.0; ktop[i] = 1.0; }

    // prologue: the lines needed at macro-steps 0 .. PF-1
#pragma unroll
    for (int f = 0; f < PF; ++f) issue_fetch();

    for (int t = 0; t < n_steps; ++t) {
        issue_fetch();   // the line needed at macro-step t + PF
        // -- increments of this macro-step: RC rows x CW coarse columns (waits for the fetch of step t)
        vec_t gv[RC];
        lds_read_rows<PF * RC>(gv, rd_lane + (unsigned)(slot * SLOT_BYTES + ((u & 7) << 4)));

        if (EDGES) {   // the edge values produced in the previous macro-step (before anything touches left / bot)
            double *const ep = prm.edges + e_pair * EP;
            if (erow_at >= 0) {
#pragma unroll
                for (int cc = 0; cc < S; cc += 2) {
                    d2_t v = {row_in_bot ? bot[cc] : erow[cc], row_in_bot ? bot[cc + 1] : erow[cc + 1]};
                    *reinterpret_cast<d2_t *>(ep + erow_at + cc) = v;
                }
            }
            if (ecol_at >= 0) {
#pragma unroll
                for (int rr = 0; rr < R; rr += 2) {
                    d2_t v = {left[rr], left[rr + 1]};
                    *reinterpret_cast<d2_t *>(ep + ecol_at + rr) = v;
                }
            }
        }

        // -- row-unit start: left boundary K[i][0] = 1
        if (u == 0) {
            asm volatile("");   // a real branch (if-converted: ten v_cndmask in every macro-step instead of five moves for one lane)
            corner = 1.0;
#pragma unroll
            for (int i = 0; i < R; ++i) left[i] = 1.0;
        }

        // -- top row of the block: bottom row of the lane above (previous macro-step), or the band boundary
        double top[S];
        if (MULTIBAND) {
            double tb[S];
            if (is_top && band > 0) {
                lds_read_row1<S>(tb, my_bnd + (unsigned)(u * S) * 8u);   // one LDS round trip
            } else {
#pragma unroll
                for (int i = 0; i < S; ++i) tb[i] = 1.0;
            }
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const double sh = dpp_shr1(bot[i], 1.0);
                top[i] = is_top ? tb[i] : sh;
            }
        } else if (FULLWAVE) {
            // one pair per wave: lane 0 is the only top lane and wave_shr leaves its `old` operand in place there.  The
            // old operand is the persistent register ktop[i], which nothing else writes: lane 0 keeps the 1.0 it was
            // initialised with, and no constant has to be re-materialised in the destination every macro-step.
#pragma unroll
            for (int i = 0; i < S; ++i) {
                ktop[i] = dpp_shr1(bot[i], ktop[i]);
                top[i] = ktop[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const double sh = dpp_shr1(bot[i], 1.0);
                top[i] = is_top ? 1.0 : sh;
            }
        }

        // -- coefficients per coarse cell
        double ca[RC][CW], cbm[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const double g = vec_get<vec_t>(gv[k], q);
                if (NAIVE) {
                    ca[k][q] = fma(g, c_half, 1.0);
                    cbm[k][q] = 1.0;
                } else {
                    const double g2 = g * g;
                    ca[k][q] = fma(g2, c_12, fma(g, c_half, 1.0));
                    cbm[k][q] = fma(g2, -c_12, 1.0);
                }
            }

        // -- sweep the R x S block column by column: K11 = a (K10 + K01) - b K00 as fma(K01, a, fma(K10, a, -b K00))
        double cand[RC][CW];
        double rowv[RC][S];   // EDGES: K on the last fine row of each coarse row of the block
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
            double above = top[cc];                        // K[i0][j+1]
            double diag = cc == 0 ? corner : top[cc - 1];  // K[i0][j]
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const double a = ca[rr >> DY][cc >> DY], b = cbm[rr >> DY][cc >> DY];
                const double k10 = left[rr];
                double v;
                if (NAIVE) v = fma(above, a, fma(k10, a, -diag));
                else v = fma(above, a, fma(k10, a, -(diag * b)));
                diag = k10;
                above = v;
                left[rr] = v;
                if ((rr & (r - 1)) == r - 1 && (cc & (r - 1)) == r - 1) cand[rr >> DY][cc >> DY] = v;
                if (EDGES && (rr & (r - 1)) == r - 1) rowv[rr >> DY][cc] = v;
            }
            bot[cc] = above;
        }
        corner = top[S - 1];

        if (MULTIBAND) {
            if (is_bot) {
#pragma unroll
                for (int i = 0; i < S; i += 2) {
                    d2_t v = {bot[i], bot[i + 1]};
                    lds_write_b128(my_bnd + (unsigned)(u * S + i) * 8u, v);
                }
            }
        }

        // -- terminal row and column of the pair: note where they go, the next macro-step stores them (see above).  The
        //    column relies on the padding columns of the last unit being zero (K is constant along zero increments).
        if (EDGES) {
            const bool pair_ok = ps >= 0 && ps < PPG && pair0 + ps < P_end;
            e_pair = pair0 + ps;                       // only read where one of the offsets below is set
            erow_at = (pair_ok && lam == prm.lam_f && band == nb - 1) ? u * S : -1;
            ecol_at = (pair_ok && u == prm.u_f) ? NUp * S + (band * L + lam) * R : -1;
            if (!row_in_bot) {                         // uniform branch, unconditional copies
#pragma unroll
                for (int kk = 0; kk < RC - 1; ++kk)
                    if (kk == prm.k_f) {
#pragma unroll
                        for (int cc = 0; cc < S; ++cc) erow[cc] = rowv[kk][cc];
                    }
            }
        }

        // -- K[MM][NN] of a pair: one lane, once per pair (the asm keeps the select chain inside the branch)
        if (u == my_uf) {
            if (band == nb - 1 && ps >= 0 && ps < PPG && pair0 + ps < P_end) {
                double v = cand[0][0];
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        double cv = cand[k][q];
                        asm volatile("" : "+v"(cv));
                        if (k * CW + q == prm.sel_f) v = cv;
                    }
                static_cast<T *>(prm.out)[pair0 + ps] = (T)v;
            }
        }

        // -- advance
        u += 1;
        if ((u & 7) == 0) {   // next line of this lane: it was fetched into the slot 8 steps further
            slot += LINE_UNITS;
            if (slot >= NSLOT) slot -= NSLOT;
            if (u == NUp) {
                u = 0;
                band += 1;
                if (band == nb) {
                    band = 0;
                    ps += 1;
                }
            }
        }
    }
    if (EDGES) {   // the values of the very last macro-step
        double *const ep = prm.edges + e_pair * EP;
        if (erow_at >= 0) {
#pragma unroll
            for (int cc = 0; cc < S; ++cc) ep[erow_at + cc] = row_in_bot ? bot[cc] : erow[cc];
        }
        if (ecol_at >= 0) {
#pragma unroll
            for (int rr = 0; rr < R; ++rr) ep[ecol_at + rr] = left[rr];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, int DY, bool NAIVE, bool MULTIBAND, bool FULLWAVE, bool EDGES, int PF>
int launch_one(const WaveParams &prm, int blocks, size_t lds_bytes, hipStream_t s) {
    auto kern = k_fwd_wave<T, DY, NAIVE, MULTIBAND, FULLWAVE, EDGES, PF>;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    SK_LAUNCH(kern, dim3(blocks), dim3(WAVE * prm.wg.wpb), lds_bytes, s, prm);
    return check_launch();
}

// Tuning knob: SK_WAVE_WPC = cap on resident waves per CU.  The prefetch distance is 2 macro-steps (3 and 4 were built as variants of
// their own until round 5 -- 128 kernel instances only an environment knob could reach; 2 measured best, DESIGN 4.1 of round 1).

template <typename T, int DY, bool NAIVE, int PF>
int launch_nv(const WaveParams &prm, bool multiband, int blocks, size_t lds_bytes, hipStream_t s) {
    const bool full = prm.logL == 6;
    if (prm.edges) {   // the adjoint's forward pass (strip edges exist up to dyadic 2 in fp64, 1 in fp32: sk_strip_edges_bytes)
        if constexpr (DY <= (sizeof(T) == 8 ? 2 : 1)) {
            if (multiband) return launch_one<T, DY, NAIVE, true, false, true, 2>(prm, blocks, lds_bytes, s);
            return full ? launch_one<T, DY, NAIVE, false, true, true, 2>(prm, blocks, lds_bytes, s)
                        : launch_one<T, DY, NAIVE, false, false, true, 2>(prm, blocks, lds_bytes, s);
        } else {
            return SK_ERR_UNSUPPORTED;
        }
    }
    if (multiband) return launch_one<T, DY, NAIVE, true, false, false, PF>(prm, blocks, lds_bytes, s);
    return full ? launch_one<T, DY, NAIVE, false, true, false, PF>(prm, blocks, lds_bytes, s)
                : launch_one<T, DY, NAIVE, false, false, false, PF>(prm, blocks, lds_bytes, s);
}

template <typename T, int DY>
int launch_dy(const WaveParams &prm, bool multiband, int pf, int blocks, size_t lds_bytes, hipStream_t s) {
    (void)pf;
    return prm.naive ? launch_nv<T, DY, true, 2>(prm, multiband, blocks, lds_bytes, s)
                     : launch_nv<T, DY, false, 2>(prm, multiband, blocks, lds_bytes, s);
}

}  // namespace

// Returns SK_ERR_UNSUPPORTED when the shape / layout is outside what this kernel handles; the caller
// then falls back to the simple kernel.
template <typename T>
int launch_fwd_wave(const T *inc_c, int64_t ld, const Geom &g, T *out_final, double *strip_edges, hipStream_t s) {
    constexpr int CW = Unit<T>::CW;
    const int PF = 2;
    const int DY = g.dyadic;
    if (DY > 3) return SK_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(inc_c) & 15) || ((ld * sizeof(T)) & 15)) return SK_ERR_UNSUPPORTED;
    const int NU = (g.Nc + CW - 1) / CW;
    if ((int64_t)NU * CW > ld) return SK_ERR_UNSUPPORTED;  // the last unit must stay inside the row
    // lanes per pair: the smallest power of two (>= 8) whose band covers all rows; else full waves and bands
    const Strip st = strip_geom(g, (int)sizeof(T));
    if (!st.ok) return SK_ERR_UNSUPPORTED;
    const int NUp = st.NUp, RC = st.RC, logL = st.logL, nb = st.nb, L = 1 << logL;
    const int S = CW << DY;
    const int G = WAVE / L;
    const bool multiband = nb > 1;

    size_t lds_bytes = (size_t)(LINE_UNITS + PF) * RC * 1024;
    if (multiband) lds_bytes += (size_t)G * NUp * S * sizeof(double);
    if (lds_bytes > 160 * 1024) return SK_ERR_UNSUPPORTED;

    // persistent waves: enough of them to fill the chip, each streaming PPG pairs per lane group
    int waves_per_cu = (int)((160 * 1024) / lds_bytes);
    const int wpc_env = knobs().wave_wpc;
    // d = 2 (10 KB of LDS per wave): 16 waves/CU 3.09 ms vs 3.43 ms at 8 on a C4 tile; d = 3 is better off with 8 (2.24 vs 2.36 ms)
    const int wpc_cap = (wpc_env > 0 || DY == 2) ? 16 : 8;
    if (waves_per_cu > wpc_cap) waves_per_cu = wpc_cap;
    // persistent waves all carry the same work: an uneven count per SIMD (5, 6, 7 waves on 4 SIMDs) makes the
    // fullest SIMD the critical path (measured: 5 waves/CU is 27 % slower than 4); an explicit override is taken as is
    if (wpc_env > 0) waves_per_cu = waves_per_cu < wpc_env ? waves_per_cu : wpc_env;
    else if (waves_per_cu > 4) waves_per_cu &= ~3;
    if (waves_per_cu < 1) waves_per_cu = 1;
    const int64_t max_waves = (int64_t)device_cu_count() * waves_per_cu;
    int64_t waves = (g.P + G - 1) / G;   // one pair per lane group at least
    if (waves > max_waves) waves = max_waves;
    int64_t PPG = (g.P + waves * G - 1) / (waves * G);
    waves = (g.P + PPG * G - 1) / (PPG * G);
    if (PPG > 0x3fffffff / (nb * NUp)) return SK_ERR_UNSUPPORTED;
    // the DMA addresses are 32-bit offsets from the wave's first pair: keep a wave's span below 2 GiB
    const int64_t pair_bytes = (int64_t)g.Mc * ld * (int64_t)sizeof(T);
    if (pair_bytes > (1LL << 30)) return SK_ERR_UNSUPPORTED;
    if (PPG * G * pair_bytes >= (1LL << 31)) {
        PPG = ((1LL << 31) - 1) / (G * pair_bytes);
        if (PPG < 1) return SK_ERR_UNSUPPORTED;
        waves = (g.P + PPG * G - 1) / (PPG * G);
    }

    WaveParams prm;
    prm.inc = inc_c; prm.out = out_final; prm.P = g.P; prm.ldb = ld * (int64_t)sizeof(T);
    prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = NUp; prm.nb = nb; prm.logL = logL; prm.PPG = (int)PPG;
    prm.n_steps = (int)(PPG * nb * NUp + (L - 1));
    prm.u_f = (g.Nc - 1) / CW;
    prm.lam_f = ((g.Mc - 1) / RC) % L;
    prm.sel_f = ((g.Mc - 1) % RC) * CW + (g.Nc - 1) % CW;
    prm.naive = g.naive;
    prm.edges = strip_edges;
    prm.k_f = (g.Mc - 1) % RC;

    // Workgroups.  With equal shares the HBM-bound streaming sweep gains nothing from even SIMD loads and is a few per cent
    // faster as single-wave workgroups (per 131072 pairs of 127 x 127: d = 0 2.70 vs 2.80 ms, with strip edges at d = 1 3.77
    // vs 3.98 ms).  Where the launch fills the chip with TWO four-wave workgroups per CU, shares by wave age rank
    // (sk_wave_common.h) beat that: 58 / 42 % 2.87 -> 2.67 ms, with strip edges 54 / 46 % 3.68 -> 3.57 ms (other residencies:
    // not measured, single-wave workgroups as before).  SK_WAVE_WPB=1 restores those everywhere.
    prm.wg = wave_group(lds_bytes, waves, knobs().wave_wpb, 1);
    prm.rs = rank_split(g.P, G, waves, -1, prm.wg.wpb, device_cu_count(), knobs().wave_rank_w);   // equal shares
    prm.rs.cnt[0] = (int)PPG;
    prm.rs.base[1] = PPG * waves * G;
    if ((knobs().wave_wpb <= 0 || knobs().wave_wpb == 4) && lds_bytes * 4 <= 160 * 1024) {
        static constexpr double plain[5][4] = {{1, 0, 0, 0}, {1, 0, 0, 0}, {0.58, 0.42, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0}};
        static constexpr double edged[5][4] = {{1, 0, 0, 0}, {1, 0, 0, 0}, {0.54, 0.46, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0}};
        const int64_t full = (g.P + G - 1) / G < max_waves ? (g.P + G - 1) / G : max_waves;
        const RankSplit rs = rank_split(g.P, G, full, max_waves, 4, device_cu_count(), knobs().wave_rank_w, strip_edges ? edged : plain);
        if (rs.nranks == 2 && (int64_t)rs.cnt[0] * G * pair_bytes < (1LL << 31)) {
            prm.rs = rs;
            waves = full;
            prm.wg = wave_group(lds_bytes, waves, knobs().wave_wpb, 4);
        }
    }
    const int blocks = wave_group_blocks(prm.wg);
    const size_t lds_block = wave_group_lds(prm.wg);
    switch (DY) {
        case 0: return launch_dy<T, 0>(prm, multiband, PF, blocks, lds_block, s);
        case 1: return launch_dy<T, 1>(prm, multiband, PF, blocks, lds_block, s);
        case 2: return launch_dy<T, 2>(prm, multiband, PF, blocks, lds_block, s);
        default: return launch_dy<T, 3>(prm, multiband, PF, blocks, lds_block, s);
    }
}

template int launch_fwd_wave<double>(const double *, int64_t, const Geom &, double *, double *, hipStream_t);
template int launch_fwd_wave<float>(const float *, int64_t, const Geom &, float *, double *, hipStream_t);

}  // namespace sk
