// sk_wave_deriv.hip -- fast solver for the signature kernel and its first and second directional
// derivatives (K, K_gamma, K_gamma_gamma) in one sweep.
//
// Same skewed row-strip mapping as sk_wave.hip (a lane owns R = 1<<DY fine rows, walks the columns in
// macro-steps of one 16-byte unit, the bottom row of a lane's block reaches the lane below by DPP), with
// three PDE states in registers and three increment arrays streamed HBM -> LDS by LDS-DMA, one whole
// 128-byte line per row and array at a time (ring slot = [array][lane/8][128 B]).
//
// Stencil: sigkernel_derivatives_Gram_cuda (reference cuda_backend.py:206-220).  With S1 = k01' + k10',
// t = k00 + k01 + k10 + K11, f1 = k00*gd + k00'*g the reference's
//   K11' = S1 - k00' + (f1 + f2 + f3 + f4)/4   collapses to
//   K11' = (gd/4)*t + (1 + g/2)*S1 + (g*gd/4)*k00 + (g*g/4 - 1)*k00'
// and likewise (td = k00' + k01' + k10' + K11', S1dd = k01'' + k10'')
//   K11'' = (gdd/4)*t + (gd/2)*td + (1 + g/2)*S1dd + (g*g/4 - 1)*k00'' + (g*gdd/4)*k00 + (g*gd/2)*k00';
// the six coefficients are formed once per coarse cell and reused by its 4^d fine cells, and the four-point sums
// t, td are built from column pair sums (k10 + K11 of one cell is k00 + k01 of the cell below): ~19 fp64
// instructions per fine cell instead of ~28.  The regrouping changes results by a few ulp only (tests compare
// with the oracle at 1e-11).
//
// Replaces: the launch at sigkernel.py:546-566 (three zero-filled (A,B,MM+2,NN+2) solution buffers in
// global memory, re-read every anti-diagonal, 1024-thread limit).
#include "sk_wave_common.h"

namespace sk {
namespace {

struct DerivParams {
    const void *inc[3];  // [P, Mc, ld] coarse increments of k, d/dgamma, d2/dgamma2
    void *out[3];        // [P] each (nullable)
    int64_t P;
    int64_t ldb;         // row stride in bytes
    int Mc, Nc;
    int NUp, nb, logL, PPG, n_steps;
    int u_f, lam_f, sel_f;
    int pf;              // prefetch distance in macro-steps (3 or 5)
    WaveGroup wg;      // workgroups of independent waves (sk_wave_common.h)
    RankSplit rs;      // pairs per wave by age rank (sk_wave_common.h); PPG / n_steps are the largest share's
};

// The increments of macro-step t+1 are requested from LDS early in macro-step t and waited for at its end: with
// one wave per SIMD nothing else hides the LDS round trip.  Issue and wait are separate asm blocks in the same loop
// iteration; the issue writes temporaries that nothing reads, the wait hands them over (see lds_read3_wait).
template <int VM, typename V>
__device__ __forceinline__ void lds_read3_issue(V (&g)[3], unsigned a) {
    asm volatile("s_waitcnt vmcnt(%4)\n\t"
                 "ds_read_b128 %0, %3\n\t"
                 "ds_read_b128 %1, %3 offset:1024\n\t"
                 "ds_read_b128 %2, %3 offset:2048"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]) : "v"(a), "n"(VM) : "memory");
}
// the wait is the instruction that turns the in-flight temporaries `t` into ordinary values `g` (outputs tied to the
// temporaries' registers): nothing may read `t` itself (tools/check_async_hazards.py lints the ISA for that)
template <typename V>
__device__ __forceinline__ void lds_read3_wait(V (&g)[3], V (&t)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "=v"(g[0]), "=v"(g[1]), "=v"(g[2]) : "0"(t[0]), "1"(t[1]), "2"(t[2]) : "memory");
}

// S consecutive doubles of each of the three states (state stride `ss` bytes): 3*S/2 ds_read_b128, ONE wait, all inside
// one asm block so that no use of a result can be scheduled before the wait
template <int S>
__device__ __forceinline__ void lds_read_states(double (&v)[3][S], unsigned addr, unsigned ss);
template <>
__device__ __forceinline__ void lds_read_states<2>(double (&v)[3][2], unsigned addr, unsigned ss) {
    d2_t t[3][1];
    asm volatile("ds_read_b128 %0, %3\n\t"
                 "ds_read_b128 %1, %4\n\t"
                 "ds_read_b128 %2, %5\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0][0]), "=&v"(t[1][0]), "=&v"(t[2][0])
                 : "v"(addr), "v"(addr + ss), "v"(addr + 2 * ss)
                 : "memory");
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 1; ++i) {
            v[s][2 * i] = t[s][i][0];
            v[s][2 * i + 1] = t[s][i][1];
        }
}
template <>
__device__ __forceinline__ void lds_read_states<4>(double (&v)[3][4], unsigned addr, unsigned ss) {
    d2_t t[3][2];
    asm volatile("ds_read_b128 %0, %6\n\t"
                 "ds_read_b128 %1, %6 offset:16\n\t"
                 "ds_read_b128 %2, %7\n\t"
                 "ds_read_b128 %3, %7 offset:16\n\t"
                 "ds_read_b128 %4, %8\n\t"
                 "ds_read_b128 %5, %8 offset:16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0][0]), "=&v"(t[0][1]), "=&v"(t[1][0]), "=&v"(t[1][1]), "=&v"(t[2][0]), "=&v"(t[2][1])
                 : "v"(addr), "v"(addr + ss), "v"(addr + 2 * ss)
                 : "memory");
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[s][2 * i] = t[s][i][0];
            v[s][2 * i + 1] = t[s][i][1];
        }
}
template <>
__device__ __forceinline__ void lds_read_states<8>(double (&v)[3][8], unsigned addr, unsigned ss) {
    d2_t t[3][4];
    asm volatile("ds_read_b128 %0, %12\n\t"
                 "ds_read_b128 %1, %12 offset:16\n\t"
                 "ds_read_b128 %2, %12 offset:32\n\t"
                 "ds_read_b128 %3, %12 offset:48\n\t"
                 "ds_read_b128 %4, %13\n\t"
                 "ds_read_b128 %5, %13 offset:16\n\t"
                 "ds_read_b128 %6, %13 offset:32\n\t"
                 "ds_read_b128 %7, %13 offset:48\n\t"
                 "ds_read_b128 %8, %14\n\t"
                 "ds_read_b128 %9, %14 offset:16\n\t"
                 "ds_read_b128 %10, %14 offset:32\n\t"
                 "ds_read_b128 %11, %14 offset:48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0][0]), "=&v"(t[0][1]), "=&v"(t[0][2]), "=&v"(t[0][3]), "=&v"(t[1][0]), "=&v"(t[1][1]), "=&v"(t[1][2]), "=&v"(t[1][3]), "=&v"(t[2][0]), "=&v"(t[2][1]), "=&v"(t[2][2]), "=&v"(t[2][3])
                 : "v"(addr), "v"(addr + ss), "v"(addr + 2 * ss)
                 : "memory");
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[s][2 * i] = t[s][i][0];
            v[s][2 * i + 1] = t[s][i][1];
        }
}

template <typename T, int DY, bool MULTIBAND, bool FULLWAVE, int PF>
__global__ __launch_bounds__(4 * WAVE) void k_deriv_wave(const DerivParams prm) {
    constexpr int CW = Unit<T>::CW;
    typedef typename Unit<T>::vec vec_t;
    constexpr int R = 1 << DY, S = CW << DY, r = 1 << DY;
    constexpr int NSLOT = LINE_UNITS + PF;
    constexpr int SLOT_BYTES = 3 * 1024;   // [array][lane/8][128 B]
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

    // ---- consumer state (see sk_wave.hip) ---------------------------------------------------------------
    int u, band, ps;
    {
        const int sig = floor_div(-lam, NUp);
        u = -lam - sig * NUp;
        ps = floor_div(sig, nb);
        band = sig - ps * nb;
    }
    const int my_uf = lam == prm.lam_f ? prm.u_f : -1;
    int PPG;               // this wave's pairs per lane group (by age rank, sk_wave_common.h), its first pair, the end of its rank
    int64_t first_pair, P_end;
    rank_share(prm.rs, wave_id, G, prm.P, PPG, first_pair, P_end);
    const int n_steps = PPG * nb * NUp + (L - 1);
    const int64_t pair0 = first_pair + (int64_t)(lane >> prm.logL) * PPG;
    const bool is_top = lam == 0, is_bot = lam == L - 1;
    int slot_off = ((((-(u & 7)) % NSLOT) + NSLOT) % NSLOT) * SLOT_BYTES;   // ring slot (byte offset) of the line being read
    const unsigned rd_lane = lds0 + (unsigned)(lane >> 3) * 128u;
    // MULTIBAND: bottom row of the previous band, [G][3][NUp*S] doubles behind the ring
    const unsigned my_bnd = lds0 + NSLOT * SLOT_BYTES + (unsigned)((lane >> prm.logL) * 3 * NUp * S) * 8u;
    const unsigned bnd_state = (unsigned)(NUp * S) * 8u;

    // ---- producer (DMA) state: one cursor, three buffer resources ----------------------------------------
    const int64_t pair_bytes = (int64_t)prm.Mc * prm.ldb;
    int64_t span = (P_end - first_pair) * pair_bytes;
    const int64_t wave_span = (int64_t)G * PPG * pair_bytes;
    span = span < wave_span ? span : wave_span;
    if (span < 0) span = 0;
    __amdgpu_buffer_rsrc_t rsrc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        rsrc[a] = __builtin_amdgcn_make_buffer_rsrc((void *)(static_cast<const char *>(prm.inc[a]) + first_pair * pair_bytes),
                                                    0, (int)span, 0x00020000);
    const int ldb = (int)prm.ldb;
    const int delta_band = L * ldb - NLp * 128;
    const int delta_pair = (int)pair_bytes - (nb - 1) * L * ldb - NLp * 128;
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
        st_off = (unsigned)((gc * PPG + ps0) * (int)pair_bytes + (st_band * L + ip * LINE_UNITS) * ldb + st_m * 128 +
                            (lane & 7) * 16);
    }
    int fj = 0, fslot_off = 0;   // class and ring slot (byte offset) of the next fetch step (uniform)
    unsigned fj_off = 0;         // fj * ldb

    auto issue_fetch = [&]() {
#pragma unroll
        for (int a = 0; a < 3; ++a)   // aux = 2: non-temporal, every line is read exactly once
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc[a], (lds_void *)(lds + fslot_off + a * 1024), 16, st_off + fj_off, 0,
                                                     0, 2);
        fslot_off = fslot_off + SLOT_BYTES == NSLOT * SLOT_BYTES ? 0 : fslot_off + SLOT_BYTES;
        fj += 1;
        fj_off += (unsigned)ldb;
        if (fj == LINE_UNITS) {
            fj = 0;
            fj_off = 0;
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

#pragma unroll
    for (int f = 0; f < PF; ++f) issue_fetch();

    vec_t gv[3];   // increments of the current macro-step
    {
        vec_t g0[3];
        lds_read3_issue<(PF - 1) * 3>(g0, rd_lane + (unsigned)(slot_off + ((u & 7) << 4)));
        lds_read3_wait(gv, g0);
    }

    for (int t = 0; t < n_steps; ++t) {
        issue_fetch();   // the line needed at macro-step t + PF

        if (u == 0) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const double bv = s == 0 ? 1.0 : 0.0;
                corner[s] = bv;
#pragma unroll
                for (int i = 0; i < R; ++i) left[s][i] = bv;
            }
        }

        // -- top row of the block for the three states
        double top[3][S];
        if (MULTIBAND) {
            // the band boundary (bottom row of the previous band) comes from LDS: all reads of the three states are
            // issued back to back and waited for once -- a wave alone on its SIMD cannot hide LDS round trips
            double tb[3][S];
            if (is_top && band > 0) {
                lds_read_states<S>(tb, my_bnd + (unsigned)(u * S) * 8u, bnd_state);
            } else {
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int i = 0; i < S; ++i) tb[s][i] = s == 0 ? 1.0 : 0.0;
            }
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    if (FULLWAVE) {   // lane 0 is the only top lane: wave_shr leaves its `old` operand (tb) in place
                        top[s][i] = dpp_shr1(bot[s][i], tb[s][i]);
                    } else {
                        const double sh = dpp_shr1(bot[s][i], s == 0 ? 1.0 : 0.0);
                        top[s][i] = is_top ? tb[s][i] : sh;
                    }
                }
        } else {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const double bv = s == 0 ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const double sh = dpp_shr1(bot[s][i], bv);
                    top[s][i] = (FULLWAVE || !is_top) ? sh : bv;   // FULLWAVE: lane 0 keeps wave_shr's `old` operand
                }
            }
        }

        // -- coefficients per coarse cell (one coarse row per lane)
        double ca[CW], cb[CW], c_t[CW], c_s[CW], c_k[CW], c_m[CW], c_tdd[CW], c_td[CW], c_kdd[CW], c_kd[CW];
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            const double g = vec_get<vec_t>(gv[0], q) * sc, gd = vec_get<vec_t>(gv[1], q) * sc,
                         gdd = vec_get<vec_t>(gv[2], q) * sc;
            const double g2 = g * g, qg = 0.25 * g;
            ca[q] = fma(g2, 1.0 / 12.0, fma(g, 0.5, 1.0));
            cb[q] = fma(g2, -1.0 / 12.0, 1.0);
            c_t[q] = 0.25 * gd;             // * t      (K')   and * td / 2 ... see below
            c_s[q] = fma(g, 0.5, 1.0);      // * S1     (K' and K'')
            c_k[q] = qg * gd;               // * k00    (K')
            c_m[q] = fma(qg, g, -1.0);      // * k00'   (K'),  * k00'' (K'')
            c_tdd[q] = 0.25 * gdd;          // * t      (K'')
            c_td[q] = 0.5 * gd;             // * td     (K'')
            c_kdd[q] = qg * gdd;            // * k00    (K'')
            c_kd[q] = (qg + qg) * gd;       // * k00'   (K'')
        }

        // -- request the increments of macro-step t + 1 (its line was fetched PF - 1 steps ago)
        vec_t gn[3];
        {
            const int nu = u + 1;
            int noff = slot_off;
            if ((nu & 7) == 0) {
                noff += LINE_UNITS * SLOT_BYTES;
                if (noff >= NSLOT * SLOT_BYTES) noff -= NSLOT * SLOT_BYTES;
            }
            lds_read3_issue<(PF - 1) * 3>(gn, rd_lane + (unsigned)(noff + ((nu & 7) << 4)));
        }

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
            double pk = diag[0] + above[0], pd = diag[1] + above[1];   // k00 + k01 of the column's first cell
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

        if (MULTIBAND) {
            if (is_bot) {
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int i = 0; i < S; i += 2) {
                        d2_t v = {bot[s][i], bot[s][i + 1]};
                        lds_write_b128(my_bnd + s * bnd_state + (unsigned)(u * S + i) * 8u, v);
                    }
            }
        }

        if (u == my_uf) {
            if (band == nb - 1 && ps >= 0 && ps < PPG && pair0 + ps < P_end) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    double v = cand[s][0];
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        double cv = cand[s][q];
                        asm volatile("" : "+v"(cv));
                        if (q == prm.sel_f) v = cv;
                    }
                    if (prm.out[s]) static_cast<T *>(prm.out[s])[pair0 + ps] = (T)v;
                }
            }
        }

        lds_read3_wait(gv, gn);

        u += 1;
        if ((u & 7) == 0) {
            slot_off += LINE_UNITS * SLOT_BYTES;
            if (slot_off >= NSLOT * SLOT_BYTES) slot_off -= NSLOT * SLOT_BYTES;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, int DY, bool MULTIBAND, bool FULLWAVE, int PF>
int launch_pf(const DerivParams &prm, int blocks, size_t lds_bytes, hipStream_t s) {
    auto kern = k_deriv_wave<T, DY, MULTIBAND, FULLWAVE, PF>;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    SK_LAUNCH(kern, dim3(blocks), dim3(WAVE * prm.wg.wpb), lds_bytes, s, prm);
    return check_launch();
}

template <typename T, int DY, bool MULTIBAND, bool FULLWAVE>
int launch_one(const DerivParams &prm, int blocks, size_t lds_bytes, hipStream_t s) {
    return launch_pf<T, DY, MULTIBAND, FULLWAVE, 3>(prm, blocks, lds_bytes, s);   // (a distance of 5 was a knob-only variant until round 5)
}

template <typename T, int DY>
int launch_dy(const DerivParams &prm, bool multiband, int blocks, size_t lds_bytes, hipStream_t s) {
    if (multiband)
        return prm.logL == 6 ? launch_one<T, DY, true, true>(prm, blocks, lds_bytes, s)
                             : launch_one<T, DY, true, false>(prm, blocks, lds_bytes, s);
    return prm.logL == 6 ? launch_one<T, DY, false, true>(prm, blocks, lds_bytes, s)
                         : launch_one<T, DY, false, false>(prm, blocks, lds_bytes, s);
}

}  // namespace

// SK_ERR_UNSUPPORTED: shape / layout outside this kernel's scope (the caller falls back to the simple kernel).
template <typename T>
int launch_deriv_wave(const T *inc, const T *inc_d, const T *inc_dd, int64_t ld, const Geom &g, T *out_k, T *out_kd,
                      T *out_kdd, hipStream_t s) {
    constexpr int CW = Unit<T>::CW;
    const int PF = 3;   // prefetch distance (macro-steps)
    const int DY = g.dyadic;
    if (DY > (sizeof(T) == 8 ? 2 : 1)) return SK_ERR_UNSUPPORTED;   // register budget: S = CW << DY columns x 3 states
    if (((reinterpret_cast<uintptr_t>(inc) | reinterpret_cast<uintptr_t>(inc_d) | reinterpret_cast<uintptr_t>(inc_dd)) & 15) ||
        ((ld * sizeof(T)) & 15))
        return SK_ERR_UNSUPPORTED;
    const int NU = (g.Nc + CW - 1) / CW;
    if ((int64_t)NU * CW > ld) return SK_ERR_UNSUPPORTED;
    const int NUp = (NU + LINE_UNITS - 1) / LINE_UNITS * LINE_UNITS;
    const int S = CW << DY;

    int logL = 3;
    while (logL < 6 && (1 << logL) < g.Mc) ++logL;
    int L = 1 << logL;
    int nb = (g.Mc + L - 1) / L;
    if (nb > 1) {
        while (L > NUp && logL > 3) { --logL; L >>= 1; }
        if (L > NUp) return SK_ERR_UNSUPPORTED;
        nb = (g.Mc + L - 1) / L;
    }
    const int G = WAVE / L;
    const bool multiband = nb > 1;

    size_t lds_bytes = (size_t)(LINE_UNITS + PF) * 3 * 1024;
    if (multiband) lds_bytes += (size_t)G * 3 * NUp * S * sizeof(double);
    if (lds_bytes > 160 * 1024) return SK_ERR_UNSUPPORTED;

    int waves_per_cu = (int)((160 * 1024) / lds_bytes);
    if (waves_per_cu > 8) waves_per_cu = 8;
    const int wpc_env = knobs().deriv_wpc;
    if (wpc_env > 0) waves_per_cu = waves_per_cu < wpc_env ? waves_per_cu : wpc_env;
    else if (waves_per_cu > 4) waves_per_cu &= ~3;
    if (waves_per_cu < 1) waves_per_cu = 1;
    const int64_t max_waves = (int64_t)device_cu_count() * waves_per_cu;
    int64_t waves = (g.P + G - 1) / G;
    if (waves > max_waves) waves = max_waves;
    const int64_t pair_bytes = (int64_t)g.Mc * ld * (int64_t)sizeof(T);
    if (pair_bytes > (1LL << 30)) return SK_ERR_UNSUPPORTED;
    // shares by wave age rank (see sk_wave_adj.hip's launcher); this kernel streams three increment arrays: the mild shares
    static constexpr double shares[5][4] = {{1, 0, 0, 0}, {1, 0, 0, 0}, {0.58, 0.42, 0, 0}, {1 / 3., 1 / 3., 1 / 3., 0}, {0.25, 0.25, 0.25, 0.25}};
    WaveGroup wg = wave_group(lds_bytes, waves, knobs().deriv_wpb);
    RankSplit rs = rank_split(g.P, G, waves, max_waves, wg.wpb, device_cu_count(), knobs().deriv_rank_w, shares);
    if (rs.nranks > 1 && (int64_t)rs.cnt[0] * G * pair_bytes >= (1LL << 31)) rs = rank_split(g.P, G, waves, -1, wg.wpb, device_cu_count(), knobs().deriv_rank_w);
    int64_t PPG = rs.cnt[0];
    if (rs.nranks == 1) {
        waves = (g.P + PPG * G - 1) / (PPG * G);
        if (PPG * G * pair_bytes >= (1LL << 31)) {
            PPG = ((1LL << 31) - 1) / (G * pair_bytes);
            if (PPG < 1) return SK_ERR_UNSUPPORTED;
            waves = (g.P + PPG * G - 1) / (PPG * G);
        }
        wg = wave_group(lds_bytes, waves, knobs().deriv_wpb);
        rs = rank_split(g.P, G, waves, -1, wg.wpb, device_cu_count(), knobs().deriv_rank_w);
        rs.cnt[0] = (int)PPG;
        rs.base[1] = PPG * waves * G;
    }
    if (PPG > 0x3fffffff / (nb * NUp)) return SK_ERR_UNSUPPORTED;

    DerivParams prm;
    prm.inc[0] = inc; prm.inc[1] = inc_d; prm.inc[2] = inc_dd;
    prm.out[0] = out_k; prm.out[1] = out_kd; prm.out[2] = out_kdd;
    prm.P = g.P; prm.ldb = ld * (int64_t)sizeof(T);
    prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = NUp; prm.nb = nb; prm.logL = logL; prm.PPG = (int)PPG;
    prm.n_steps = (int)(PPG * nb * NUp + (L - 1));
    prm.u_f = (g.Nc - 1) / CW;
    prm.lam_f = (g.Mc - 1) % L;
    prm.sel_f = (g.Nc - 1) % CW;
    prm.pf = PF;

    prm.wg = wg;
    prm.rs = rs;
    const int blocks = wave_group_blocks(prm.wg);
    const size_t lds_block = wave_group_lds(prm.wg);
    switch (DY) {
        case 0: return launch_dy<T, 0>(prm, multiband, blocks, lds_block, s);
        case 1: return launch_dy<T, 1>(prm, multiband, blocks, lds_block, s);
        default:
            if constexpr (sizeof(T) == 8) return launch_dy<T, 2>(prm, multiband, blocks, lds_block, s);
            return SK_ERR_UNSUPPORTED;
    }
}

template int launch_deriv_wave<double>(const double *, const double *, const double *, int64_t, const Geom &, double *,
                                       double *, double *, hipStream_t);
template int launch_deriv_wave<float>(const float *, const float *, const float *, int64_t, const Geom &, float *, float *,
                                      float *, hipStream_t);

}  // namespace sk
