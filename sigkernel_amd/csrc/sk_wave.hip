// sk_wave.hip -- the fast forward solver: skewed row-strip wavefront sweep, one 64-lane wavefront per
// 64/L pairs, PDE state in registers, increments streamed HBM -> LDS by LDS-DMA.
//
// Mapping.  A lane owns R = RC<<DY consecutive fine rows of a "band" of
// L*R fine rows and walks along the columns in macro-steps of S = CW<<DY fine columns (CW coarse
// columns = one 16-byte unit of the increment row).  Lane l runs l macro-steps behind lane l-1, so the
// bottom row of lane l-1's block is exactly what lane l needs next: it arrives by one DPP wave_shr:1
// per 32-bit half, no LDS round trip.  Bands of one pair, and then the next pair of the lane group,
// follow each other without draining the skew (the top lane starts the next band while the bottom lanes
// finish the previous one), so the pipeline fills once per kernel, not once per pair.
//
// Increments: the coarse matrix is read exactly once.  Each chunk (4 units = 64 B per row) is fetched
// with global_load_lds_dwordx4: 4 adjacent lanes fetch one row segment, and the (row, unit) each DMA
// lane fetches is chosen so that the linear LDS image is bank-conflict-free for the consumers'
// ds_read_b128 (slot = (k*64 + lane)*4 + ((unit + (lane>>2)) & 3)).  The fetch for lane l is skewed by l
// units, so a 2- or 3-deep ring of chunks is all the LDS the sweep needs.
//
// Replaces: sigkernel_cuda / sigkernel_Gram_cuda (reference cuda_backend.py:6-49, :121-160), whose
// thread-per-row sweep re-reads the solution grid from global memory every anti-diagonal.
#include <cstdlib>

#include "sk_internal.h"

namespace sk {
namespace {

constexpr int WAVE = 64;
constexpr int UPC = 4;  // 16-byte units per chunk and row

struct WaveParams {
    const void *inc;   // [P, Mc, ld] coarse increments
    void *out;         // [P] final values
    int64_t P;
    int64_t ldb;       // row stride in bytes
    int Mc, Nc;
    int NUp;           // 16-byte units per row, padded to a multiple of UPC
    int nb;            // bands per pair
    int logL;          // lanes per pair group = 1 << logL
    int PPG;           // pairs per lane group
    int n_chunks;      // chunks each wave sweeps (incl. drain)
    int u_f, lam_f, sel_f;  // where K[MM][NN] lives: unit, lane-in-group, k_f*CW + cw_f inside the block
    int naive;
};

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ double dpp_shr1(double v, double fill) {
    // lane l receives lane l-1's value; lane 0 keeps `fill`
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int floor_div(int a, int b) {  // b > 0
    int q = a / b;
    return (a % b < 0) ? q - 1 : q;
}

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename T> struct Unit;  // one 16-byte unit of increments
template <> struct Unit<double> { static constexpr int CW = 2; typedef d2_t vec; };
template <> struct Unit<float> { static constexpr int CW = 4; typedef f4_t vec; };

template <typename V> __device__ __forceinline__ double vec_get(const V &v, int i) { return (double)v[i]; }

// All LDS traffic of the sweep goes through inline asm.  hipcc cannot tell that a ds_read does not alias
// an LDS-DMA still in flight and would drain the whole prefetch ring with s_waitcnt vmcnt(0) before every
// read; here the DMA queue is counted by hand (vmcnt(N) = chunks still allowed in flight) and the asm
// block itself waits for its own reads (lgkmcnt(0)) before any output is consumed.
template <int VM, typename V>
__device__ __forceinline__ void lds_read_chunk(V (&g)[4], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile("s_waitcnt vmcnt(%8)\n\t"
                 "ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %5\n\t"
                 "ds_read_b128 %2, %6\n\t"
                 "ds_read_b128 %3, %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(VM)
                 : "memory");
}
template <int VM, typename V>
__device__ __forceinline__ void lds_read_chunk(V (&g)[8], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile("s_waitcnt vmcnt(%12)\n\t"
                 "ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %9\n\t"
                 "ds_read_b128 %2, %10\n\t"
                 "ds_read_b128 %3, %11\n\t"
                 "ds_read_b128 %4, %8 offset:4096\n\t"
                 "ds_read_b128 %5, %9 offset:4096\n\t"
                 "ds_read_b128 %6, %10 offset:4096\n\t"
                 "ds_read_b128 %7, %11 offset:4096\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(VM)
                 : "memory");
}
template <int VM, typename V>
__device__ __forceinline__ void lds_read_chunk(V (&g)[16], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile("s_waitcnt vmcnt(%20)\n\t"
                 "ds_read_b128 %0, %16\n\t"
                 "ds_read_b128 %1, %17\n\t"
                 "ds_read_b128 %2, %18\n\t"
                 "ds_read_b128 %3, %19\n\t"
                 "ds_read_b128 %4, %16 offset:4096\n\t"
                 "ds_read_b128 %5, %17 offset:4096\n\t"
                 "ds_read_b128 %6, %18 offset:4096\n\t"
                 "ds_read_b128 %7, %19 offset:4096\n\t"
                 "ds_read_b128 %8, %16 offset:8192\n\t"
                 "ds_read_b128 %9, %17 offset:8192\n\t"
                 "ds_read_b128 %10, %18 offset:8192\n\t"
                 "ds_read_b128 %11, %19 offset:8192\n\t"
                 "ds_read_b128 %12, %16 offset:12288\n\t"
                 "ds_read_b128 %13, %17 offset:12288\n\t"
                 "ds_read_b128 %14, %18 offset:12288\n\t"
                 "ds_read_b128 %15, %19 offset:12288\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]),
                   "=&v"(g[8]), "=&v"(g[9]), "=&v"(g[10]), "=&v"(g[11]), "=&v"(g[12]), "=&v"(g[13]), "=&v"(g[14]),
                   "=&v"(g[15])
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(VM)
                 : "memory");
}
__device__ __forceinline__ double lds_read_f64(unsigned addr) {
    double v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write_f64(unsigned addr, double v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned lds_offset(const void *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}

template <int DY> struct Tile {
    static constexpr int RC = DY == 0 ? 4 : DY == 1 ? 2 : 1;  // coarse rows per lane
    static constexpr int R = RC << DY;                         // fine rows per lane
};

// ------------------------------------------------------------------------------------------------
template <typename T, int DY, bool NAIVE, bool MULTIBAND, bool FULLWAVE, int NBUF>
__global__ __launch_bounds__(WAVE) void k_fwd_wave(const WaveParams prm) {
    constexpr int CW = Unit<T>::CW;
    typedef typename Unit<T>::vec vec_t;
    constexpr int RC = Tile<DY>::RC, R = Tile<DY>::R, S = CW << DY, r = 1 << DY;
    constexpr int IPC = RC * UPC;             // DMA instructions per chunk
    constexpr int CHUNK_BYTES = IPC * 1024;   // 64 lanes x 16 B each
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned lds0 = lds_offset(lds);

    const int lane = threadIdx.x;
    const int L = 1 << prm.logL, G = WAVE >> prm.logL;
    const int lam = lane & (L - 1);
    const int NUp = prm.NUp, nb = prm.nb;
    const double sc = 1.0 / (double)(1 << (2 * DY));  // 4^-d
    const double c_half = 0.5 * sc, c_12 = sc * sc / 12.0;

    // ---- consumer state: virtual unit v = t - lam, row unit sigma = floor(v / NUp), pair = sigma / nb ----
    int u, band, ps;
    {
        const int sig = floor_div(-lam, NUp);
        u = -lam - sig * NUp;
        ps = floor_div(sig, nb);
        band = sig - ps * nb;
    }
    const int64_t pair0 = ((int64_t)blockIdx.x * G + (lane >> prm.logL)) * prm.PPG;
    const bool is_top = lam == 0, is_bot = lam == L - 1;
    // MULTIBAND: bottom row of the previous band, [G][NUp*S] doubles behind the chunk ring
    const unsigned my_bnd = lds0 + NBUF * CHUNK_BYTES + (unsigned)((lane >> prm.logL) * NUp * S) * 8u;

    // ---- producer (DMA) state: this lane fetches unit w_d of the chunk for 4 consumer lanes -------------
    // Addresses are 32-bit offsets into a per-wave buffer resource (base = first pair of this wave): rows past
    // the end of a pair, pairs past P and the not-yet-started lanes of the pipeline fall outside num_records
    // (or into a neighbouring pair) and the bounds-checked buffer load returns without touching memory.
    const int w_d = ((lane & 3) - (lane >> 4)) & 3;
    const int64_t pair_bytes = (int64_t)prm.Mc * prm.ldb;
    const int64_t first_pair = (int64_t)blockIdx.x * G * prm.PPG;
    int64_t span = ((int64_t)prm.P - first_pair) * pair_bytes;
    const int64_t wave_span = (int64_t)G * prm.PPG * pair_bytes;
    span = span < wave_span ? span : wave_span;
    if (span < 0) span = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(static_cast<const char *>(prm.inc) + first_pair * pair_bytes), 0, (int)span, 0x00020000);
    const int ldb = (int)prm.ldb;
    const int delta_band = L * RC * ldb - NUp * 16;                      // next band of the same pair
    const int delta_pair = (int)pair_bytes - (nb - 1) * L * RC * ldb - NUp * 16;  // first band of the next pair
    int d_u[4], d_band[4];
    unsigned d_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lc = j * 16 + (lane >> 2);
        const int lamc = lc & (L - 1);
        const int v0 = w_d - lamc;
        const int sig = floor_div(v0, NUp);
        d_u[j] = v0 - sig * NUp;
        const int ps0 = floor_div(sig, nb);
        d_band[j] = sig - ps0 * nb;
        d_off[j] = (unsigned)(((lc >> prm.logL) * prm.PPG + ps0) * (int)pair_bytes + (d_band[j] * L + lamc) * RC * ldb +
                              d_u[j] * 16);
    }

    auto issue_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                const int q = k * 4 + j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + buf * CHUNK_BYTES + q * 1024), 16,
                                                         d_off[j] + (unsigned)(k * ldb), 0, 0, 0);
            }
            // advance this stream by one chunk
            d_u[j] += UPC;
            d_off[j] += UPC * 16;
            if (d_u[j] >= NUp) {
                d_u[j] -= NUp;
                const bool last = d_band[j] == nb - 1;
                d_off[j] += last ? delta_pair : delta_band;
                d_band[j] = last ? 0 : d_band[j] + 1;
            }
        }
    };

    // consumer LDS offsets: slot(l,k,w) = (k*64 + l)*4 + ((w + (l>>2)) & 3)
    const int rot = (lane >> 2) & 3;
    unsigned rd_off[UPC];
#pragma unroll
    for (int w = 0; w < UPC; ++w) rd_off[w] = lds0 + lane * 64 + (((w + rot) & 3) << 4);

    double left[R], bot[S], corner = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) left[i] = 1.0;
#pragma unroll
    for (int i = 0; i < S; ++i) bot[i] = 1.0;

    // prologue: NBUF-1 chunks in flight
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c) issue_chunk(c);

    int buf = 0, pbuf = NBUF - 1;
    for (int c = 0; c < prm.n_chunks; ++c) {
        issue_chunk(pbuf);
        // this lane's increments for the 4 macro-steps of the chunk: gall[k*UPC + w]
        vec_t gall[RC * UPC];
        {
            const unsigned cbo = buf * CHUNK_BYTES;
            lds_read_chunk<(NBUF - 1) * IPC>(gall, rd_off[0] + cbo, rd_off[1] + cbo, rd_off[2] + cbo, rd_off[3] + cbo);
        }

#pragma unroll
        for (int w = 0; w < UPC; ++w) {
            // -- increments of this macro-step: RC rows x CW coarse columns
            vec_t gv[RC];
#pragma unroll
            for (int k = 0; k < RC; ++k) gv[k] = gall[k * UPC + w];

            // -- row-unit start: left boundary K[i][0] = 1
            if (u == 0) {
                corner = 1.0;
#pragma unroll
                for (int i = 0; i < R; ++i) left[i] = 1.0;
            }

            // -- top row of the block: bottom row of the lane above (previous macro-step), or the band boundary
            double top[S];
            if (MULTIBAND) {
                double tb[S];
                if (is_top && band > 0) {
#pragma unroll
                    for (int i = 0; i < S; ++i) tb[i] = lds_read_f64(my_bnd + (unsigned)(u * S + i) * 8u);
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
                // one pair per wave: lane 0 is the only top lane and wave_shr leaves its `old` operand (1.0) in place
#pragma unroll
                for (int i = 0; i < S; ++i) top[i] = dpp_shr1(bot[i], 1.0);
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

            // -- sweep the R x S block column by column
            double cand[RC][CW];
#pragma unroll
            for (int cc = 0; cc < S; ++cc) {
                double above = top[cc];                      // K[i0][j+1]
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
                }
                bot[cc] = above;
            }
            corner = top[S - 1];

            if (MULTIBAND) {
                if (is_bot) {
#pragma unroll
                    for (int i = 0; i < S; ++i) lds_write_f64(my_bnd + (unsigned)(u * S + i) * 8u, bot[i]);
                }
            }

            // -- K[MM][NN] of a pair: one lane, once per pair (the asm keeps the select chain inside the branch)
            if (u == prm.u_f && band == nb - 1 && lam == prm.lam_f && ps >= 0 && ps < prm.PPG && pair0 + ps < prm.P) {
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

            // -- advance
            u += 1;
            if (u == NUp) {
                u = 0;
                band += 1;
                if (band == nb) { band = 0; ps += 1; }
            }
        }
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        pbuf = pbuf + 1 == NBUF ? 0 : pbuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, int DY, bool NAIVE, bool MULTIBAND, bool FULLWAVE, int NBUF>
int launch_one(const WaveParams &prm, int blocks, size_t lds_bytes, hipStream_t s) {
    auto kern = k_fwd_wave<T, DY, NAIVE, MULTIBAND, FULLWAVE, NBUF>;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVE), lds_bytes, s, prm);
    return check_launch();
}

// Tuning knobs (environment, read at launch): SK_WAVE_NBUF = chunks in the LDS ring (2 or 3),
// SK_WAVE_WPC = cap on resident waves per CU.  Defaults are what measured best on MI355X.
int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

template <typename T, int DY, bool NAIVE, int NBUF>
int launch_nv(const WaveParams &prm, bool multiband, int blocks, size_t lds_bytes, hipStream_t s) {
    const bool full = prm.logL == 6;
    if (multiband) return launch_one<T, DY, NAIVE, true, false, NBUF>(prm, blocks, lds_bytes, s);
    return full ? launch_one<T, DY, NAIVE, false, true, NBUF>(prm, blocks, lds_bytes, s)
                : launch_one<T, DY, NAIVE, false, false, NBUF>(prm, blocks, lds_bytes, s);
}

template <typename T, int DY>
int launch_dy(const WaveParams &prm, bool multiband, int nbuf, int blocks, size_t lds_bytes, hipStream_t s) {
    if (nbuf == 2)
        return prm.naive ? launch_nv<T, DY, true, 2>(prm, multiband, blocks, lds_bytes, s)
                         : launch_nv<T, DY, false, 2>(prm, multiband, blocks, lds_bytes, s);
    return prm.naive ? launch_nv<T, DY, true, 3>(prm, multiband, blocks, lds_bytes, s)
                     : launch_nv<T, DY, false, 3>(prm, multiband, blocks, lds_bytes, s);
}

}  // namespace

// Returns SK_ERR_UNSUPPORTED when the shape / layout is outside what this kernel handles; the caller
// then falls back to the simple kernel.
template <typename T>
int launch_fwd_wave(const T *inc_c, int64_t ld, const Geom &g, T *out_final, hipStream_t s) {
    constexpr int CW = Unit<T>::CW;
    const int NBUF = env_int("SK_WAVE_NBUF", 3) == 2 ? 2 : 3;
    const int DY = g.dyadic;
    if (DY > 3) return SK_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(inc_c) & 15) || ((ld * sizeof(T)) & 15)) return SK_ERR_UNSUPPORTED;
    const int NU = (g.Nc + CW - 1) / CW;
    if ((int64_t)NU * CW > ld) return SK_ERR_UNSUPPORTED;  // the last unit must stay inside the row
    const int NUp = (NU + UPC - 1) / UPC * UPC;
    const int RC = DY == 0 ? 4 : DY == 1 ? 2 : 1;
    const int S = CW << DY;

    // lanes per pair: the smallest power of two whose band covers all rows; else full waves and bands
    int logL = 2;
    while (logL < 6 && (RC << logL) < g.Mc) ++logL;
    int L = 1 << logL;
    int nb = (g.Mc + L * RC - 1) / (L * RC);
    if (nb > 1) {
        // band b+1 reads what band b's bottom lane wrote L-1 macro-steps after the top lane: needs NUp >= L
        while (L > NUp && logL > 2) { --logL; L >>= 1; }
        if (L > NUp) return SK_ERR_UNSUPPORTED;
        nb = (g.Mc + L * RC - 1) / (L * RC);
    }
    const int G = WAVE / L;
    const bool multiband = nb > 1;

    size_t lds_bytes = (size_t)NBUF * RC * UPC * 1024;
    if (multiband) lds_bytes += (size_t)G * NUp * S * sizeof(double);
    if (lds_bytes > 160 * 1024) return SK_ERR_UNSUPPORTED;

    // persistent waves: enough of them to fill the chip, each streaming PPG pairs per lane group
    int waves_per_cu = (int)((160 * 1024) / lds_bytes);
    if (waves_per_cu > 8) waves_per_cu = 8;
    // Row-major increments are fetched as lane-skewed 64-byte pieces, so every 128-byte line is touched by two
    // consecutive chunks; the second touch only hits while the lines of all resident waves fit the XCD's 4 MiB
    // L2 (measured: 1.02x HBM traffic at 3 waves/CU, 1.61x at 6 for RC = 2).  Keep the resident set small.
    const int l2_cap = 6 / RC > 2 ? 6 / RC : 2;
    if (waves_per_cu > l2_cap) waves_per_cu = l2_cap;
    const int wpc_cap = env_int("SK_WAVE_WPC", 0);
    if (wpc_cap > 0 && waves_per_cu > wpc_cap) waves_per_cu = wpc_cap;
    if (waves_per_cu < 1) waves_per_cu = 1;
    const int64_t max_waves = 256LL * waves_per_cu;
    const int64_t groups_needed = g.P;  // one pair per group at least
    int64_t waves = (groups_needed + G - 1) / G;
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
    const int64_t steps = PPG * nb * NUp + (L - 1);
    prm.n_chunks = (int)((steps + UPC - 1) / UPC);
    prm.u_f = (g.Nc - 1) / CW;
    prm.lam_f = ((g.Mc - 1) / RC) % L;
    prm.sel_f = ((g.Mc - 1) % RC) * CW + (g.Nc - 1) % CW;
    prm.naive = g.naive;

    switch (DY) {
        case 0: return launch_dy<T, 0>(prm, multiband, NBUF, (int)waves, lds_bytes, s);
        case 1: return launch_dy<T, 1>(prm, multiband, NBUF, (int)waves, lds_bytes, s);
        case 2: return launch_dy<T, 2>(prm, multiband, NBUF, (int)waves, lds_bytes, s);
        default: return launch_dy<T, 3>(prm, multiband, NBUF, (int)waves, lds_bytes, s);
    }
}

template int launch_fwd_wave<double>(const double *, int64_t, const Geom &, double *, hipStream_t);
template int launch_fwd_wave<float>(const float *, int64_t, const Geom &, float *, hipStream_t);

}  // namespace sk
