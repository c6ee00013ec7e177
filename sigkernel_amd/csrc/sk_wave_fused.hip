// sk_wave_fused.hip -- forward solver with the static kernel fused in (KIND 0: LINEAR, KIND 1: RBF, below).  The linear
// increments
//     inc[p][q] = s^2 <x[p+1]-x[p], y[q+1]-y[q]>
// are formed inside the sweep from the path differences, so neither G_static nor the increment matrix exists in HBM
// (SURVEY 8(f) #1 taken to its end).  The sweep is the one of sk_wave.hip (skewed row strips in registers, DPP neighbour
// exchange, persistent pipelining over pairs); what changes is where a macro-step's RC x 2 coarse increments come from:
//   * a lane keeps the (scaled) differences of ITS rows in registers, dxr[RC][8], reloaded when it starts a new pair
//     from a small LDS ring that the wave fills 8 lanes at a time, one macro-step ahead of the first lane that needs it;
//   * the y differences of the 2 coarse columns of the macro-step, dy[8][2], are read from an LDS ring over the lane
//     group's virtual column stream: slabs of 8 units x 8 dims (1 KiB, one LDS-DMA instruction every 8 macro-steps,
//     fetched a whole slab ahead).  Lanes 8 apart read the same unit of neighbouring slabs; so that they hit different
//     halves of the 256-byte bank row without padding, odd slabs are stored with each pair of dimension rows swapped
//     (the DMA lanes simply fetch the other row) and the reader addresses even and odd dimensions separately;
//   * 32 extra FMAs per macro-step (RC*2 coarse cells x 8 dims) replace the increment read.
// HBM traffic: the paths (MBs).  The kernel is bound by fp64 issue.  Scope: dim <= 8 (zero-padded; ND = 4 variants skip
// the upper four dimensions), one band per pair (M-1 <= 64*RC), dyadic <= 2; everything else takes sk_static_increments
// + sk_solve_fwd.
//
// KIND 1 (RBF): the two rings hold the path POINTS.  A lane evaluates the nodes G[p][q] = exp(-|x_p - y_q|^2 / sigma) of the
// TOP node rows of its RC coarse rows at the two node columns of a unit (one exp per coarse cell) and takes the node row
// under its last coarse row from the lane BELOW by DPP (wave_shl:1).  That lane is one macro-step behind, so the node
// evaluation runs LAG = 2 units ahead of the block sweep: (u, ps) is the node cursor (it drives the rings and the reload
// of the x rows), (uk, psk) the sweep cursor (it drives the K state, the outputs and the edges).  own[k][0..5] holds a
// lane's node rows at the columns of units uk, uk+1, uk+2, bel[0..3] the row below at units uk, uk+1; the increments are
// the reference's 4-corner differences ((G11 + G00) - G10) - G01.  The last lane's missing neighbour and the one column
// a pair's last unit borrows from the next pair are padding: M <= 64*RC and N <= 2*NUp.
// All waves are independent (no barrier, private LDS slice) and launched four per workgroup (sk_wave_common.h).
#include "sk_wave_common.h"
#include <algorithm>

namespace sk {
namespace {

constexpr int FD = 8;              // dims carried (inputs are zero-padded to 8)
constexpr int y_slab_pitch(int nd) { return nd * 128; }   // ND dimension rows of 8 units (the four-dimension variants stage and keep
                                                          // dims 0..3 only); no padding (parity swizzle, see above)
constexpr int X_SLOTS = 2;   // the window being consumed + the one in flight
constexpr int x_row_bytes(int nd) { return nd == 4 ? 32 : 64; }

struct FusedParams {
    const double *dXr;   // [A][Mrows][8]: kappa s^2 (x[p+1]-x[p]), kappa = 4^-d / sqrt(12) (sk_linear_prescale); zero rows/dims beyond Mc / D
    const double *dYt;   // [Bn][8][Ncp]: y[q+1]-y[q], dimension-major, zero columns/dims beyond Nc / D
    void *out;           // [P] K[MM][NN]
    double *edges;       // EDGES variant: [P][NUp*S + L*R] terminal row and column in the strip layout (sk_wave.hip)
    int64_t P, B;        // B > 0: Gram, pair p = (p / B, p % B); B == 0: paired, pair p = (p, p)
    int Mrows, Ncp;
    int Mc, Nc, NUp, logL, PPG, n_steps;
    int u_f, lam_f, sel_f, naive;
    double inv_sigma;    // RBF: G = exp(-|x - y|^2 * inv_sigma)
    int dims;            // path dimensions that can be non-zero (<= 8)
    int e_NUp, e_L;      // EDGES: units per row / lanes per pair of the strip layout the adjoint reads (strip_geom)
    int tri;             // bits 0..1: 1: the P = A (A + 1) / 2 pairs enumerate the upper triangle (a <= b, row-major) of an A x A Gram of ONE
                         // path batch (A = B); out is [A][A] and receives both (a, b) and (b, a)
                         // 2: the LOSS layout (sk_solve_fwd_loss_f64): both staged arrays hold ONE batch Z of B paths; pairs [0, A B)
                         // are the rectangle (p / B, p % B) -- rows Z[0 .. A) against all of Z -- and the pairs behind them the STRICT
                         // upper triangle (i < j, row-major) of the tri_n paths from Z[A] on; out is [P], in pair order; with EDGES only
                         // the rectangle's pairs keep theirs.  A = bits 2..16, tri_n = bits 17..31 (both < 32768): ONE scalar register
                         // for a mode most launches do not use -- the step loop of this kernel lives at the limit of the scalar file
    WaveGroup wg;
    // The pairs of a launch are dealt to the waves as a stream of chunks (PairStream, below): chunk 0 of every wave is fixed --
    // C0 pairs per lane group, wave w starts at pair w G C0 -- and the rest is drawn, 2^logC pairs per lane group at a time,
    // from the launch's counter `queue` (zeroed by the launcher), so that the waves finish together whatever their SIMD's
    // arbitration, their XCD's clock or what else runs on the chip.  queue == nullptr: C0 is the whole share, nothing is drawn.
    unsigned long long *queue;
    int64_t q_first;     // first pair handed out by the counter (= waves G C0)
    int C0, logC;
    int n_big;           // queue == nullptr: waves [0, n_big) take C0 + 1 pairs per lane group (wave w starts at G (w C0 + min(w, n_big)))
    // queue == nullptr, rk_n > 0: shares by wave AGE RANK instead (sk_wave_common.h: the SIMD arbiter favours its oldest wave, so equal
    // shares leave a SIMD with two waves, then one, for the last third of a launch): a wave of rank r = w / rk_wpr takes rk_cnt[r]
    // pairs per lane group from pair rk_base[r] + (w - r rk_wpr) G rk_cnt[r] on
    int rk_n, rk_wpr;                 // (cnt packed 4 x 16 bits; rk_base[r] = G rk_wpr (cnt[0] + .. + cnt[r-1]) follows from them)
    unsigned long long rk_cnt;
};

template <int N>
__device__ __forceinline__ void lds_read_units(d2_t (&v)[N], unsigned addr);
// y units of a macro-step: even dimensions at a_even + {0, 256, 512, 768}, odd ones at a_odd + the same; the two addresses
// differ by +-128 (parity swizzle of the slabs).
// The reads carry no wait: they are issued at the end of a macro-step for the next one, so that their round trip
// overlaps this wave's own block sweep.  `t` is written by the LDS and read by nothing until lds_dims_wait hands it
// over (outputs tied to the temporaries' registers; tools/check_async_hazards.py lints the ISA for early uses).
__device__ __forceinline__ void lds_read_dims_issue(d2_t (&t)[8], unsigned a_even, unsigned a_odd) {
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %9\n\t"
                 "ds_read_b128 %2, %8 offset:256\n\t"
                 "ds_read_b128 %3, %9 offset:256\n\t"
                 "ds_read_b128 %4, %8 offset:512\n\t"
                 "ds_read_b128 %5, %9 offset:512\n\t"
                 "ds_read_b128 %6, %8 offset:768\n\t"
                 "ds_read_b128 %7, %9 offset:768"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                 : "v"(a_even), "v"(a_odd)
                 : "memory");
}
// dims 0..3 only (ND = 4)
__device__ __forceinline__ void lds_read_dims_issue(d2_t (&t)[4], unsigned a_even, unsigned a_odd) {
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %5\n\t"
                 "ds_read_b128 %2, %4 offset:256\n\t"
                 "ds_read_b128 %3, %5 offset:256"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                 : "v"(a_even), "v"(a_odd)
                 : "memory");
}
__device__ __forceinline__ void lds_dims_wait(d2_t (&v)[4], d2_t (&t)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]) : "memory");
}
// the first 32 bytes of two consecutive 64-byte rows (ND = 4), one wait
__device__ __forceinline__ void lds_read_half_rows(d2_t (&v)[4], unsigned a) {
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:16\n\t"
                 "ds_read_b128 %2, %4 offset:64\n\t"
                 "ds_read_b128 %3, %4 offset:80\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(a)
                 : "memory");
}
__device__ __forceinline__ void lds_dims_wait(d2_t (&v)[8], d2_t (&t)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
                 : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]), "4"(t[4]), "5"(t[5]), "6"(t[6]), "7"(t[7])
                 : "memory");
}
// 128 contiguous bytes (two coarse rows of x differences), one wait
__device__ __forceinline__ void lds_read_line(d2_t (&v)[8], unsigned a) {
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %8 offset:16\n\t"
                 "ds_read_b128 %2, %8 offset:32\n\t"
                 "ds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %8 offset:64\n\t"
                 "ds_read_b128 %5, %8 offset:80\n\t"
                 "ds_read_b128 %6, %8 offset:96\n\t"
                 "ds_read_b128 %7, %8 offset:112\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(a)
                 : "memory");
}
template <>
__device__ __forceinline__ void lds_read_units<4>(d2_t (&v)[4], unsigned a) {
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:16\n\t"
                 "ds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(a)
                 : "memory");
}

// x-row reloads straight into the row registers (read-write operands: under a divergent branch the inactive lanes keep theirs).
// No wait inside: lds_rows_wait (or any later s_waitcnt lgkmcnt(0) that precedes the first use) hands the rows over.
__device__ __forceinline__ void lds_rows_wait(d2_t (&r)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
}
__device__ __forceinline__ void lds_rows_wait(d2_t (&r)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]) : : "memory");
}
__device__ __forceinline__ void lds_load_line(d2_t (&r0)[4], d2_t (&r1)[4], unsigned a) {     // 128 contiguous bytes: two rows of 8 dims
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\t"
                 "ds_read_b128 %7, %8 offset:112"
                 : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]), "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]), "+v"(r1[3])
                 : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_load_two_half_rows(d2_t (&r0)[2], d2_t (&r1)[2], unsigned a) {   // two consecutive 32-byte rows
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                 : "+v"(r0[0]), "+v"(r0[1]), "+v"(r1[0]), "+v"(r1[1]) : "v"(a) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_load_row(d2_t (&r)[N], unsigned a);
template <>
__device__ __forceinline__ void lds_load_row<4>(d2_t (&r)[4], unsigned a) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void lds_load_row<2>(d2_t (&r)[2], unsigned a) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "+v"(r[0]), "+v"(r[1]) : "v"(a) : "memory");
}

// RCX: coarse rows per lane when not the strip kernels' own (Tile<DY>::RC) -- 2 for the RBF kernel at dyadic 0 with 8 staged dims,
// whose four-row form spills (its edges go into the strip layout all the same: e_L, see launch_fwd_fused)
template <typename TO, int DY, bool NAIVE, bool FULLWAVE, bool EDGES, int KIND, int ND, int RCX = 0>
__global__ __launch_bounds__(4 * WAVE) void k_fwd_fused(const FusedParams prm) {
    constexpr bool RBF = KIND == 1;   // ND: dimensions that can be non-zero (4 or 8); the arrays always carry FD = 8
    constexpr bool AHEAD = !(RBF && EDGES && ND == 8);   // y units read one macro-step ahead: 2 * ND more VGPRs, which the RBF kernel
                                              // that also keeps edges does not have below the 3-waves-per-SIMD line (168)
    constexpr int LAG = RBF ? 2 : 0;   // macro-steps by which the block sweep trails the node evaluation (see the header)
    constexpr int CW = 2;
    constexpr int RC = RCX ? RCX : Tile<DY>::RC, R = RC << DY, S = CW << DY, r = 1 << DY;
    // (RCX with EDGES: the launcher sets e_L so that e_L R is the strip layout's padded row count -- the rows this variant's lanes
    // do not cover are padding)
    // x rows in LDS: 64 bytes (8 dims) each, or -- the four-dimension variants -- the 32 bytes that can be non-zero: with four or
    // eight pairs per wave (short paths) the x ring was what held a CU to one wave per SIMD at dyadic 0
    constexpr int XROW = x_row_bytes(ND);
    constexpr int XSLAB = RC * 8 * XROW;   // 8 lanes x RC rows
    constexpr int Y_SLAB_PITCH = y_slab_pitch(ND);
    extern __shared__ __attribute__((aligned(16))) char lds_block[];
    char *lds;
    const int64_t wave_id = wave_slot(prm.wg, lds_block, lds);   // independent waves, see sk_wave_common.h
    if (wave_id < 0) return;
    const unsigned lds0 = lds_offset(lds);

    const int lane = threadIdx.x & (WAVE - 1);
    const int L = 1 << prm.logL, G = WAVE >> prm.logL;
    const int lam = lane & (L - 1), grp = lane >> prm.logL;
    const int NUp = prm.NUp;
    const int NSLAB = (L >> 3) + 2;                       // y slabs resident per lane group
    const unsigned y_bytes = (unsigned)(NSLAB * Y_SLAB_PITCH);
    const unsigned x_base0 = (unsigned)G * y_bytes;       // x rings behind all y rings
    const double sc = 1.0 / (double)(1 << (2 * DY));
    const double c_half = 0.5 * sc, c_12 = sc * sc / 12.0;

    // ---- consumer state: virtual unit v = t - lam; one band per pair, so the row unit IS the pair ----------
    int u, ps;
    {
        ps = floor_div(-lam, NUp);
        u = -lam - ps * NUp;
    }
    int uk = u, psk = ps;   // RBF: the unit / pair the block sweep is at (LAG steps behind); linear: the same as (u, ps)
    if (RBF) {
        psk = floor_div(-lam - LAG, NUp);
        uk = -lam - LAG - psk * NUp;
    }
    int yslab, ypar;   // slab of the y ring holding virtual unit v, and that slab's storage parity
    {
        const int s0 = floor_div(-lam, 8);
        yslab = ((s0 % NSLAB) + NSLAB) % NSLAB;
        ypar = (s0 + grp) & 1;
    }
    const int my_uf = lam == prm.lam_f ? prm.u_f : -1;
    const int lam7 = lam & 7;
    // Without edges nothing needs the per-lane cursors in every macro-step: the step counter is kept modulo NUp in scalar
    // registers (tm, tq) and a lane compares it with constants of its own -- u == 0 when tm == c_u0, uk == 0 when
    // tm == c_uk0, the pair's K[MM][NN] is ready when tm == c_out -- and the y ring is walked by one running address (a_e:
    // even dimension rows, the odd ones at a_e ^ 128, see lds_read_dims_issue).
    constexpr bool CUR = EDGES;   // per-lane (u, ps, uk, psk, yslab, ypar) cursors
    constexpr bool MID = false;   // fetch_next in the middle of a step instead of at its end (see there)
    // x rows may stay in flight across the loop edge -- not in the variant that is short of registers, where the allocator
    // moves the destinations of pending loads (tools/check_async_hazards.py)
    constexpr bool INFLIGHT_X = !(KIND == 1 && DY == 0 && ND == 8 && RCX == 0);
    int tm = 0, tq = 0;
    int c_u0 = lam % NUp, c_uk0 = (lam + LAG) % NUp;
    int c_out = lam == prm.lam_f ? (lam + LAG + prm.u_f) % NUp : -1;
    int c_kq = floor_div(-lam - LAG, NUp), c_kr = (-lam - LAG) - c_kq * NUp;   // t - lam - LAG = (tq + c_kq) NUp + tm + c_kr
    int c_u0m1 = (c_u0 + NUp - 1) % NUp;                                        // tm of the step BEFORE the lane starts a pair
    asm volatile("" : "+v"(c_u0), "+v"(c_uk0), "+v"(c_out), "+v"(c_kq), "+v"(c_kr), "+v"(c_u0m1));
    unsigned a_e;   // the odd rows are at a_e ^ 128: wave slices and slabs are 256-byte aligned, a slab row is 128 bytes
    // ---- the wave's stream of pairs: position i of lane group g is pair cb[k] + g size(k) + off, (k, off) = chunk and offset of i.
    // Pair indices are 32-bit here (the launcher refuses P >= 2^31 - 2^20); NOPAIR marks "no such pair".
    // (The chunk bases are wave-uniform, but the compiler cannot see that through the atomic: readfirstlane says so, or they
    // live in VGPRs and every producer call computes its addresses with vector instructions.)
    constexpr unsigned NOPAIR = 0xffffffffu;
    const unsigned P32 = (unsigned)prm.P;
    // pairs from here on keep no edges (the loss layout's triangle); worked out where it is needed, not kept in a scalar register
    auto edge_pairs = [&]() __attribute__((always_inline)) -> unsigned {
        return (prm.tri & 3) == 2 ? (unsigned)((prm.tri >> 2) & 0x7fff) * (unsigned)prm.B : P32;
    };
    // (launches without a queue deal the pairs out as evenly as whole pairs allow: the first n_big waves take one pair more
    // per lane group than the others -- everything below, t_end included, follows from this wave's own C0.  Spreading those
    // waves evenly over the wave numbers instead was measured slower: 0.215 vs 0.197 ms on 128 x 128 symmetric pairs.)
    const int w32 = __builtin_amdgcn_readfirstlane((int)wave_id);
    int c0_ = prm.C0 + (w32 < prm.n_big ? 1 : 0);
    unsigned cb0_ = (unsigned)(G * (w32 * prm.C0 + (w32 < prm.n_big ? w32 : prm.n_big)));
    if (prm.rk_n > 0) {   // shares by age rank (scalar selects: rk_n <= 4)
        int rr = w32 / prm.rk_wpr;
        rr = rr >= prm.rk_n ? prm.rk_n - 1 : rr;
        const unsigned long long pk = prm.rk_cnt;
        c0_ = (int)((pk >> (16 * rr)) & 0xffffu);
        const unsigned long long below = pk & ((1ull << (16 * rr)) - 1ull);      // the counts of the older ranks
        const unsigned used = (unsigned)(below & 0xffffu) + (unsigned)((below >> 16) & 0xffffu) + (unsigned)((below >> 32) & 0xffffu);
        cb0_ = (used * (unsigned)prm.rk_wpr + (unsigned)((w32 - rr * prm.rk_wpr) * c0_)) * (unsigned)G;
    }
    const int C0 = __builtin_amdgcn_readfirstlane(c0_), logC = prm.logC, CQ = 1 << logC;
    unsigned cb0 = (unsigned)__builtin_amdgcn_readfirstlane((int)cb0_), cb1 = NOPAIR, cb2 = NOPAIR, cb3 = NOPAIR;   // chunk k in cb[k & 3]
    int have = 1;                     // chunks known so far
    int t_end = 0x7fffffff;           // macro-steps this wave runs: known once a draw comes back empty
    // (masks, not a chain of selects: hipcc turns `k == 0 ? cb0 : k == 1 ? cb1 : ...` into an indexed array in scratch memory)
    auto ring_at = [&](int kk) __attribute__((always_inline)) -> unsigned {
        return (cb0 & -(unsigned)(kk == 0)) | (cb1 & -(unsigned)(kk == 1)) | (cb2 & -(unsigned)(kk == 2)) | (cb3 & -(unsigned)(kk == 3));
    };
    auto chunk_of = [&](int i, int &off) __attribute__((always_inline)) -> int {
        if (i < C0) { off = i; return 0; }
        off = (i - C0) & (CQ - 1);
        return 1 + ((i - C0) >> logC);
    };
    // pair at stream position i of lane group g (NOPAIR: none), and how many positions of its chunk follow it (left)
    auto stream_pair_left = [&](int g, int i, int &left) __attribute__((always_inline)) -> unsigned {
        left = 0;
        if (i < 0) return NOPAIR;
        int off;
        const int k = chunk_of(i, off);
        const int size = k == 0 ? C0 : CQ;
        left = size - 1 - off;
        const unsigned b = ring_at(k & 3);
        const unsigned p = b + (unsigned)(g * size + off);
        return (b >= P32 || p >= P32) ? NOPAIR : p;
    };
    auto stream_pair = [&](int g, int i) __attribute__((always_inline)) -> unsigned {      // wave-uniform arguments in the producers
        int left;
        return stream_pair_left(g, i, left);
    };
    auto lane_pair = [&](int i) __attribute__((always_inline)) -> unsigned {               // per-lane position, this lane's group
        int left;
        return stream_pair_left(grp, i, left);
    };
    // make sure the chunk of stream position f is known (the producers call this with the furthest position they touch)
    auto ensure = [&](int f) __attribute__((always_inline)) {
        int off;
        const int kf = chunk_of(f, off);
        while (have <= kf) {
            unsigned b = NOPAIR;
            if (prm.queue && t_end == 0x7fffffff) {
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(prm.queue, (unsigned long long)(G * CQ));
                const unsigned long long q = (unsigned long long)prm.q_first +
                                             (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                              (unsigned)__builtin_amdgcn_readfirstlane((int)v));
                b = q < (unsigned long long)P32 ? (unsigned)q : NOPAIR;
            }
            if (b == NOPAIR && t_end == 0x7fffffff)
                // the stream ends where chunk `have` would begin; without edges to keep, a wave is done with the step in which
                // the lane of the last row stores the last pair's K[MM][NN] (its unit u_f, not the padded NUp - 1)
                t_end = EDGES ? (C0 + (have - 1) * CQ) * NUp + (L - 1) + LAG
                              : (C0 + (have - 1) * CQ - 1) * NUp + prm.u_f + prm.lam_f + LAG + 1;
            const int kk = have & 3;
            const unsigned m0 = -(unsigned)(kk == 0), m1 = -(unsigned)(kk == 1), m2 = -(unsigned)(kk == 2), m3 = -(unsigned)(kk == 3);
            cb0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb0 & ~m0) | (b & m0)));
            cb1 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb1 & ~m1) | (b & m1)));
            cb2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb2 & ~m2) | (b & m2)));
            cb3 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb3 & ~m3) | (b & m3)));
            have += 1;
        }
    };
    const bool is_top = lam == 0;
    const unsigned my_y = lds0 + (unsigned)grp * y_bytes;
    const unsigned y_lim = my_y + y_bytes;
    const unsigned ring_bytes = FULLWAVE ? (unsigned)(((WAVE >> 3) + 2) * Y_SLAB_PITCH) : y_bytes;   // = y_bytes, as a literal when L = 64
    {
        const unsigned ya = my_y + (unsigned)(yslab * Y_SLAB_PITCH + ((-lam & 7) << 4));
        a_e = ya + (unsigned)(ypar << 7);
    }
    // lanes NUp apart start (different) pairs at the same macro-step: one x slab per such "lap" j = lam / NUp
    const int JMAX = (L + NUp - 1) / NUp;
    const unsigned my_x = lds0 + x_base0 + (unsigned)((grp * X_SLOTS * JMAX) * XSLAB + (lam / NUp) * XSLAB) +
                          (unsigned)((lam & 7) * RC * XROW);

    // ---- producers (uniform control; per-lane source offsets) ------------------------------------------------
    // y slab s = virtual units [8s, 8s+8) of every lane group: dims k = lane/8, unit x = lane%8
    // The producers run once per 8 macro-steps with wave-uniform control.  Their cursors advance incrementally (no division
    // by NUp), and the pair -> (a, b) split uses 32-bit arithmetic whenever the pair count allows: the 64-bit division
    // sequence is ~150 scalar instructions, and there are G of them per call.
    // (the pairs of the triangular layouts -- symmetric Gram, loss layout -- come from a table [P][2] of int32 (a, b) that sk_prep_cat_*
    // writes right BEHIND the staged columns: found from dYt, B and Ncp, which the producers hold anyway.  The triangle arithmetic
    // inside this kernel -- a square root and two correction loops per look-up, inlined four times -- sat in the scalar registers of
    // EVERY launch: 25 v_readlane / v_writelane in the headline variant against 17 without it, profiles/r05_ab_r04_vs_r05.txt)
    auto split_ab = [&](int64_t p, bool want_b) __attribute__((always_inline)) -> int64_t {
        if (prm.B <= 0) return p;
        if (prm.tri) {
            const int *tab = reinterpret_cast<const int *>(prm.dYt + prm.B * (int64_t)FD * prm.Ncp);
            return (int64_t)tab[2 * p + (want_b ? 1 : 0)];
        }
        // (32-bit: the launcher refuses P >= 2^31 - 2^20, and B <= P in a Gram launch -- the 64-bit division sequence is ~150 scalar
        // instructions, and it used to be inlined here twice for a case that cannot occur)
        if (want_b) return (int64_t)((uint32_t)p % (uint32_t)prm.B);
        return (int64_t)((uint32_t)p / (uint32_t)prm.B);
    };
    auto split_b = [&](int64_t p) -> int64_t { return split_ab(p, true); };
    auto split_a = [&](int64_t p) -> int64_t { return split_ab(p, false); };
    int y_pi = 0, y_u0 = 0, y_slot = 0, y_par = 0;   // next y slab: pair-in-group, first unit (NUp % 8 == 0: no straddling),
    auto issue_y = [&]() __attribute__((always_inline)) {                            // ring slot, parity of the virtual slab number
        ensure(y_pi);
        for (int g = 0; g < G; ++g) {
            const unsigned sp = stream_pair(g, y_pi);
            const int64_t p = sp == NOPAIR ? 0 : (int64_t)sp;   // past the end: fetch something valid, never consumed
            const int64_t b = split_b(p);
            const int krow = (lane >> 3) ^ ((y_par + g) & 1);   // odd slabs (per group): dimension rows swapped in pairs
            const double *src = prm.dYt + ((b * FD + krow) * (int64_t)prm.Ncp + (int64_t)(y_u0 + (lane & 7)) * 2);
            if (ND == 8 || lane < 32)   // (ND = 4: dimension rows 0..3 = the lower half of the wave; LDS-DMA writes lane l's 16 bytes at base + 16 l)
                __builtin_amdgcn_global_load_lds(src, (lds_void *)(lds + g * y_bytes + y_slot * Y_SLAB_PITCH), 16, 0, 0);
        }
        y_slot = y_slot + 1 == NSLAB ? 0 : y_slot + 1;
        y_par ^= 1;
        y_u0 += 8;
        if (y_u0 == NUp) { y_u0 = 0; y_pi += 1; }
    };
    // x slabs for the lanes that start a pair during macro-steps [t0, t0+8): lanes lam0 + j*NUp .. +7 start pair
    // t0/NUp - j, rows (lam0 + j*NUp .. +7)*RC of its x
    int x_q0 = 0, x_lam0 = 0, x_slot = 0;   // next window: t0 / NUp, t0 % NUp, ring slot
    auto issue_x = [&]() __attribute__((always_inline)) {
        ensure(x_q0);
        for (int j = 0; j < JMAX; ++j) {
            const int lamj = x_lam0 + j * NUp, pi = x_q0 - j;
            if (lamj >= L) break;
            for (int g = 0; g < G; ++g) {
                const unsigned sp = stream_pair(g, pi);
                const int64_t p = sp == NOPAIR ? 0 : (int64_t)sp;
                const int64_t a = split_a(p);
                const char *src = reinterpret_cast<const char *>(prm.dXr + (a * prm.Mrows + (int64_t)lamj * RC) * FD);
                char *dst = lds + x_base0 + ((g * X_SLOTS + x_slot) * JMAX + j) * XSLAB;
#pragma unroll
                for (int c = 0; c < (XSLAB + 1023) / 1024; ++c)
                    if (c * 1024 + lane * 16 < XSLAB) {
                        // LDS-DMA lands lane l's 16 bytes at dst + 16 l; XROW = 32: the first two 16-byte pieces of every 64-byte row
                        const int so = XROW == 64 ? c * 1024 + lane * 16 : (c * 32 + (lane >> 1)) * 64 + (lane & 1) * 16;
                        __builtin_amdgcn_global_load_lds(src + so, (lds_void *)(dst + c * 1024), 16, 0, 0);
                    }
            }
        }
        x_slot = x_slot + 1 == X_SLOTS ? 0 : x_slot + 1;
        x_lam0 += 8;
        if (x_lam0 == NUp) { x_lam0 = 0; x_q0 += 1; }
    };

    // this lane's x rows as 16-byte register pairs: the reload's ds_read_b128 lands in them directly (tied "+v" operands:
    // the lanes that do not start a pair keep their values under the exec mask) -- no temporaries, no v_mov per macro-step
    d2_t dxq[RC][ND / 2];
#pragma unroll
    for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int j = 0; j < ND / 2; ++j) dxq[k][j] = d2_t{0.0, 0.0};
    auto load_x_rows = [&](unsigned xa) {
        if constexpr (RC % 2 == 0 && ND == 8) {   // two rows per instruction group
#pragma unroll
            for (int k = 0; k < RC; k += 2) lds_load_line(dxq[k], dxq[k + 1], xa + k * 64u);
        } else if constexpr (RC % 2 == 0) {      // ND = 4: 32-byte rows, two of them are 64 contiguous bytes
#pragma unroll
            for (int k = 0; k < RC; k += 2) lds_load_two_half_rows(dxq[k], dxq[k + 1], xa + k * (unsigned)XROW);
        } else {
#pragma unroll
            for (int k = 0; k < RC; ++k) lds_load_row<ND / 2>(dxq[k], xa + k * (unsigned)XROW);
        }
    };
    // RBF: node values of this lane's rows at the columns of units uk, uk + 1, uk + 2 (the last two filled this step), and
    // of the first row of the lane below at the columns of units uk and uk + 1
    double own[RBF ? RC : 1][6], bel[4];
    // the variant that also keeps edges has no 22 VGPRs to spare for the coefficients below the 3-waves-per-SIMD line
    constexpr bool PIN_EXP = RBF && !EDGES;
    ExpCoef expc;
    if (PIN_EXP) expc.init();
#pragma unroll
    for (int k = 0; k < (RBF ? RC : 1); ++k)
#pragma unroll
        for (int c = 0; c < 6; ++c) own[k][c] = 1.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) bel[c] = 1.0;
    double left[R], bot[S], ktop[S], corner = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) left[i] = 1.0;
#pragma unroll
    for (int i = 0; i < S; ++i) { bot[i] = 1.0; ktop[i] = 1.0; }

    // EDGES: terminal row / column of every pair, register -> global, held one macro-step (see sk_wave.hip)
    // Everything the per-step edge bookkeeping compares against is loop-invariant and PER LANE, and is kept in VGPRs (the
    // asm pins): as wave-uniform values it filled the scalar file and came back through v_readlane in every macro-step.
    const int EP = EDGES ? (prm.e_NUp * S + prm.e_L * R) : 0;
    int erow_lim = EDGES && lam == prm.lam_f ? prm.e_NUp : 0;          // this lane holds the terminal row: units below this
    int ecol_uf = EDGES && lam < prm.e_L ? prm.u_f : -1;               // the unit whose block holds the terminal column
    int ecol_off = EDGES ? prm.e_NUp * S + lam * R : 0;
    asm volatile("" : "+v"(erow_lim), "+v"(ecol_uf), "+v"(ecol_off));
    // EDGES: edge block of the pair this lane's sweep is in (nullptr: no such pair / a pair that keeps no edges), moved where psk changes.
    // Consecutive stream positions of a chunk are consecutive pairs, so inside a chunk the pointer just advances; the FIRST pair of a
    // chunk has to be looked up (chunk bases are whatever the queue handed out).  That look-up used to sit in the macro-step path
    // (sweep_pair_is, taken by some lane 0.4-0.8 times per step under the queue), and although it is only ~50 instructions it cost the
    // edge-keeping forward ~5 % (profiles/r05_edges_ablation.txt): everything it reads -- the chunk ring, C0, CQ, P, the layout --
    // stayed in scalar registers across the step loop, and the loop's own values came back through v_readlane instead.  Now every lane
    // keeps the block of its NEXT chunk's first pair ready (nx_*), refreshed once per 8 macro-steps in the producers' block (where those
    // scalars are at home), and entering a chunk is three moves.  A chunk lasts >= NUp >= 8 steps and is drawn >= 9 steps before lane 0
    // enters it, so the refresh between two entries of a lane always finds the chunk known.
    double *ep_cur = nullptr, *nx_ptr = nullptr;
    int ep_left = 0, nx_left = 0;         // pairs of the lane's chunk after the current one / of the next chunk after its first
    int ep_ok = 0, nx_ok = 0;             // ... how many of those keep edges (the loss layout's triangle keeps none)
    int ep_k = 0;                         // chunks this lane's sweep has entered = index of its next chunk
    auto refresh_next = [&]() __attribute__((always_inline)) {
        if (EDGES) {
            if (ep_k < have) {
                asm volatile("");
                const int size = ep_k == 0 ? C0 : CQ;
                const unsigned b = ring_at(ep_k & 3);
                const unsigned p = b + (unsigned)(grp * size);
                const unsigned ne = edge_pairs();
                const bool ok = b < P32 && p < P32 && p < ne;
                nx_ptr = ok ? prm.edges + (uint64_t)p * (uint64_t)(unsigned)EP : nullptr;
                nx_left = size - 1;
                const unsigned room = ok ? ne - 1u - p : 0u;
                nx_ok = (int)(room < (unsigned)(size - 1) ? room : (unsigned)(size - 1));
            }
        }
    };
    auto sweep_pair_is = [&](int pk) __attribute__((always_inline)) {
        if (EDGES) {
            if (ep_left > 0) {
                ep_left -= 1;
                if (ep_ok > 0) {
                    ep_ok -= 1;
                    ep_cur += EP;
                } else {
                    ep_cur = nullptr;
                }
            } else if (pk >= 0) {
                ep_cur = nx_ptr;
                ep_left = nx_left;
                ep_ok = nx_ok;
                ep_k += 1;
            } else {
                ep_cur = nullptr;
            }
        }
    };
    double *e_ptr = nullptr;    // ... of the values held for the next step's stores
    double erow[S];
    int erow_at = -1, ecol_at = -1;
    const int k_f = (prm.Mc - 1) % RC;
    const bool row_in_bot = k_f == RC - 1;   // then the row values are the `bot` state (see sk_wave.hip)

    // DMA: slab / window n+1 is issued when the steps of slab / window n begin, and waited for when they end.  The y
    // differences of a macro-step are read from LDS at the end of the previous one.
    d2_t dyn[ND];
    auto read_y = [&]() {
        if constexpr (CUR) {
            const unsigned ya = my_y + (unsigned)(yslab * Y_SLAB_PITCH + ((u & 7) << 4));
            lds_read_dims_issue(dyn, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
        } else {
            lds_read_dims_issue(dyn, a_e, a_e ^ 128u);
        }
    };
    unsigned x_rd_off = 0;   // ring slot the x rows of this 8-step window are read from: ((t >> 3) % X_SLOTS) * JMAX * XSLAB
    static_assert(X_SLOTS == 2, "x_rd_off toggles between two slots");
    issue_y();
    issue_x();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue_y();
    issue_x();
    refresh_next();
    sweep_pair_is(psk);
    refresh_next();
    if constexpr (!CUR) {
        if (c_u0 == 0) {   // lanes that start a pair in macro-step 0 (their K state is 1.0 already)
            load_x_rows(my_x);
#pragma unroll
            for (int k = 0; k < RC; ++k) lds_rows_wait(dxq[k]);
        }
    }
    if (AHEAD) read_y();
    for (int t = 0; t < t_end; ++t) {
        if (EDGES) {   // the edge values of the previous macro-step, straight from the state registers
            double *const ep = e_ptr;
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

        // -- top row of the block from the lane above
        double top[S];
        if (FULLWAVE) {   // lane 0 keeps the 1.0 of the persistent `old` register ktop[i] (see sk_wave.hip)
            // corner = the previous step's last top value, taken before the DPP overwrites it.  (An opaque move: left to the
            // compiler, the copy is made three times over.)
            asm volatile("v_mov_b64 %0, %1" : "=v"(corner) : "v"(ktop[S - 1]));
#pragma unroll
            for (int i = 0; i < S; ++i) {
                ktop[i] = dpp_shr1(bot[i], ktop[i]);
                top[i] = ktop[i];
            }
        } else {
            // KTOP_KEPT: into the persistent ktop as well -- the DPP's `old` operand is then a register that is live anyway (a fresh 1.0
            // per value is two moves per value and macro-step); the tops of the other lane groups are selected as before.  It costs 2 S
            // VGPRs, so only the doubled-row variants take it: their LDS holds them to two waves per SIMD whatever the registers (one
            // exception: linear with 8 dims and edges at dyadic 2 would drop from three waves to two)
            constexpr bool KTOP_KEPT = RCX != 0 && RCX == 2 * Tile<DY>::RC && !(KIND == 0 && ND == 8 && DY == 2);
#pragma unroll
            for (int i = 0; i < S; ++i) {
                if constexpr (KTOP_KEPT) {
                    ktop[i] = dpp_shr1(bot[i], ktop[i]);
                    top[i] = is_top ? 1.0 : ktop[i];
                } else {
                    const double sh = dpp_shr1(bot[i], 1.0);
                    top[i] = is_top ? 1.0 : sh;
                }
            }
        }

        // -- start of a pair: left boundary K[i][0] = 1, and this lane's x rows.  Without edges the rows were fetched during the
        // previous macro-step (below); the wait for the y units hands them over.
        if constexpr (!CUR) {
            if (tm == (RBF ? c_uk0 : c_u0)) {
                asm volatile("");   // a real branch: if-converted, the five moves become ten v_cndmask in every macro-step
                corner = 1.0;
#pragma unroll
                for (int i = 0; i < R; ++i) left[i] = 1.0;
            }
        } else {
            if (RBF && uk == 0) {
                asm volatile("");
                corner = 1.0;
#pragma unroll
                for (int i = 0; i < R; ++i) left[i] = 1.0;
            }
            if (u == 0) {
                if (!RBF) {
                    asm volatile("");
                    corner = 1.0;
#pragma unroll
                    for (int i = 0; i < R; ++i) left[i] = 1.0;
                }
                load_x_rows(my_x + x_rd_off);
#pragma unroll
                for (int k = 0; k < RC; ++k) lds_rows_wait(dxq[k]);
            }
        }

        // -- y differences of the two coarse columns of this macro-step, all 8 dims
        d2_t dyv[ND];
        if (!AHEAD) read_y();
        lds_dims_wait(dyv, dyn);

        // -- increments and coefficients per coarse cell
        double ca[RC][CW], cbm[RC][CW];
        double ginc[RC][CW];
        if constexpr (RBF) {
            // nodes G[p][q] = exp(-|x_p - y_q|^2 / sigma) of this lane's RC top node rows at the two columns of unit u
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) {
                    double d2 = 0.0;
#pragma unroll
                    for (int j = 0; j < ND; ++j) {
                        const double df = dxq[k][j >> 1][j & 1] - dyv[j][q];
                        d2 = fma(df, df, d2);
                    }
                    // d2 * 0 is 0 for finite distances and NaN for an infinite (or NaN) one: the reference's
                    // |x|^2 + |y|^2 - 2<x,y> is inf - inf = NaN for an infinite coordinate, and so is this exponent
                    const double ex = fma(-d2, prm.inv_sigma, d2 * 0.0);
                    own[k][4 + q] = PIN_EXP ? exp_nonpos(ex, expc) : exp_nonpos(ex);
                }
            // the node row below this lane's last coarse row is the first row of the lane below, which is one macro-step
            // behind: what it has just evaluated are the columns of unit u - 1 = uk + 1
            bel[2] = dpp_shl1(own[0][4], bel[2]);
            bel[3] = dpp_shl1(own[0][5], bel[3]);
            // 4-corner differences in the reference's order (sigkernel.py:362-363): G11 + G00 - G10 - G01
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) {
                    const double t0 = own[k][q], t1 = own[k][q + 1];
                    const double b0 = k + 1 < RC ? own[(k + 1) % RC][q] : bel[q];
                    const double b1 = k + 1 < RC ? own[(k + 1) % RC][q + 1] : bel[q + 1];
                    ginc[k][q] = ((b1 + t0) - b0) - t1;
                }
            if (EDGES) {
                // the adjoint sweeps the padded strip with zero increments in the padding: the edges must come from the same
                // grid, so the (meaningless) node differences of padding rows / columns are dropped here
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int q = 0; q < CW; ++q)
                        if (lam * RC + k >= prm.Mc || uk * CW + q >= prm.Nc) ginc[k][q] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) { own[k][c] = own[k][c + 2]; asm volatile("" : "+v"(own[k][c])); }
            bel[0] = bel[2];
            bel[1] = bel[3];
            asm volatile("" : "+v"(bel[0]), "+v"(bel[1]));
            // (the empty asm statements make every shifted value an opaque definition: left to itself the copy coalescer resolved the
            // window's shift -- a chain of 2 RC + 1 overlapping copies -- through extra temporaries, 106 v_mov per macro-step of the
            // four-row variant against 54 now, 197 -> 169 VGPRs; 512 x 512 RBF pairs 1.89 -> 1.81 ms, C2 0.161 -> 0.157,
            // profiles/r06_exp_ab.txt.  Re-labelling the slots instead -- the loop unrolled by three -- was tried first: 181 -> 269
            // VGPRs and 72 lint hazards, the allocator copies the pending y units around.  The same pins on the adjoint's and the
            // multi-band forward's histories change nothing or add moves: not applied there.)
        } else {
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) {
                    double g = 0.0;
#pragma unroll
                    for (int j = 0; j < ND; ++j) g = fma(dxq[k][j >> 1][j & 1], dyv[j][q], g);
                    ginc[k][q] = g;
                }
        }
        // The y units of the NEXT macro-step and the x rows of a lane that starts a pair in it, fetched at the end of this
        // step and handed over by the single LDS wait at the top of the next (one LDS round trip per step instead of two).  On
        // a window boundary the rows are in the window whose DMA is waited for here.  Issuing the reads in the middle of the
        // step, under the sweep, was tried (MID): no faster, and the compiler copies the destination registers around.
        auto fetch_next = [&]() {
            const bool turn = ((t + 1) & 7) == 0;   // the next step opens a y slab (for lane 0) and an x window: their DMA
            if (__builtin_expect(turn, 0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // was issued 8 steps ago
            a_e += 16;
            if (((t + 1) & 7) == lam7) {   // next slab: the other parity, one slab less one row on, wrapping at the end of the ring
                asm volatile("");          // (a real branch: if-converted, the update costs two more VALU instructions per step)
                const unsigned e = (a_e + (unsigned)(Y_SLAB_PITCH - 128)) ^ 128u;
                a_e = e - (e >= y_lim ? ring_bytes : 0u);
            }
            read_y();
            if (tm == c_u0m1) {
                load_x_rows(my_x + (turn ? x_rd_off ^ (unsigned)(JMAX * XSLAB) : x_rd_off));
                if constexpr (!INFLIGHT_X) {
#pragma unroll
                    for (int k = 0; k < RC; ++k) lds_rows_wait(dxq[k]);
                }
            }
        };
        if constexpr (!CUR && MID) fetch_next();
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const double g = ginc[k][q];
                if constexpr (!RBF) {
                    // LINEAR: the staged x differences carry kappa = 4^-d / sqrt(12) (sk_linear_prescale), so g is kappa times the
                    // increment and 1 + inc/2 4^-d + inc^2 4^-2d / 12 = 1 + g (sqrt 3 + g), 1 - inc^2 4^-2d / 12 = 1 - g g:
                    // three operations per coarse cell instead of four
                    if (NAIVE) {
                        ca[k][q] = fma(g, 1.7320508075688772, 1.0);
                        cbm[k][q] = 1.0;
                    } else {
                        ca[k][q] = fma(g, g + 1.7320508075688772, 1.0);
                        cbm[k][q] = fma(-g, g, 1.0);
                    }
                } else if (NAIVE) {
                    ca[k][q] = fma(g, c_half, 1.0);
                    cbm[k][q] = 1.0;
                } else {
                    const double g2 = g * g;
                    ca[k][q] = fma(g2, c_12, fma(g, c_half, 1.0));
                    cbm[k][q] = fma(g2, -c_12, 1.0);
                }
            }

        // -- sweep the R x S block
        double cand[RC][CW];
        double rowv[RC][S];   // EDGES: K on the last fine row of each coarse row of the block
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
            double above = top[cc];
            double diag = cc == 0 ? corner : top[cc - 1];
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
        if (!FULLWAVE) corner = top[S - 1];

        if (EDGES) {
            const bool pair_ok = ep_cur != nullptr;
            e_ptr = ep_cur;
            erow_at = (pair_ok && uk < erow_lim) ? uk * S : -1;
            ecol_at = (pair_ok && uk == ecol_uf) ? ecol_off : -1;
            if (!row_in_bot) {
#pragma unroll
                for (int kk = 0; kk < RC - 1; ++kk)
                    if (kk == k_f) {
#pragma unroll
                        for (int cc = 0; cc < S; ++cc) erow[cc] = rowv[kk][cc];
                    }
            }
        }

        // -- K[MM][NN] of a pair
        if (__builtin_expect(CUR ? uk == my_uf : tm == c_out, 0)) {
            int pv = CUR ? psk : tq + c_kq + (tm + c_kr >= NUp ? 1 : 0);
            asm volatile("" : "+v"(pv));   // keeps the pair tests inside this (rarely taken) branch instead of in every step
            const unsigned pair_u = lane_pair(pv);
            const int64_t pair_v = (int64_t)pair_u;
            if (pair_u != NOPAIR) {
                double v = cand[0][0];
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        double cv = cand[k][q];
                        asm volatile("" : "+v"(cv));
                        if (k * CW + q == prm.sel_f) v = cv;
                    }
                if ((prm.tri & 3) == 1) {      // the pair and its mirror image
                    const int *tab = reinterpret_cast<const int *>(prm.dYt + prm.B * (int64_t)FD * prm.Ncp);
                    const int64_t a = tab[2 * pair_v], b = tab[2 * pair_v + 1];
                    static_cast<TO *>(prm.out)[a * prm.B + b] = (TO)v;
                    static_cast<TO *>(prm.out)[b * prm.B + a] = (TO)v;
                } else {
                    static_cast<TO *>(prm.out)[pair_v] = (TO)v;
                }
            }
        }

        // -- advance
        if constexpr (!CUR) {
            if constexpr (!MID) fetch_next();
            tm += 1;
            if (tm == NUp) { tm = 0; tq += 1; }
        }
        if (CUR && RBF) {
            uk += 1;
            if (uk == NUp) {
                uk = 0;
                psk += 1;
                sweep_pair_is(psk);
            }
        }
        if constexpr (CUR) u += 1;
        if (CUR && ((t + 1) & 7) == lam7) {   // (u & 7) == 0: u = t + 1 - lam modulo 8 (NUp is a multiple of 8)
            yslab = yslab + 1 == NSLAB ? 0 : yslab + 1;
            ypar ^= 1;
            if (u == NUp) {
                u = 0;
                ps += 1;
                if (!RBF) sweep_pair_is(ps);
            }
        }
        if (CUR && !RBF) { uk = u; psk = ps; }
        if (__builtin_expect(((t + 1) & 7) == 0, 0)) {
            // everything issued 8 macro-steps ago has had a whole slab period to land (leaving this step's edge stores in
            // flight with a counted wait was measured: no gain, their cost is issue slots, not latency)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            issue_y();       // slab ((t + 1) >> 3) + 1
            issue_x();       // window t + 9 .. t + 16
            x_rd_off ^= (unsigned)(JMAX * XSLAB);
            refresh_next();
        }
        if (AHEAD && CUR) read_y();   // for macro-step t + 1
    }
    if constexpr (!CUR) {   // the last read-ahead is never used, but its registers are not free before it has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (no VGPR is written between here and the end of the kernel)
    } else if (AHEAD) {
        d2_t drain[ND];
        lds_dims_wait(drain, dyn);
    }
    if (EDGES) {
        double *const ep = e_ptr;
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

// What the launcher has worked out before the kernel variant (and with it the register budget) is known
struct FusedPlan {
    int64_t P;
    int G, NUp, L, lag;
    size_t lds_bytes;   // per wave
    int waves_per_cu;   // from LDS and the measured optimum; still to be capped by the variant's VGPR use
    int rcx;            // coarse rows per lane when not the strip kernels' own (0: Tile<DY>::RC)
};

template <typename TO, int DY, bool NAIVE, bool FULLWAVE, bool EDGES, int KIND, int ND, int RCX = 0>
int launch_fused_nd(FusedParams prm, const FusedPlan &pl, hipStream_t s) {
    auto kern = k_fwd_fused<TO, DY, NAIVE, FULLWAVE, EDGES, KIND, ND, RCX>;
    // persistent waves: all of them must be resident at once, so the variant's VGPR count caps the waves per SIMD
    // (512 VGPRs per lane and SIMD; a variant over 168 holds two waves per SIMD, not three)
    // (a property of this variant's code object, the same on every gfx950 device: an immutable constant initialised once,
    // thread-safely, at the variant's first launch -- not mutable library state)
    static const int vgprs = [&] {
        hipFuncAttributes attr;
        return hipFuncGetAttributes(&attr, (const void *)kern) == hipSuccess && attr.numRegs > 0 ? attr.numRegs : 128;
    }();
    int waves_per_cu = pl.waves_per_cu;
    const int by_regs = 4 * (512 / ((vgprs + 7) & ~7));
    if (waves_per_cu > by_regs) waves_per_cu = by_regs;
    if (waves_per_cu < 1) waves_per_cu = 1;
    int64_t max_waves = (int64_t)device_cu_count() * waves_per_cu;
    int64_t waves = (pl.P + pl.G - 1) / pl.G;
    if (waves > max_waves) waves = max_waves;
    int64_t per = (pl.P + waves * pl.G - 1) / (waves * pl.G);      // the equal share, pairs per lane group
    if (per < 8 && waves_per_cu >= 8) {
        // A launch of a few pairs per lane group (no queue, see below): whole pairs do not divide evenly, every wave pays the
        // skew's fill (L - 1 + lag macro-steps) once, and fewer resident waves run faster each.  Take the resident waves per
        // SIMD q that minimise (macro-steps of the busiest SIMD) x (time per macro-step with q waves on it): 1.00 / 0.735 /
        // 0.65 for q = 1 / 2 / 3, from per-wave timestamps of the 128 x 128 symmetric RBF launch (1.13 / 0.83 / 0.73 us,
        // tools/experiments/r03_c2_wave_times.py) -- there 8256 pairs are 4096 lane groups x 2 + 64, and q = 2 (0.193 ms)
        // beats q = 3 (6144 lane groups, a third of them with two pairs: 0.220 ms).
        // (round 6, per-wave time of a macro-step against a lone wave's, from launches of exactly k pairs per lane group at q waves per
        // SIMD, profiles/r06_rc4_steps.txt: two rows per lane 1 / 0.70 / 0.62 without edges, 1 / 0.69 / 0.56 with; four rows per lane
        // 1 / 0.69 without, 1 / 0.79 with -- and a SIMD whose waves are of unequal length carries fewer of them once the short ones
        // are done: two phases, not one rate over the sum of the steps)
        static const double rate_std[4] = {1.0, 1.0, 0.70, 0.60}, rate_dbl[4] = {1.0, 1.0, EDGES ? 0.79 : 0.69, 0.62};
        const double *rate = (RCX != 0 && RCX == 2 * Tile<DY>::RC) ? rate_dbl : rate_std;
        const int64_t nsimd = (int64_t)device_cu_count() * 4;
        const int fill = pl.L - 1 + pl.lag;
        double best = 0;
        int best_q = 0;
        for (int q = 1; q <= waves_per_cu / 4 && q <= 3; ++q) {
            int64_t W = (pl.P + pl.G - 1) / pl.G;
            if (W > nsimd * q) W = nsimd * q;
            const int64_t base = pl.P / (W * pl.G), rem = pl.P - base * W * pl.G, nbig = (rem + pl.G - 1) / pl.G;
            if (base == 0) W = nbig;
            const int64_t on_simd = (W + nsimd - 1) / nsimd, big_on_simd = std::min(on_simd, (nbig + nsimd - 1) / nsimd);
            const double longest = (double)((base + (big_on_simd > 0 ? 1 : 0)) * pl.NUp + fill), shortest = (double)(base * pl.NUp + fill);
            const double all = (on_simd > big_on_simd && base > 0) ? shortest : longest;      // steps every wave of the SIMD runs
            const double cost = all * (double)on_simd * rate[on_simd] + (longest - all) * (double)big_on_simd * rate[big_on_simd > 0 ? big_on_simd : 1];
            if (best_q == 0 || cost < best * 0.98) { best = cost; best_q = q; }
        }
        if (best_q && knobs().fused_wpc <= 0) {
            waves_per_cu = 4 * best_q;
            max_waves = (int64_t)device_cu_count() * waves_per_cu;
            waves = (pl.P + pl.G - 1) / pl.G;
            if (waves > max_waves) waves = max_waves;
            per = (pl.P + waves * pl.G - 1) / (waves * pl.G);
        }
    }
    if (per > 0x1fffffff / pl.NUp) return SK_ERR_UNSUPPORTED;
    // drawn chunks: small, but never so small that more than three of them are in flight between the producers' frontier and
    // the last lane of the sweep (the kernel keeps a ring of four chunk bases)
    if (pl.P >= 0x7ff00000LL) return SK_ERR_UNSUPPORTED;           // (pair indices are 32-bit inside the kernel)
    const int span = (pl.L - 1 + pl.lag + 24) / pl.NUp + 2;
    int logC = 0;
    while ((span >> logC) + 1 > 3) ++logC;
    const int pct = knobs().fused_q_static > 0 ? (knobs().fused_q_static > 100 ? 100 : knobs().fused_q_static)
                                                : (int)cost_by_name(KIND == 1 ? "fused_static_share_rbf" : "fused_static_share_linear");
    // ... and not smaller than needed either: ~24 draws per lane group balance a launch to a per cent or two, while every
    // chunk costs each lane one look-up of its first pair (the variants that keep edges do that in the macro-step path)
    while ((per * (100 - pct) / 100) >> (logC + 1) >= 24 && logC < 8) ++logC;
    if (prm.queue && waves == max_waves && per >= (8 << logC) && pct < 100) {
        // the launch fills the chip: `pct` per cent of the equal share is dealt out up front, the rest is drawn from the counter
        prm.C0 = (int)(per * pct / 100);
        prm.n_big = 0;
        prm.logC = logC;
        prm.q_first = waves * pl.G * (int64_t)prm.C0;
        if (hipMemsetAsync(prm.queue, 0, sizeof(unsigned long long), s) != hipSuccess) return SK_ERR_LAUNCH;
    } else {
        // as even as whole pairs allow: every lane group takes floor(P / groups) pairs and the first n_big waves one more
        // (128 x 128 symmetric pairs: 8256 = 4096 groups x 2 + 64 -- an equal share of 3 would run a third fewer waves
        // for a third more macro-steps each)
        const int64_t base = pl.P / (waves * pl.G), rem = pl.P - base * waves * pl.G;
        prm.queue = nullptr;
        prm.C0 = (int)base;
        prm.n_big = (int)((rem + pl.G - 1) / pl.G);
        if (base == 0) waves = prm.n_big;                          // no more waves than the pairs need
        prm.logC = logC;      // (the chunks after the first are all empty here, but the ring must not wrap onto the first)
        prm.q_first = pl.P;
        // A launch that fills the chip with a dozen pairs per wave and more, but too few for the queue (a 64-row shard of the
        // headline Gram: 10.7 pairs per wave): shares by wave age rank, sized so that the waves of a SIMD finish together --
        // (share + the skew's fill) proportional to the measured issue shares of the ranks (SK_FUSED_RANK_W overrides them,
        // SK_FUSED_MID=0 restores the equal shares)
        const int nr = waves_per_cu / 4;
        const WaveGroup wg0 = wave_group(pl.lds_bytes, waves, knobs().fused_wpb);
        const int64_t wpr = (int64_t)device_cu_count() * wg0.wpb;
        const int64_t T = (pl.P + wpr * pl.G - 1) / (wpr * pl.G);      // pairs per lane group summed over the ranks of one SIMD slot
        if (knobs().fused_mid != 0 && waves == max_waves && waves_per_cu % 4 == 0 && nr >= 2 && nr <= 4 && wg0.wpb == 4 &&
            wpr * nr == waves && T >= (int64_t)cost_by_name("fused_mid_min_pairs_per_rank") * nr) {
            static constexpr double dflt[5][4] = {{1, 0, 0, 0}, {1, 0, 0, 0}, {0.66, 0.34, 0, 0}, {0.53, 0.30, 0.17, 0}, {0.40, 0.27, 0.19, 0.14}};
            double w[4];
            for (int r = 0; r < 4; ++r) w[r] = dflt[nr][r];
            rank_override(knobs().fused_rank_w, nr, w);
            const double fill = (double)(pl.L - 1 + pl.lag), total = (double)T * pl.NUp + nr * fill;
            int64_t used = 0, cmax = 0;
            unsigned long long pack = 0;
            bool ok = true;
            for (int r = 0; r < nr; ++r) {
                int64_t c = r + 1 < nr ? (int64_t)((w[r] * total - fill) / pl.NUp + 0.5) : T - used;
                if (c < 1 || c > 0xffff || used + c > T - (nr - 1 - r)) { ok = false; break; }
                pack |= (unsigned long long)c << (16 * r);
                cmax = c > cmax ? c : cmax;
                used += c;
            }
            if (ok && (uint64_t)T * (uint64_t)wpr * (uint64_t)pl.G < 0x7ff00000ULL) {
                prm.rk_n = nr;
                prm.rk_wpr = (int)wpr;
                prm.rk_cnt = pack;
                per = cmax;
            }
        }
    }
    if (per > 0x1fffffff / pl.NUp) return SK_ERR_UNSUPPORTED;
    prm.wg = wave_group(pl.lds_bytes, waves, knobs().fused_wpb);
    prm.PPG = (int)per;
    prm.n_steps = 0;
    const size_t lds_block = wave_group_lds(prm.wg);
    if (lds_block > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
    SK_LAUNCH(kern, dim3(wave_group_blocks(prm.wg)), dim3(WAVE * prm.wg.wpb), lds_block, s, prm);
    return check_launch();
}

// Coarse rows per lane of the one-band forward (0: the strip kernels' own, Tile<DY>::RC).  TWICE the strip kernels' rows wherever the
// registers hold them (round 6): a pair then takes half the lanes -- half the skew's fill, twice as many pairs per wave -- and a
// macro-step does twice the cells for 1.4-1.6x the time, because what a macro-step costs is to a good part independent of the rows
// (the neighbour exchange, the pair-start block some lane runs in every step, cursors, ring addresses, the scalar stream:
// profiles/r06_small_launch_pmc.txt).  Same arithmetic per cell in the same order: bit-identical results (104 arrays,
// tools/experiments/r06_rc4.py).  Same-box A/B (profiles/r06_rc4_ab.txt): headline 4.13 -> 3.85 ms, 64-row shard 0.641 -> 0.606,
// linear with edges 5.39 -> 4.78, rbf with edges 9.88 -> 8.94 (1024 x 1024 pairs of 64 points), a 32 + 32-path MMD step 0.303 -> 0.270.
//   dyadic 1: linear (every variant) and rbf of dim <= 4: FOUR rows (pairs of up to 129 points on <= 32 lanes: no full-wave variant);
//             rbf of dim 5..8: two (the four-row form would not fit 256 registers);
//   dyadic 2: the variants that keep edges (linear, rbf of dim <= 4; fp64, default stencil): TWO rows (-6 %; without edges: no gain);
//   dyadic 0: rbf with 8 staged dims: two instead of four (the four-row form spills), everything else the strip kernels' four.
constexpr int fused_rcx(int kind, int dy, int nd, bool edges, bool plain /* fp64 output, default stencil */) {
    return dy == 1 ? ((kind == 0 || nd == 4) ? 4 : 0)
         : dy == 2 ? ((edges && plain && (kind == 0 || nd == 4)) ? 2 : 0)
         : ((kind == 1 && nd == 8) ? 2 : 0);
}
constexpr int fused_nd(int dims, bool plain) { return plain && dims <= 4 ? 4 : 8; }   // the four-dimension variants: fp64, default stencil

template <typename TO, int DY, bool NAIVE, bool FULLWAVE, bool EDGES, int KIND, int ND>
int launch_fused_v(const FusedParams &prm, const FusedPlan &pl, hipStream_t s) {
    constexpr bool PLAIN = !NAIVE && sizeof(TO) == 8;
    constexpr int RCX = fused_rcx(KIND, DY, ND, EDGES, PLAIN);
    if (pl.rcx != RCX) return SK_ERR_UNSUPPORTED;      // (the plan was sized for another row count: host and device rules disagree)
    // the doubled forms serve a pair on at most 32 lanes: their full-wave instances would be dead code
    if constexpr (FULLWAVE && RCX != 0 && RCX == 2 * Tile<DY>::RC) return SK_ERR_UNSUPPORTED;
    // RBF at dyadic 0 with 8 staged dims: the edges' reader exists for the default stencil in fp64 only
    else if constexpr (KIND == 1 && DY == 0 && ND == 8 && EDGES && !PLAIN) return SK_ERR_UNSUPPORTED;
    else return launch_fused_nd<TO, DY, NAIVE, FULLWAVE, EDGES, KIND, ND, RCX>(prm, pl, s);
}

template <typename TO, int DY, bool NAIVE, bool FULLWAVE, bool EDGES, int KIND>
int launch_fused_e(const FusedParams &prm, const FusedPlan &pl, hipStream_t s) {
    // paths of dimension <= 4 skip the four zero dimensions (fp64, default scheme: the variants that are worth their build time)
    if constexpr (!NAIVE && sizeof(TO) == 8) {
        if (prm.dims <= 4) return launch_fused_v<TO, DY, NAIVE, FULLWAVE, EDGES, KIND, 4>(prm, pl, s);
    }
    // (RBF at dyadic 0 with 8 staged dims in fp64 compiles to 280-290 registers with spills in the four-row form: no variant for it --
    // reads left in flight are unsafe there, tools/check_async_hazards.py: scan_pressure -- with TWO rows per lane it fits: fused_rcx)
    return launch_fused_v<TO, DY, NAIVE, FULLWAVE, EDGES, KIND, 8>(prm, pl, s);
}

template <typename TO, int DY, bool NAIVE, bool FULLWAVE, int KIND>
int launch_fused_one(const FusedParams &prm, const FusedPlan &pl, hipStream_t s) {
    if constexpr (sizeof(TO) == 8) {   // the adjoint that consumes the edges exists for fp64, d = 0..2
        if (prm.edges) return launch_fused_e<TO, DY, NAIVE, FULLWAVE, true, KIND>(prm, pl, s);
    }
    if (prm.edges) return SK_ERR_UNSUPPORTED;
    return launch_fused_e<TO, DY, NAIVE, FULLWAVE, false, KIND>(prm, pl, s);
}

template <typename TO, int DY, int KIND>
int launch_fused_dy(const FusedParams &prm, const FusedPlan &pl, hipStream_t s) {
    const bool full = prm.logL == 6;
    if (prm.naive)
        return full ? launch_fused_one<TO, DY, true, true, KIND>(prm, pl, s) : launch_fused_one<TO, DY, true, false, KIND>(prm, pl, s);
    return full ? launch_fused_one<TO, DY, false, true, KIND>(prm, pl, s) : launch_fused_one<TO, DY, false, false, KIND>(prm, pl, s);
}

// KIND 0: dXr [A][Mrows][8] / dYt [Bn][8][Ncp] are path differences; KIND 1: the same layouts hold the path points.
// SK_ERR_UNSUPPORTED outside the kernel's scope.
template <typename TO, int KIND>
int launch_fwd_fused(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                     double inv_sigma, TO *out, double *strip_edges, void *queue, hipStream_t s, int tri = 0, const int64_t *loss = nullptr) {
    // (tri = 2: the caller -- sk_solve_fwd_loss_f64 -- has checked that the pair table lies behind dYt)
    if (tri == 1 && (strip_edges || A != B || g.P != A * (A + 1) / 2)) return SK_ERR_UNSUPPORTED;
    // the loss layout: loss = {tri_n, tri_off}; A rows against the B paths of the one batch, then the strict triangle of tri_n of them
    if (tri == 2 && (!loss || B <= 0 || loss[0] < 0 || loss[1] != A || loss[0] + loss[1] > B || A > B ||
                     g.P != A * B + (loss[0] > 1 ? loss[0] * (loss[0] - 1) / 2 : 0)))
        return SK_ERR_BAD_ARG;
    if (tri == 2 && (A > 0x7fff || loss[0] > 0x7fff || A * B >= 0x7ff00000LL)) return SK_ERR_UNSUPPORTED;   // (packed into prm.tri)
    const int DY = g.dyadic;
    if (DY > 2 || D < 1 || D > FD) return SK_ERR_UNSUPPORTED;
    // (RBF at dyadic 0 beyond the four-dimension fp64 default-stencil variant: the two-row form, see launch_fused_e)
    const bool four_dim = !g.naive && sizeof(TO) == 8 && D <= 4;
    const bool rbf0_two_rows = KIND == 1 && DY == 0 && !four_dim;
    if (rbf0_two_rows && strip_edges && (g.naive || sizeof(TO) != 8)) return SK_ERR_UNSUPPORTED;
    // linear: one unit = two increment columns.  RBF: one unit = two NODE columns, and the sweep of a pair's last unit
    // reads one node column of the following unit, which therefore has to exist as padding inside the pair's stream;
    // likewise the lanes of a pair must cover M node rows, not M - 1 increment rows
    const int NU = KIND == 1 ? (g.Nc + 2) / 2 : (g.Nc + 1) / 2;
    const int rows = KIND == 1 ? g.Mc + 1 : g.Mc;
    const int NUp = (NU + LINE_UNITS - 1) / LINE_UNITS * LINE_UNITS;
    if (Ncp < NUp * 2 || (Ncp & 1)) return SK_ERR_UNSUPPORTED;
    // rows per lane: the strip kernels' own or twice that (fused_rcx, above -- the same rule the variant dispatch applies)
    const int nd_v = fused_nd(D, !g.naive && sizeof(TO) == 8);
    const int rcx = fused_rcx(KIND, DY, nd_v, strip_edges != nullptr, !g.naive && sizeof(TO) == 8);
    const int RC = rcx ? rcx : (DY == 0 ? 4 : DY == 1 ? 2 : 1);
    const bool doubled = rcx == 2 * (DY == 0 ? 4 : DY == 1 ? 2 : 1);
    int logL = 3;
    while (logL < 6 && (RC << logL) < rows) ++logL;
    if (rbf0_two_rows && strip_edges) {
        // the edges are read in the strip layout, whose padded rows (K constant along the zero increments of the padding) must all
        // be WRITTEN: this variant's lanes have to cover them -- twice the strip kernels' lanes
        const Strip st = strip_geom(rbf_edge_geom(g), 8);
        if (!st.ok || st.nb != 1 || st.logL > 5) return SK_ERR_UNSUPPORTED;
        logL = st.logL + 1;
    }
    const int L = 1 << logL;
    if (L * RC < rows) return SK_ERR_UNSUPPORTED;   // more than one band per pair
    if (Mrows < L * RC) return SK_ERR_UNSUPPORTED;
    const int G = WAVE / L;
    const int JMAX = (L + NUp - 1) / NUp;
    const int nd = (!g.naive && sizeof(TO) == 8 && D <= 4) ? 4 : 8;   // the variant launch_fused_e picks
    const size_t lds_bytes = (size_t)G * (((L >> 3) + 2) * y_slab_pitch(nd) + X_SLOTS * JMAX * RC * 8 * x_row_bytes(nd));   // (a multiple of 256: the y reads rely on 256-byte aligned slices)
    if (lds_bytes > 160 * 1024) return SK_ERR_UNSUPPORTED;

    int waves_per_cu = (int)((160 * 1024) / lds_bytes);
    const int wpc_env = knobs().fused_wpc;
    // measured on the headline (512 x 512 pairs, len 128, dim 8, d = 1) with four-wave workgroups, i.e. the same number of
    // waves on every SIMD: 8 waves/CU 5.66 ms, 12: 5.30, 13: 6.65 (single-wave workgroups, whose placement is uneven:
    // 8: 7.58, 10: 6.31, 12: 6.77).  d = 0 (four coarse rows per lane): 4: 4.60 ms, 8: 3.33, 12: 3.32; d = 2: 8: 3.31,
    // 12: 3.07 on 512 x 512 pairs of length 64.  SK_FUSED_WPC overrides.
    // (d = 2 with four-dimension slabs: the RBF variant that keeps edges has 112 VGPRs and room for a fourth wave per SIMD --
    // C4's training step 359.6 -> 357.2 ms; the variants above 128 VGPRs stay at three through by_regs)
    const int cap = wpc_env > 0 ? 16 : (DY == 0 ? 8 : (DY == 2 && nd == 4) ? 16 : 12);
    if (waves_per_cu > cap) waves_per_cu = cap;
    if (wpc_env > 0) waves_per_cu = waves_per_cu < wpc_env ? waves_per_cu : wpc_env;
    else if (waves_per_cu > 4) waves_per_cu &= ~3;   // whole four-wave workgroups
    if (waves_per_cu < 1) waves_per_cu = 1;
    const FusedPlan pl{g.P, G, NUp, L, KIND == 1 ? 2 : 0, lds_bytes, waves_per_cu, rcx};

    FusedParams prm;
    prm.dXr = dXr; prm.dYt = dYt; prm.out = out; prm.edges = strip_edges; prm.P = g.P; prm.B = B;
    prm.Mrows = Mrows; prm.Ncp = Ncp; prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = NUp; prm.logL = logL;
    prm.inv_sigma = inv_sigma;
    prm.dims = D;
    prm.rk_n = 0; prm.rk_wpr = 1; prm.rk_cnt = 0;
    prm.tri = tri == 2 ? (2 | ((int)A << 2) | ((int)loss[0] << 17)) : tri;
    prm.queue = (unsigned long long *)queue;
    prm.e_NUp = NUp;
    prm.e_L = L;
    if (strip_edges) {   // the layout sk_solve_adj_* reads (for the linear kernel it is this kernel's own)
        const Strip st = strip_geom(KIND == 1 ? rbf_edge_geom(g) : g, 8);
        if (!st.ok || st.nb != 1 || (st.RC != RC && !rbf0_two_rows && !doubled) || ((1 << st.logL) * st.RC) % RC) return SK_ERR_UNSUPPORTED;
        prm.e_NUp = st.NUp;
        prm.e_L = ((1 << st.logL) * st.RC) / RC;      // e_L x (this kernel's rows per lane) = the strip layout's padded rows
        if (prm.e_L > L) return SK_ERR_UNSUPPORTED;   // (every padded row of the strip layout must be some lane's)
    }
    prm.u_f = (g.Nc - 1) / 2;
    prm.lam_f = ((g.Mc - 1) / RC) % L;
    prm.sel_f = ((g.Mc - 1) % RC) * 2 + (g.Nc - 1) % 2;
    prm.naive = g.naive;
    (void)A;
    switch (DY) {
        case 0: return launch_fused_dy<TO, 0, KIND>(prm, pl, s);
        case 1: return launch_fused_dy<TO, 1, KIND>(prm, pl, s);
        default: return launch_fused_dy<TO, 2, KIND>(prm, pl, s);
    }
}

}  // namespace

template <typename TO>
int launch_fwd_fused_linear(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                            TO *out, double *strip_edges, void *queue, hipStream_t s, int tri, const int64_t *loss) {
    return launch_fwd_fused<TO, 0>(dXr, dYt, A, B, Mrows, Ncp, D, g, 0.0, out, strip_edges, queue, s, tri, loss);
}
// Xr [A][Mrows][8]: path points x_p (zero rows / dims beyond M / D); Yt [Bn][8][Ncp]: y_q, dimension-major
template <typename TO>
int launch_fwd_fused_rbf(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                         double inv_sigma, TO *out, double *strip_edges, void *queue, hipStream_t s, int tri, const int64_t *loss) {
    return launch_fwd_fused<TO, 1>(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, out, strip_edges, queue, s, tri, loss);
}

template int launch_fwd_fused_linear<double>(const double *, const double *, int64_t, int64_t, int, int, int, const Geom &, double *,
                                             double *, void *, hipStream_t, int, const int64_t *);
template int launch_fwd_fused_linear<float>(const double *, const double *, int64_t, int64_t, int, int, int, const Geom &, float *,
                                            double *, void *, hipStream_t, int, const int64_t *);
template int launch_fwd_fused_rbf<double>(const double *, const double *, int64_t, int64_t, int, int, int, const Geom &, double, double *,
                                          double *, void *, hipStream_t, int, const int64_t *);
template int launch_fwd_fused_rbf<float>(const double *, const double *, int64_t, int64_t, int, int, int, const Geom &, double, float *,
                                         double *, void *, hipStream_t, int, const int64_t *);

}  // namespace sk
