// sk_wave_fused_mb.hip -- forward solver with the static kernel fused in, for pairs that need SEVERAL BANDS of a wavefront
// (long paths: M - 1 > 64 RC) and for path dimensions up to 16.  KIND 0: LinearKernel, KIND 1: RBFKernel.
//
// One 64-lane wavefront sweeps one pair at a time (L = 64, persistent over its share of the pairs).  Lane `lam` owns RC
// coarse rows of a band of 64 RC rows and runs `lam` macro-steps behind lane 0 (skewed row strips, DPP neighbour exchange:
// sk_wave.hip); the bands of a pair and then the next pair follow each other in one virtual column stream, so the skew
// fills once per wave.  As in sk_wave_fused.hip nothing of size pairs x M x N exists: the increments of a macro-step's
// RC x 2 coarse cells are formed from the paths, which the wave streams through two small LDS rings (y: slabs of 8 units
// x FD dims over the virtual column stream, parity-swizzled; x: the rows of the 8 lanes that start a band during the next
// 8 macro-steps).  What is new here:
//
//   * Bands.  The bottom lane's last fine row is the top boundary of the next band, needed NUp - 63 macro-steps later by
//     lane 0.  It travels through a per-wave row in GLOBAL memory (caller's workspace, L2-resident) instead of LDS, where
//     a whole row of the pair (16 KB at 2044 fine columns) would cost the kernel its occupancy -- but never with a memory
//     operation per macro-step: lane 63 stages its entries in LDS, every 8 macro-steps the wave writes the finished chunk
//     of 8 units through to L2 with one coalesced store, and lane 0's chunk of the coming window arrives by LDS-DMA together
//     with the path slabs (one window ahead, past the L1).  All global traffic is issued at window boundaries and waited
//     for 8 macro-steps later; the first version loaded lane 0's entry from global memory every step and ran at the
//     latency of that load (1.2 us per macro-step).
//   * RBF nodes "from above".  A lane evaluates the nodes G[p][q] = exp(-|x_p - y_q|^2 / sigma) of the BOTTOM node rows of
//     its RC coarse rows (rows r0+1 .. r0+RC) one unit ahead of the block sweep, and takes the node row r0 above its first
//     coarse row from the lane above -- which is one macro-step ahead, so the values exist already and no lag between lanes
//     is needed (the single-band kernel takes the row BELOW from the lane below and runs its sweep two units behind).  Lane
//     0 gets that row from the previous band's bottom lane through the same global row mechanism, and in band 0 -- node
//     row 0 of the pair -- evaluates it itself: a divergent branch that the wave pays for 1 / nb of the time.
//   * FD = 16 dimensions (zero-padded), two LDS-DMA instructions per y slab.
//
// Scope: dyadic <= 2, path dim <= 16, either stencil, any M and N.  A pair's stream has NUp >= 80 units per band (a chunk must be
// flushed and acknowledged before the window that consumes it is fetched: NUp - 63 >= 17 macro-steps; and at most one row unit may
// start per window): second paths shorter than ~160 points are swept with padding units behind them (wasted steps, same result --
// the caller may swap the arguments when the first path is the longer one: the kernel is symmetric).
// Replaces, for LinearKernel / RBFKernel on long or wide paths, static_kernels.py:26-33 / :58-73 + sigkernel.py:362-382.
#include "sk_wave_common.h"

namespace sk {
namespace {

constexpr int MB_L = WAVE;   // lanes per pair
constexpr int MB_X_SLOTS = 2;

struct FusedMbParams {
    const double *Xr;   // [A][Mrows][FD]: KIND 0: s^2 (x[p+1]-x[p]);  KIND 1: x[p]; zero rows / dims beyond the path
    const double *Yt;   // [Bn][FD][Ncp]: KIND 0: y[q+1]-y[q];  KIND 1: y[q]; dimension-major, zero-padded
    void *out;          // [P]
    double *edges;      // EDGES: [P][nb NUp S + nb 64 R] of the PADDED grid (whose padding carries no increments: K[MMp][j] = K[MM][min(j, NN)]
                        // and likewise down the column): the rows K[MMp - b 64 R][1..NNp], b = 0..nb-1 -- the terminal row and the bottom
                        // row of every other band, from which sk_rbf_adjoint_fused_mb_f64 restarts its backward recompute of K band by
                        // band -- then the terminal column K[1..MMp][NNp]
    double *ws;         // per wave: [NUp + 8][E] band-boundary row + a chunk of ones, E = S doubles of K (+ 2 node values, KIND 1) per unit
    int64_t P, B;       // B > 0: Gram, pair p = (p / B, p % B); B == 0: paired
    int Mrows, Ncp, Mc, Nc, NUp, nb, PPW, n_steps;
    int u_f, lam_f, band_f, sel_f;
    double inv_sigma;
    int64_t ws_stride;  // doubles per wave
    WaveGroup wg;
    // the wave's stream of pairs (as in sk_wave_fused.hip): C0 pairs fixed per wave, then one pair at a time drawn from `queue`
    unsigned long long *queue;   // the launch's counter (zeroed by the launcher; inside the workspace), nullptr: equal static shares
    int64_t q_first;
    int C0;
    int naive;          // _naive_solver stencil (cython_backend.pyx:114): the g^2 / 12 terms drop out of both coefficients
    // SPLIT mode (few pairs of long paths: fewer pairs than resident waves): the BANDS of a pair run on different waves.  A stream
    // position is then one ITEM = (band, pair), item id = band Pn + pair, handed out in increasing order by the launch's counter
    // (band-major: every pair's band 0 first); the kernel runs with nb = 1 -- every position is one row unit -- and takes a
    // position's TRUE band from the item.  Band b's bottom row goes to the item's own row in `rows` ([items][NUp][E], written
    // through), its progress (chunks of 8 units flushed) to `prog`; the wave that sweeps band b + 1 of the pair trails it through
    // that counter, a few chunks behind.  An item only ever waits for an item with a SMALLER id, which a running wave holds
    // (it drew the ticket) or has finished: no deadlock whatever is resident.
    int split;
    int64_t Pn;          // pairs of the launch (P keeps the number of ITEMS in split mode)
    int nb_true;         // bands per pair
    double *rows;        // [items][row_stride]
    int64_t row_stride;  // doubles per item row (NUp E)
    unsigned *prog;      // [items] chunks flushed, zeroed by the launcher
    unsigned long long *status;   // the LAST 8 bytes of the caller's workspace: items whose wait for the band above gave up (zeroed by the launcher)
    int lead;            // chunks a band must be ahead of the band below it before that band's next window is fetched (the counter
                         // is then polled once per `lead` windows, not once per window); <= MB_LEAD
};

constexpr int MB_LEAD = 8;   // the largest `lead` (SK_FUSEDMB_LEAD; default 4): what the launcher's length limit is sized for

// 16-byte asynchronous global load past the L1 (sc0 sc1: the line was written by another lane of this wave a few
// macro-steps ago and must come from L2).  Destination handling as in sk_wave_common.h (load_async / async_wait).
__device__ __forceinline__ void async_begin2(d2_t &t) { asm volatile("" : "=v"(t)); }
__device__ __forceinline__ void load_async_x4(d2_t &dst, const double *p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(dst) : "v"(p) : "memory");
}
template <int BYTE_OFF>
__device__ __forceinline__ void load_async_x4_at(d2_t &dst, const double *p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1" : "=v"(dst) : "v"(p), "n"(BYTE_OFF) : "memory");
}
// 16-byte store written through to L2 (device scope): another lane of this wave loads it from there a few macro-steps
// later; a wave-scope store may be acknowledged by the L1 first
__device__ __forceinline__ void store_through(double *p, d2_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait2(d2_t (&o)[1], d2_t (&t)[1]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "=v"(o[0]) : "0"(t[0]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait2(d2_t (&o)[4], d2_t (&t)[4]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3])
                 : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait2(d2_t (&o)[2], d2_t (&t)[2]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "=v"(o[0]), "=v"(o[1]) : "0"(t[0]), "1"(t[1]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait2(d2_t (&o)[3], d2_t (&t)[3]) {
    asm volatile("s_waitcnt vmcnt(%6)" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]) : "0"(t[0]), "1"(t[1]), "2"(t[2]), "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void async_wait2(d2_t (&o)[5], d2_t (&t)[5]) {
    asm volatile("s_waitcnt vmcnt(%10)" : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4])
                 : "0"(t[0]), "1"(t[1]), "2"(t[2]), "3"(t[3]), "4"(t[4]), "n"(VM) : "memory");
}

// N consecutive 16-byte LDS reads at a stride of 256 bytes from two interleaved bases (even / odd dimension rows of a
// parity-swizzled y slab, see sk_wave_fused.hip), one wait
template <int ND>
__device__ __forceinline__ void lds_read_dims(d2_t (&v)[ND], unsigned a_even, unsigned a_odd);
template <>
__device__ __forceinline__ void lds_read_dims<8>(d2_t (&v)[8], unsigned a_even, unsigned a_odd) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\t"
                 "ds_read_b128 %2, %8 offset:256\n\tds_read_b128 %3, %9 offset:256\n\t"
                 "ds_read_b128 %4, %8 offset:512\n\tds_read_b128 %5, %9 offset:512\n\t"
                 "ds_read_b128 %6, %8 offset:768\n\tds_read_b128 %7, %9 offset:768\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(a_even), "v"(a_odd) : "memory");
}
template <>
__device__ __forceinline__ void lds_read_dims<16>(d2_t (&v)[16], unsigned a_even, unsigned a_odd) {
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
// FD consecutive doubles (one x row), one wait
template <int ND>
__device__ __forceinline__ void lds_read_xrow(double (&x)[ND], unsigned a) {
    static_assert(ND == 8 || ND == 16, "");
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

// E consecutive doubles of a top lane's boundary entry (E = 2, 4, 6, 8 or 10), issued WITHOUT a wait: the reads complete
// at the macro-step's first lgkmcnt(0) (the y read right behind them); lds_pend_take hands the values over afterwards.
template <int NP>
__device__ __forceinline__ void lds_read_pend(d2_t (&t)[NP], unsigned a);
template <> __device__ __forceinline__ void lds_read_pend<1>(d2_t (&t)[1], unsigned a) {
    asm volatile("ds_read_b128 %0, %1" : "=&v"(t[0]) : "v"(a) : "memory");
}
template <> __device__ __forceinline__ void lds_read_pend<2>(d2_t (&t)[2], unsigned a) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(t[0]), "=&v"(t[1]) : "v"(a) : "memory");
}
template <> __device__ __forceinline__ void lds_read_pend<3>(d2_t (&t)[3], unsigned a) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]) : "v"(a) : "memory");
}
template <> __device__ __forceinline__ void lds_read_pend<4>(d2_t (&t)[4], unsigned a) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]) : "v"(a) : "memory");
}
template <> __device__ __forceinline__ void lds_read_pend<5>(d2_t (&t)[5], unsigned a) {
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\tds_read_b128 %3, %5 offset:48\n\t"
                 "ds_read_b128 %4, %5 offset:64"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]) : "v"(a) : "memory");
}
template <int NP>
__device__ __forceinline__ void lds_pend_take(d2_t (&o)[NP], d2_t (&t)[NP]) { async_wait2<63>(o, t); }   // vmcnt(63): no wait at all

// Y32: the y ring holds fp32 points (fp32 inputs: exactly the caller's values), two dimensions per 16-byte unit --
// {dim 2j col 0, dim 2j col 1, dim 2j+1 col 0, dim 2j+1 col 1} -- so a slab has FD / 2 rows and the ring half the bytes: at
// FD = 16 that is what lets two waves share a SIMD.  Everything is still computed in fp64 (one v_cvt per value).  The slab
// carries one more row, |y_q|^2 in fp64, and the squared distance is formed like the reference does
// (static_kernels.py:70-73), |x|^2 + |y|^2 - 2<x,y>: FD FMAs per node instead of 2 FD operations.  With fp32 inputs the
// products are exact in fp64 and the cancellation costs nothing the inputs could resolve; the fp64-ring variants keep the
// direct sum over (x - y)^2.  The slab pitch (9 x 128 bytes) is an odd multiple of 128, so lanes 8 apart (same unit of
// neighbouring slabs) hit different halves of the bank row without the parity swizzle of the fp64 ring.
// RCX: coarse rows per lane; the strip kernels' Tile<DY>::RC, except for the RBF edges at dyadic 0, which are kept for
// k_adj_fused_rbf_mb<0, 2, ..> (two rows per lane there: four would need > 256 registers) in ITS bands of 128 rows
// SPLIT: the bands of a pair on several waves (FusedMbParams::split; a variant of its own: its ring of item bands, row pointers and
// progress bookkeeping would otherwise sit in the scalar registers of every launch -- measured as a runtime flag: C5 80.9 -> 83.5 ms)
template <typename TO, int DY, bool Y32, int KIND, int FD, bool EDGES, int RCX = Tile<DY>::RC, bool SPLIT = false>
__global__ __launch_bounds__(4 * WAVE) void k_fwd_fused_mb(const FusedMbParams prm) {
    // The _naive_solver stencil (k10 + k01)(1 + g/2) - k00 is the default one with c_12 = 0: a = 1 + g/2 + 0 g^2 and b = 1 - 0 g^2 = 1
    // exactly (finite g), so diag * b = diag -- a launch-time constant instead of a second set of kernel variants
    constexpr bool NAIVE = false;
    constexpr int FDY = Y32 ? FD / 2 : FD;   // rows of a y slab
    constexpr bool RBF = KIND == 1;
    constexpr int LAG = RBF ? 1 : 0;   // macro-steps by which the block sweep trails the node evaluation
    constexpr int CW = 2;
    constexpr int RC = RCX, R = RCX << DY, S = CW << DY, r = 1 << DY;
    constexpr int L = MB_L;
    constexpr int XROW = FD * 8;              // bytes of one x row
    constexpr int XSLAB = 8 * RC * XROW;      // the rows of 8 lanes
    constexpr int YSLAB = FDY * 128 + (Y32 ? 128 : 0);   // FDY rows of 8 units (+ Y32: one row of |y|^2, fp64)
    constexpr int NSLAB = L / 8 + 2;
    constexpr int NDMA_Y = YSLAB / 1024, NDMA_X = (XSLAB + 1023) / 1024;
    // band boundary: per unit of the stream S doubles of K (bottom fine row of the band being left) and, RBF, the 2 node
    // values under it -- E doubles; 8 units make a chunk
    constexpr int E = S + (RBF ? 2 : 0), NP = E / 2, CHUNK = 8 * E * 8;
    extern __shared__ __attribute__((aligned(16))) char lds_block[];
    char *lds;
    const int64_t wave_id = wave_slot(prm.wg, lds_block, lds);
    if (wave_id < 0) return;
    const unsigned lds0 = lds_offset(lds);
    // LDS map of a wave: [y ring: NSLAB slabs][x ring: 2 slabs][boundary chunks in: 2 slots][boundary chunk out]
    //                    [RBF: x row 0 of the pair, 2 slots of XROW bytes]
    constexpr unsigned X_BASE = NSLAB * YSLAB, BI_BASE = X_BASE + MB_X_SLOTS * XSLAB, BO_BASE = BI_BASE + 2 * CHUNK,
                       T_BASE = BO_BASE + CHUNK;

    const int lam = threadIdx.x & (WAVE - 1);
    const int NUp = prm.NUp, nb = prm.nb;
    const double sc = 1.0 / (double)(1 << (2 * DY));
    const double c_half = 0.5 * sc, c_12 = prm.naive ? 0.0 : sc * sc / 12.0;
    const double two_inv_sigma = 2.0 * prm.inv_sigma;
    const bool is_top = lam == 0, is_bot = lam == L - 1;

    // ---- cursors: (u, band, ps) = where this lane's node evaluation / path reads are; (uk, bandk, psk) = its block sweep,
    //      LAG macro-steps behind.  Virtual unit v = t - lam over the stream [pair][band][unit].
    int u, band, ps, uk, bandk, psk;
    {
        int sig = floor_div(-lam, NUp);
        u = -lam - sig * NUp;
        ps = floor_div(sig, nb);
        band = sig - ps * nb;
        sig = floor_div(-lam - LAG, NUp);
        uk = -lam - LAG - sig * NUp;
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
    // ---- the wave's stream of pairs: positions 0 .. C0-1 are pairs wave_id C0 + i; every later position is one pair drawn from the
    // launch's counter (a pair is nb NUp >= 160 macro-steps long, the lanes' skew 63: at most two pairs are in flight, plus
    // the producers' look-ahead -- a ring of four).  Pair indices are 32-bit; NOPAIR = none.
    constexpr unsigned NOPAIR = 0xffffffffu;
    const unsigned P32 = (unsigned)prm.P;
    const int C0 = prm.C0;
    constexpr bool split = SPLIT;
    const unsigned Pn32 = (unsigned)prm.Pn;
    const unsigned base0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(wave_id * C0));
    unsigned cb1 = NOPAIR, cb2 = NOPAIR, cb3 = NOPAIR, cb0 = NOPAIR;   // drawn pair of position C0 + j in cb[j & 3]
    int tb0 = 0, tb1 = 0, tb2 = 0, tb3 = 0;                            // split mode: the TRUE band of that position's item
    int have = 0;                     // drawn positions known so far
    int t_end = 0x7fffffff;
    const int tail = (MB_L - 1) + (KIND == 1 ? 1 : 0);
    auto stream_pair = [&](int i) __attribute__((always_inline)) -> unsigned {
        if (i < 0) return NOPAIR;
        if (i < C0) { const unsigned p = base0 + (unsigned)i; return p < P32 ? p : NOPAIR; }
        const int kk = (i - C0) & 3;
        return (cb0 & -(unsigned)(kk == 0)) | (cb1 & -(unsigned)(kk == 1)) | (cb2 & -(unsigned)(kk == 2)) | (cb3 & -(unsigned)(kk == 3));
    };
    auto stream_band = [&](int i) __attribute__((always_inline)) -> int {    // split mode (C0 = 0): 0 for positions without an item
        if (i < 0) return 0;
        const int kk = i & 3;
        return (tb0 & -(int)(kk == 0)) | (tb1 & -(int)(kk == 1)) | (tb2 & -(int)(kk == 2)) | (tb3 & -(int)(kk == 3));
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
            if (split) {   // item -> (true band, pair); the ring keeps the pair, the band beside it
                const int bq = b == NOPAIR ? 0 : (int)(b / Pn32);
                if (b != NOPAIR) b -= (unsigned)bq * Pn32;
                tb0 = __builtin_amdgcn_readfirstlane((tb0 & ~(int)m0) | (bq & (int)m0));
                tb1 = __builtin_amdgcn_readfirstlane((tb1 & ~(int)m1) | (bq & (int)m1));
                tb2 = __builtin_amdgcn_readfirstlane((tb2 & ~(int)m2) | (bq & (int)m2));
                tb3 = __builtin_amdgcn_readfirstlane((tb3 & ~(int)m3) | (bq & (int)m3));
            }
            cb0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb0 & ~m0) | (b & m0)));
            cb1 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb1 & ~m1) | (b & m1)));
            cb2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb2 & ~m2) | (b & m2)));
            cb3 = (unsigned)__builtin_amdgcn_readfirstlane((int)((cb3 & ~m3) | (b & m3)));
            have += 1;
        }
    };
    const int my_uf = lam == prm.lam_f ? prm.u_f : -1;
    const unsigned my_x = lds0 + X_BASE + (unsigned)((lam & 7) * RC * XROW);
    int tband = 0;   // split mode: the true band of this lane's path-read cursor (`band` itself stays 0: nb = 1)

    auto split_b = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p % (uint32_t)prm.B);   // (32-bit: the launcher refuses P >= 2^31 - 2^20, and B <= P)
    };
    auto split_a = [&](int64_t p) -> int64_t {
        if (prm.B <= 0) return p;
        return (int64_t)((uint32_t)p / (uint32_t)prm.B);
    };

    // the wave's boundary row in global memory: [NUp][E] doubles, position = the producing / consuming lane's unit u
    double *const wsrow = prm.ws + wave_id * prm.ws_stride;

    // ---- producers (wave-uniform control), once per window of 8 macro-steps ------------------------------------------
    // y slab s = virtual units [8s, 8s+8): unit offset y_u0 in the row, of pair-in-wave y_pi (every band re-reads its pair's y)
    int y_pi = 0, y_band = 0, y_u0 = 0, y_slot = 0, y_par = 0;
    auto issue_y = [&]() {
        ensure(y_pi);
        const unsigned spy = stream_pair(y_pi);
        const int64_t p = spy == NOPAIR ? 0 : (int64_t)spy;    // past the end: fetch something valid, never consumed
        const int64_t b = split_b(p);
        if constexpr (Y32) {
            const double *row0 = prm.Yt + b * (FDY + 1) * (int64_t)prm.Ncp;
            __builtin_amdgcn_global_load_lds(row0 + ((lam >> 3) * (int64_t)prm.Ncp + (int64_t)(y_u0 + (lam & 7)) * 2),
                                             (lds_void *)(lds + y_slot * YSLAB), 16, 0, 0);
            if (lam < 8)   // the |y|^2 row: 128 bytes
                __builtin_amdgcn_global_load_lds(row0 + (FDY * (int64_t)prm.Ncp + (int64_t)(y_u0 + lam) * 2),
                                                 (lds_void *)(lds + y_slot * YSLAB + FDY * 128), 16, 0, 0);
        } else {
#pragma unroll
            for (int c = 0; c < NDMA_Y; ++c) {
                const int krow = (c * 8 + (lam >> 3)) ^ (y_par & 1);     // odd slabs: dimension rows swapped in pairs
                const double *src = prm.Yt + ((b * FDY + krow) * (int64_t)prm.Ncp + (int64_t)(y_u0 + (lam & 7)) * 2);
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
    // x slab for the lanes x_lam0 .. x_lam0+7 that start row unit (x_pi, x_band) during the window (NUp > L: at most one row
    // unit starts per window); KIND 1 owns the node rows r0+1 .. r0+RC, KIND 0 the rows r0 ..;  and the boundary chunk lane 0
    // consumes during the window: positions x_lam0 .. x_lam0+7 of the wave's global row (lane 0's unit IS the window's offset)
    int x_pi = 0, x_band = 0, x_lam0 = 0, x_slot = 0;
    auto issue_x = [&]() {
        ensure(x_pi);
        const unsigned spx = stream_pair(x_pi);
        const int64_t p = spx == NOPAIR ? 0 : (int64_t)spx;
        const int64_t a = split_a(p);
        const int lamj = x_lam0 < L ? x_lam0 : 0;     // nobody starts: fetch something valid
        const int xb = split ? stream_band(x_pi) : x_band;     // the band whose rows the window's row unit sweeps
        const char *src = reinterpret_cast<const char *>(prm.Xr + (a * prm.Mrows + (int64_t)(xb * L + lamj) * RC + (RBF ? 1 : 0)) * FD);
        char *dst = lds + X_BASE + x_slot * XSLAB;
#pragma unroll
        for (int c = 0; c < NDMA_X; ++c) {
            const int off = c * 1024 + lam * 16;
            if (XSLAB % 1024 == 0 || off < XSLAB)   // partial last piece: the lanes past its end are masked out of the DMA
                __builtin_amdgcn_global_load_lds(src + off, (lds_void *)(dst + c * 1024), 16, 0, 0);
        }
        if (RBF) {   // node row 0 of the pair the window's row unit belongs to (lane 0 evaluates it itself in band 0)
            const char *s0 = reinterpret_cast<const char *>(prm.Xr + a * prm.Mrows * FD) + lam * 16;
            if (lam < XROW / 16)   // XROW bytes only: the other lanes are masked out of the DMA
                __builtin_amdgcn_global_load_lds(s0, (lds_void *)(lds + T_BASE + (x_pi & 1) * XROW), 16, 0, 0);
        }
        if (lam < CHUNK / 16) {   // past the L1 (aux = sc0 | sc1): the chunk was written through to L2 by this wave's flush
            // K[0][.] = 1 is the top boundary of band 0: entries whose K part belongs to a sweep in band 0 come from the ones
            // chunk behind the row, so that the consumer needs no select.  The sweep trails the window's row unit by LAG
            // units: with LAG = 1 entry 0 of a row unit's first window still belongs to the PREVIOUS band's sweep.
            // (split mode: the row of the item one band up -- item id - Pn --, polled for by wait_producer; the K part of a row's
            // entry 0 belongs to the sweep of the PREVIOUS position's last unit, which is padding there (the launcher sees to it): any
            // finite value serves)
            const bool ones_rest = xb == 0;
            const bool ones_first = split || LAG == 0 || x_lam0 > 0 ? ones_rest : x_band == (nb > 1 ? 1 : 0);
            const int piece = lam * 2;                           // doubles from the chunk's start
            const bool first_k = piece < S;                      // K part of entry 0
            const bool ones = first_k ? ones_first : ones_rest;  // (node parts of band-0 windows are never used)
            const double *row = split ? prm.rows + ((int64_t)(xb - 1) * prm.Pn + (spx == NOPAIR ? 0 : (int64_t)spx)) * prm.row_stride : wsrow;
            const double *sb = (ones ? wsrow + (int64_t)NUp * E : row + (int64_t)x_lam0 * E) + piece;
            __builtin_amdgcn_global_load_lds(sb, (lds_void *)(lds + BI_BASE + x_slot * CHUNK), 16, 0, 17);
        }
        x_slot ^= 1;
        x_lam0 += 8;
        if (x_lam0 == NUp) {
            x_lam0 = 0;
            x_band += 1;
            if (x_band == nb) { x_band = 0; x_pi += 1; }
        }
    };
    // the bottom lane's staged chunk (its units f_pos .. f_pos+7) -> the wave's global row, written through to L2
    int f_pos = -(L - 1) - 7 + 7;   // unit at which the next chunk of the bottom lane starts; < 0: lane 63 has not started
    {
        // lane 63's stream unit at macro-step t is t - 63; its first whole chunk [0, 8) is complete at t = 70
        f_pos = 0;
    }
    // split mode: stream positions [bad_lo, bad_hi] of this wave whose producer's counter did not move for seconds -- THEIR results
    // (and, through the poisoned boundary rows, those of the bands below them) are NaN; the wave's other items are not touched, and
    // the launch's status word counts the give-ups (a stall must not show up as unexplained NaNs of unrelated pairs)
    int bad_lo = 0x7fffffff, bad_hi = -1;
    int f_i = 0;                 // split mode: the stream position of the bottom lane's row unit
    unsigned *pub_ptr = nullptr; // ... the progress counter a flush has yet to publish (after the wait for its stores), and the count
    unsigned pub_cnt = 0;
    auto flush_chunk = [&]() {
        double *frow = wsrow;
        if (split) {
            const unsigned fp = stream_pair(f_i);
            if (fp != NOPAIR) {
                const int64_t item = (int64_t)stream_band(f_i) * prm.Pn + fp;
                frow = prm.rows + item * prm.row_stride;
                pub_ptr = prm.prog + item;
                pub_cnt = (unsigned)(f_pos >> 3) + 1u;
            }
        }
        if (lam < CHUNK / 16) {
            d2_t v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds0 + BO_BASE + (unsigned)(lam * 16)) : "memory");
            if (SPLIT && f_i >= bad_lo && f_i <= bad_hi) v = d2_t{__longlong_as_double(0x7ff8000000000000LL), __longlong_as_double(0x7ff8000000000000LL)};   // poison the bands below
            store_through(frow + (int64_t)f_pos * E + lam * 2, v);
        }
        f_pos += 8;
        if (f_pos == NUp) { f_pos = 0; f_i += 1; }
    };
    // split mode, at a window boundary AFTER the wait for everything in flight: the chunk flushed one macro-step ago is in memory
    auto publish = [&]() {
        if (pub_ptr) {
            if (lam == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(pub_ptr), "v"(pub_cnt) : "memory");
            pub_ptr = nullptr;
        }
    };
    // split mode, before the window (x_pi, x_lam0) is fetched: the band above must have flushed its chunk -- and MB_LEAD more, so that
    // the counter is read once per `lead` windows.  (The wait below also drains this wave's own DMA queue: call it first.)
    int seen_pos = -1;
    unsigned seen = 0;
    auto wait_producer = [&]() {
        ensure(x_pi);
        const unsigned sp = stream_pair(x_pi);
        const int xb = stream_band(x_pi);
        if (sp == NOPAIR || xb == 0) return;
        const unsigned total = (unsigned)(NUp >> 3);
        const unsigned need = (unsigned)(x_lam0 >> 3) + 1u;
        if (seen_pos != x_pi) { seen_pos = x_pi; seen = 0; }
        if (seen >= need) return;
        const unsigned want = need + (unsigned)prm.lead < total ? need + (unsigned)prm.lead : total;
        const unsigned *pp = prm.prog + ((int64_t)(xb - 1) * prm.Pn + sp);
        // (bounded: the ticket order rules out a deadlock, but a counter that never moves -- a corrupted workspace, a fault in another
        // wave -- must cost a wrong result, which the NaN below makes visible, not a hung device: ~2^22 polls of >= 0.5 us)
        for (unsigned spins = 0;; ++spins) {
            unsigned v;
            asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(pp) : "memory");
            seen = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            if (seen >= want) break;
            if (spins >= (1u << 22)) {
                bad_lo = x_pi < bad_lo ? x_pi : bad_lo;
                bad_hi = x_pi > bad_hi ? x_pi : bad_hi;
                if (lam == 0 && prm.status) atomicAdd(prm.status, 1ULL);
                seen = total;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    };

    double xr[RC][FD], xsn[RC];   // Y32: xsn = -|x_row|^2 / sigma
#pragma unroll
    for (int k = 0; k < RC; ++k) xsn[k] = 0.0;
#pragma unroll
    for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int j = 0; j < FD; ++j) xr[k][j] = 0.0;
    // RBF: node values of this lane's rows at the columns of units uk (0..1) and uk+1 (2..3); the row above at the same columns
    double own[RBF ? RC : 1][4], abv[4];
#pragma unroll
    for (int k = 0; k < (RBF ? RC : 1); ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) own[k][c] = 1.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) abv[c] = 1.0;
    double left[R], bot[S], corner = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) left[i] = 1.0;
#pragma unroll
    for (int i = 0; i < S; ++i) bot[i] = 1.0;
    ExpCoefT<(Y32 ? 9 : 11)> expc;   // the polynomial's coefficients in VGPRs (the fp32 ring's results are fp32: degree 9): as SGPR pairs they spill the scalar state (v_readlane in the loop)
    if (RBF) expc.init();
    double *e_base = nullptr;   // EDGES: edge block of the pair the lane's sweep is in

    // ones for the windows whose sweep is in band 0 (see issue_x); visible to the LDS-DMA after the vmcnt(0) below
    if (lam < CHUNK / 16) store_through(wsrow + (int64_t)NUp * E + lam * 2, d2_t{1.0, 1.0});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (split) wait_producer();
    issue_y();
    issue_x();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (split) wait_producer();
    issue_y();
    issue_x();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (split) tband = stream_band(ps);   // (lane 0 starts in position 0; the others pick theirs up when their unit wraps)

    for (int t = 0; t < t_end; ++t) {
        // -- lane 0: the boundary entry of its unit u (K row for the sweep of uk, node pair at the columns of unit u), from
        //    the chunk the window's LDS-DMA brought in; no wait here (see lds_read_pend)
        d2_t pend[NP], bnd[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) async_begin2(pend[i]);
        // (lane 0's unit is t modulo NUp: a uniform address, read by every lane as a broadcast -- no divergent region around an
        // asynchronous read)
        // (the variants beyond 256 VGPRs -- RBF, dyadic 0, 16 dims: one wave per SIMD, registers spill to AGPRs -- read the entry with
        // blocking reads: the allocator may copy the destination of a read left in flight, lds_read_block in sk_wave_common.h)
        constexpr bool PEND = !(RBF && DY == 0 && FD == 16);
        if constexpr (PEND) {
            lds_read_pend<NP>(pend, lds0 + BI_BASE + (unsigned)(((t >> 3) & 1) * CHUNK + (t & 7) * (E * 8)));
        } else {
            double ent[E];
            lds_read_block<E>(ent, lds0 + BI_BASE + (unsigned)(((t >> 3) & 1) * CHUNK + (t & 7) * (E * 8)));
#pragma unroll
            for (int i = 0; i < NP; ++i) bnd[i] = d2_t{ent[2 * i], ent[2 * i + 1]};
        }

        // -- start of a row unit for the sweep: left boundary K[i][0] = 1
        if (uk == 0) {
            asm volatile("");   // a real branch (if-converted: ten v_cndmask in every macro-step instead of five moves for one lane)
            corner = 1.0;
#pragma unroll
            for (int i = 0; i < R; ++i) left[i] = 1.0;
            if constexpr (EDGES) {
                if (bandk == 0) {   // the sweep enters a pair: its edge block (1 / nb of the band starts)
                    asm volatile("");
                    const unsigned pair_e = stream_pair(psk);
                    e_base = pair_e != NOPAIR ? prm.edges + (int64_t)pair_e * ((int64_t)nb * NUp * S + (int64_t)nb * L * R) : nullptr;
                }
            }
        }
        // -- start of a row unit for the path reads: this lane's x rows (differences / points)
        if (u == 0) {
            const unsigned xa = my_x + (unsigned)(((t >> 3) & 1) * XSLAB);
#pragma unroll
            for (int k = 0; k < RC; ++k) lds_read_xrow<FD>(xr[k], xa + k * XROW);
            if constexpr (Y32) {
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    double q2 = 0.0;
#pragma unroll
                    for (int j = 0; j < FD; ++j) q2 = fma(xr[k][j], xr[k][j], q2);
                    xsn[k] = -q2 * prm.inv_sigma;
                }
            }
        }

        // -- y differences / points of the two columns of unit u, all FD dims (its lgkmcnt(0) also covers lane 0's entry)
        d2_t yv[FD];
        d2_t ysq_p[1], ysq_t[1];
        async_begin2(ysq_p[0]);
        {
            const unsigned ya = lds0 + (unsigned)(yslab * YSLAB + ((u & 7) << 4));
            if constexpr (Y32) {
                d2_t raw[FDY];
                if constexpr (PEND) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=&v"(ysq_p[0]) : "v"(ya) : "memory");   // |y|^2 of the two columns
                lds_read_dims<FDY>(raw, ya, ya + 128u);    // no swizzle (see the header)
#pragma unroll
                for (int jp = 0; jp < FDY; ++jp) {
                    const f4_t f = __builtin_bit_cast(f4_t, raw[jp]);
                    yv[2 * jp] = d2_t{(double)f[0], (double)f[1]};
                    yv[2 * jp + 1] = d2_t{(double)f[2], (double)f[3]};
                }
            } else {
                lds_read_dims<FD>(yv, ya + (unsigned)(ypar << 7), ya + (unsigned)((ypar ^ 1) << 7));
            }
        }
        if constexpr (PEND) lds_pend_take<NP>(bnd, pend);
        if constexpr (PEND || !Y32) {
            lds_pend_take<1>(ysq_t, ysq_p);
        } else {   // blocking (see PEND)
            double t2[2];
            lds_read_row1<2>(t2, lds0 + (unsigned)(yslab * YSLAB + ((u & 7) << 4)) + 1024u);
            ysq_t[0] = d2_t{t2[0], t2[1]};
        }
        double ysn[CW];   // Y32: -|y_q|^2 / sigma
#pragma unroll
        for (int q = 0; q < CW; ++q) ysn[q] = Y32 ? -ysq_t[0][q] * prm.inv_sigma : 0.0;

        // -- top row of the block: from the lane above, or (lane 0) the band boundary / the pair's boundary K[0][.] = 1
        //    (wave_shr leaves lane 0's destination = the `old` operand untouched: the boundary entry, ones in band 0)
        double top[S];
#pragma unroll
        for (int i = 0; i < S; ++i) top[i] = dpp_shr1(bot[i], bnd[i >> 1][i & 1]);

        // -- increments per coarse cell
        double ginc[RC][CW];
        if constexpr (RBF) {
            // the row above at the columns of unit u = uk + 1: the lane above evaluated them one macro-step ago
            abv[2] = dpp_shr1(own[RC - 1][0], bnd[NP - 1][0]);   // lane 0 keeps the boundary entry's node pair
            abv[3] = dpp_shr1(own[RC - 1][1], bnd[NP - 1][1]);
            if (is_top && (split ? tband : band) == 0) {   // node row 0 of the pair: nobody above has it
                double x0[FD];
                lds_read_xrow<FD>(x0, lds0 + T_BASE + (unsigned)((ps & 1) * XROW));
                if constexpr (Y32) {
                    double q2 = 0.0;
#pragma unroll
                    for (int j = 0; j < FD; ++j) q2 = fma(x0[j], x0[j], q2);
                    const double x0n = -q2 * prm.inv_sigma;
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        double xy = 0.0;
#pragma unroll
                        for (int j = 0; j < FD; ++j) xy = fma(x0[j], yv[j][q], xy);
                        abv[2 + q] = exp_nonpos(fma(xy, two_inv_sigma, x0n + ysn[q]), expc);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        double d2 = 0.0;
#pragma unroll
                        for (int j = 0; j < FD; ++j) {
                            const double df = x0[j] - yv[j][q];
                            d2 = fma(df, df, d2);
                        }
                        abv[2 + q] = exp_nonpos(fma(-d2, prm.inv_sigma, d2 * 0.0), expc);
                    }
                }
            }
            // this lane's node rows at the two columns of unit u = uk + 1
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) {
                    if constexpr (Y32) {
                        // -(|x|^2 + |y|^2 - 2<x,y>) / sigma, the reference's form (an infinite coordinate gives inf - inf = NaN as there)
                        double xy = 0.0;
#pragma unroll
                        for (int j = 0; j < FD; ++j) xy = fma(xr[k][j], yv[j][q], xy);
                        own[k][2 + q] = exp_nonpos(fma(xy, two_inv_sigma, xsn[k] + ysn[q]), expc);
                    } else {
                        double d2 = 0.0;
#pragma unroll
                        for (int j = 0; j < FD; ++j) {
                            const double df = xr[k][j] - yv[j][q];
                            d2 = fma(df, df, d2);
                        }
                        // d2 * 0 is NaN for an infinite / NaN distance, as the reference's |x|^2 + |y|^2 - 2<x,y> is (sk_wave_fused.hip)
                        own[k][2 + q] = exp_nonpos(fma(-d2, prm.inv_sigma, d2 * 0.0), expc);
                    }
                }
            // 4-corner differences in the reference's order (sigkernel.py:362-363): ((G11 + G00) - G10) - G01
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) {
                    const double t0 = k == 0 ? abv[q] : own[(k + RC - 1) % RC][q];
                    const double t1 = k == 0 ? abv[q + 1] : own[(k + RC - 1) % RC][q + 1];
                    ginc[k][q] = ((own[k][q + 1] + t0) - own[k][q]) - t1;
                }
        } else {
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q) {
                    double g = 0.0;
#pragma unroll
                    for (int j = 0; j < FD; ++j) g = fma(xr[k][j], yv[j][q], g);
                    ginc[k][q] = g;
                }
        }
        if constexpr (EDGES) {   // no increments in the padding: the terminal edges of the padded grid are then the clamped ones
#pragma unroll
            for (int k = 0; k < RC; ++k)
#pragma unroll
                for (int q = 0; q < CW; ++q)
                    if (!((bandk * L + lam) * RC + k < prm.Mc && 2 * uk + q < prm.Nc)) ginc[k][q] = 0.0;
        }
        double ca[RC][CW], cbm[RC][CW];
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const double g = ginc[k][q];
                if (NAIVE) {
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
            }
            bot[cc] = above;
        }
        corner = top[S - 1];

        // -- lane 63: this step's boundary entry (position = its unit u) into the outgoing chunk
        if (is_bot) {
            const unsigned ea = lds0 + BO_BASE + (unsigned)((u & 7) * (E * 8));
#pragma unroll
            for (int cc = 0; cc < S; cc += 2) lds_write_b128(ea + cc * 8u, d2_t{bot[cc], bot[cc + 1]});
            if (RBF) lds_write_b128(ea + S * 8u, d2_t{own[RC - 1][2], own[RC - 1][3]});
        }

        if constexpr (EDGES) {
            // the bottom fine row of every band (row nb-1-band of the pair's block: the LAST band's is the terminal row), from the
            // bottom lane; the terminal column: every lane's rows after the last unit
            // (e_base: the edge block of the pair this lane's sweep is in, looked up once per pair -- see the top of the step)
            if (is_bot && e_base) {
                double *er = e_base + ((nb - 1 - bandk) * NUp + uk) * S;
#pragma unroll
                for (int cc = 0; cc < S; cc += 2) *reinterpret_cast<d2_t *>(er + cc) = d2_t{bot[cc], bot[cc + 1]};
            }
            if (uk == NUp - 1 && e_base) {
                double *ec = e_base + (nb * NUp * S + (bandk * L + lam) * R);
#pragma unroll
                for (int rr = 0; rr < R; rr += 2) *reinterpret_cast<d2_t *>(ec + rr) = d2_t{left[rr], left[rr + 1]};
            }
        }

        // -- K[MM][NN] of a pair
        if (uk == my_uf) {
            int pv = psk, bv = bandk;
            asm volatile("" : "+v"(pv), "+v"(bv));
            if (split) bv = stream_band(pv);
            const unsigned pair_v = bv == prm.band_f ? stream_pair(pv) : NOPAIR;
            if (pair_v != NOPAIR) {
                double v = cand[0][0];
#pragma unroll
                for (int k = 0; k < RC; ++k)
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        double cv = cand[k][q];
                        asm volatile("" : "+v"(cv));
                        if (k * CW + q == prm.sel_f) v = cv;
                    }
                if (SPLIT && pv >= bad_lo && pv <= bad_hi) v = __longlong_as_double(0x7ff8000000000000LL);
                static_cast<TO *>(prm.out)[pair_v] = (TO)v;
            }
        }

        // -- shift the node history
        if constexpr (RBF) {
#pragma unroll
            for (int k = 0; k < RC; ++k) { own[k][0] = own[k][2]; own[k][1] = own[k][3]; }
            abv[0] = abv[2];
            abv[1] = abv[3];
        }

        // -- advance the cursors
        if (RBF) {
            uk += 1;
            if (uk == NUp) {
                uk = 0;
                bandk += 1;
                if (bandk == nb) { bandk = 0; psk += 1; }
            }
        }
        u += 1;
        if (((t + 1) & 7) == lam7) {   // (u & 7) == 0
            yslab = yslab + 1 == NSLAB ? 0 : yslab + 1;
            ypar ^= 1;
            if (u == NUp) {
                u = 0;
                band += 1;
                if (band == nb) { band = 0; ps += 1; }
                if (split) tband = stream_band(ps);
            }
        }
        if (!RBF) { uk = u; bandk = band; psk = ps; }

        // -- window bookkeeping.  Lane 63's unit is t - 63: its chunk is complete when (t + 1) & 7 == 7 and goes out then;
        //    one macro-step later everything in flight is waited for and the next window's LDS-DMA is issued
        if (((t + 1) & 7) == 7 && t >= L - 1 + 7) flush_chunk();
        if (((t + 1) & 7) == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (split) {
                publish();
                wait_producer();
            }
            issue_y();
            issue_x();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (split) {
        publish();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <typename TO, int DY, bool Y32, int KIND, int FD, bool EDGES = false, int RCX = Tile<DY>::RC, bool SPLIT = false>
int launch_mb_one(FusedMbParams prm, int64_t P, size_t lds_bytes, int waves_per_cu, double *ws, size_t ws_bytes, hipStream_t s) {
    if constexpr (!SPLIT && !EDGES) {
        if (prm.split) return launch_mb_one<TO, DY, Y32, KIND, FD, false, RCX, true>(prm, P, lds_bytes, waves_per_cu, ws, ws_bytes, s);
    }
    auto kern = k_fwd_fused_mb<TO, DY, Y32, KIND, FD, EDGES, RCX, SPLIT>;
    // (a property of this variant's code object, the same on every gfx950 device: an immutable constant initialised once,
    // thread-safely, at the variant's first launch -- not mutable library state)
    static const int vgprs = [&] {
        hipFuncAttributes attr;
        return hipFuncGetAttributes(&attr, (const void *)kern) == hipSuccess && attr.numRegs > 0 ? attr.numRegs : 256;
    }();
    const int by_regs = 4 * (512 / ((vgprs + 7) & ~7));
    if (waves_per_cu > by_regs) waves_per_cu = by_regs;
    if (waves_per_cu < 1) waves_per_cu = 1;
    const int64_t max_waves = (int64_t)device_cu_count() * waves_per_cu;
    int64_t waves = P < max_waves ? P : max_waves;
    if (P >= 0x7ff00000LL) return SK_ERR_UNSUPPORTED;            // (pair indices are 32-bit inside the kernel)
    int64_t per = (P + waves - 1) / waves;                       // the equal share, pairs per wave
    if (per > 0x1fffffff / ((int64_t)prm.nb * prm.NUp)) return SK_ERR_UNSUPPORTED;
    const int pct = knobs().fusedmb_q_static > 0 ? (knobs().fusedmb_q_static > 100 ? 100 : knobs().fusedmb_q_static) : 50;
    if (prm.split) {
        // every position is a ticket: [per-wave rows (only their chunk of ones is used)][item rows][progress counters][the counter]
        if constexpr (!SPLIT) return SK_ERR_UNSUPPORTED;
        const size_t wave_doubles = (size_t)waves * (size_t)prm.ws_stride, row_doubles = (size_t)P * (size_t)prm.row_stride;
        const size_t prog_bytes = ((size_t)P * sizeof(unsigned) + 63) / 64 * 64;
        if (!ws || ws_bytes < (wave_doubles + row_doubles) * sizeof(double) + prog_bytes + 64) return SK_ERR_WORKSPACE;
        prm.rows = ws + wave_doubles;
        prm.prog = reinterpret_cast<unsigned *>(ws + wave_doubles + row_doubles);
        prm.queue = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(prm.prog) + prog_bytes);
        prm.C0 = 0;
        prm.q_first = 0;
        if (hipMemsetAsync(prm.prog, 0, prog_bytes + sizeof(unsigned long long), s) != hipSuccess) return SK_ERR_LAUNCH;
        // the status word: the last 8 bytes of the workspace AS PASSED (inside the 64 bytes of slack behind the counter whatever the
        // resident waves turn out to be, so that the caller finds it without knowing the layout)
        prm.status = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ws) + (ws_bytes & ~(size_t)7)) - 1;
        if (hipMemsetAsync(prm.status, 0, sizeof(unsigned long long), s) != hipSuccess) return SK_ERR_LAUNCH;
    } else if (!ws || ws_bytes < (size_t)waves * (size_t)prm.ws_stride * sizeof(double) + 64) {
        return SK_ERR_WORKSPACE;
    } else if (waves == max_waves && per >= 8 && pct < 100) {
        // the launch fills the chip: `pct` per cent of the equal share is dealt out up front, the rest is drawn pair by pair
        prm.C0 = (int)(per * pct / 100);
        prm.queue = reinterpret_cast<unsigned long long *>(ws + (size_t)waves * (size_t)prm.ws_stride);
        prm.q_first = waves * (int64_t)prm.C0;
        if (hipMemsetAsync(prm.queue, 0, sizeof(unsigned long long), s) != hipSuccess) return SK_ERR_LAUNCH;
    } else {
        waves = (P + per - 1) / per;
        prm.C0 = (int)per;
        prm.queue = nullptr;
        prm.q_first = P;
    }
    prm.wg = wave_group(lds_bytes, waves, knobs().fusedmb_wpb);
    prm.PPW = (int)per;
    prm.n_steps = 0;
    prm.ws = ws;
    const size_t lds_block = wave_group_lds(prm.wg);
    if (lds_block > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
    SK_LAUNCH(kern, dim3(wave_group_blocks(prm.wg)), dim3(WAVE * prm.wg.wpb), lds_block, s, prm);
    return check_launch();
}

struct MbPlan {
    int RC, S, NUp, nb, fd, waves_per_cu;
    size_t lds_bytes;
    int64_t ws_stride;   // doubles per wave
    bool ok;
};

MbPlan mb_plan(int kind, int Mc, int Nc, int dyadic, int D, bool y32 = false, bool edges = false, bool split = false) {
    MbPlan pl{};
    pl.ok = false;
    if (dyadic < 0 || dyadic > 2 || D < 1 || D > 16 || (kind != 0 && kind != 1)) return pl;
    pl.RC = dyadic == 0 ? (kind == 1 && edges ? 2 : 4) : dyadic == 1 ? 2 : 1;   // (rbf edges at dyadic 0: the adjoint's two rows per lane)
    pl.S = 2 << dyadic;
    pl.fd = D <= 8 ? 8 : 16;
    const int NU = kind == 1 ? (Nc + 2) / 2 : (Nc + 1) / 2;
    pl.NUp = (NU + LINE_UNITS - 1) / LINE_UNITS * LINE_UNITS;
    // band boundary slack, and at most one row unit starting per window (see the header): a shorter second path is swept with
    // padding units behind it -- causality keeps K[MM][NN] what it is, the EDGES variant masks the padding's increments
    if (pl.NUp < MB_L + 16) pl.NUp = MB_L + 16;
    // split mode: the RBF sweep trails the node evaluation by one unit, and the sweep of a pair's LAST real unit must end inside its
    // own stream position (the next position is another pair's band): one padding unit behind it
    if (split && kind == 1 && pl.NUp < (Nc - 1) / 2 + 2) pl.NUp += LINE_UNITS;
    pl.nb = (Mc + (edges && kind == 1 ? 1 : 0) + MB_L * pl.RC - 1) / (MB_L * pl.RC);   // edges: the bands of sk_wave_adj_fused_mb.hip (rbf: node rows)
    const size_t xslab = (size_t)8 * pl.RC * pl.fd * 8;
    const size_t chunk = (size_t)8 * (pl.S + (kind == 1 ? 2 : 0)) * 8;
    pl.lds_bytes = (size_t)(MB_L / 8 + 2) * ((y32 ? pl.fd / 2 + 1 : pl.fd) * 128) + MB_X_SLOTS * xslab + 3 * chunk + (kind == 1 ? 2 * pl.fd * 8 : 0);
    pl.ws_stride = (int64_t)(pl.NUp + 8) * (pl.S + (kind == 1 ? 2 : 0));   // the row + a chunk of ones
    int wpc = (int)((160 * 1024) / pl.lds_bytes);
    const int wpc_env = knobs().fusedmb_wpc;
    if (wpc > 12) wpc = 12;
    if (wpc_env > 0) wpc = wpc < wpc_env ? wpc : wpc_env;
    else if (wpc > 4) wpc &= ~3;
    if (wpc < 1) wpc = 1;
    pl.waves_per_cu = wpc;
    pl.ok = true;
    return pl;
}

template <typename TO, int DY, int KIND>
int launch_mb_dy(const FusedMbParams &prm, const MbPlan &pl, bool naive, bool y32, int64_t P, double *ws, size_t ws_bytes,
                 hipStream_t s) {
    (void)naive;   // (a launch-time constant of the kernel: FusedMbParams::naive)
    if (prm.edges) {   // with the edges: what sk_rbf_adjoint_fused_mb_f64 (dyadic 1..2) / sk_linear_adjoint_fused_mb_f64 sweep
        if constexpr (KIND == 0) {
            if (y32) return SK_ERR_UNSUPPORTED;
            if (pl.fd == 8) return launch_mb_one<TO, DY, false, KIND, 8, true>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
            return launch_mb_one<TO, DY, false, KIND, 16, true>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
        }
        if constexpr (KIND == 1 && DY == 0) {
            if (y32) return SK_ERR_UNSUPPORTED;
            if (pl.fd == 8) return launch_mb_one<TO, 0, false, KIND, 8, true, 2>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
            return launch_mb_one<TO, 0, false, KIND, 16, true, 2>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
        }
        if constexpr (KIND == 1 && DY >= 1) {
            if (y32) {
                if constexpr (sizeof(TO) == 4) {
                    if (pl.fd == 16) return launch_mb_one<TO, DY, true, KIND, 16, true>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
                }
                return SK_ERR_UNSUPPORTED;
            }
            if (pl.fd == 8) return launch_mb_one<TO, DY, false, KIND, 8, true>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
            // (fp32 RBF paths of 9..16 dims at dyadic >= 1 are staged as packed fp32 -- y32 -- by every caller: the fp64-ring form of that
            // case was an A/B knob's instance, removed in round 6)
            if constexpr (sizeof(TO) == 4) return SK_ERR_UNSUPPORTED;
            else return launch_mb_one<TO, DY, false, KIND, 16, true>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
        }
        return SK_ERR_UNSUPPORTED;
    }
    if (y32) {   // fp32 y ring: built where it pays (RBF points of fp32 inputs, 16 dims, dyadic >= 1: at dyadic 0 the four-row form has no registers for it)
        if constexpr (KIND == 1 && sizeof(TO) == 4 && DY >= 1) {
            if (pl.fd == 16) return launch_mb_one<TO, DY, true, KIND, 16>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
        }
        return SK_ERR_UNSUPPORTED;
    }
    if (pl.fd == 8) return launch_mb_one<TO, DY, false, KIND, 8>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
    if constexpr (KIND == 1 && sizeof(TO) == 4 && DY >= 1) return SK_ERR_UNSUPPORTED;     // (see above: y32 serves this case)
    else return launch_mb_one<TO, DY, false, KIND, 16>(prm, P, pl.lds_bytes, pl.waves_per_cu, ws, ws_bytes, s);
}

}  // namespace

// SPLIT mode (see FusedMbParams): worth it when the pairs alone leave at least half of the resident waves idle and a band is long
// enough for the trailing to be cheap (and for the deadlock argument: a band's first MB_LEAD + 16 chunks never reach into the
// last eight of the band above); SK_FUSEDMB_SPLIT=0 switches it off.  Bytes of the item rows, or 0.
constexpr size_t MB_SPLIT_MAX_BYTES = (size_t)2 << 30;
size_t mb_split_rows_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int D, bool y32, bool edges) {
    if (edges || knobs().fusedmb_split == 0) return 0;
    const MbPlan pl = mb_plan(kind, Mc, Nc, dyadic, D, y32, false, true);
    if (!pl.ok || pl.nb < 2 || pl.NUp < 8 * (MB_LEAD + 24) || pl.NUp < (int)cost_by_name("mb_split_min_units")) return 0;
    const int64_t resident = (int64_t)device_cu_count() * pl.waves_per_cu;
    if ((double)P > cost_by_name("mb_split_max_resident_share") * (double)resident || P * (int64_t)pl.nb >= 0x7ff00000LL) return 0;
    const size_t bytes = (size_t)P * pl.nb * (size_t)pl.NUp * (pl.S + (kind == 1 ? 2 : 0)) * sizeof(double);
    return bytes <= MB_SPLIT_MAX_BYTES ? bytes : 0;
}

// Workspace (bytes) of sk_solve_fwd_static_*: one boundary row per resident wave (+ split mode: one per band of every pair and the
// progress counters); 0 outside the kernel's scope.
size_t fused_mb_workspace_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int D) {
    const MbPlan pl = mb_plan(kind, Mc, Nc, dyadic, D);
    if (!pl.ok || P <= 0) return 0;
    const int64_t max_waves = (int64_t)device_cu_count() * 16;   // an upper bound on the resident waves whatever the variant's register count
    const int64_t waves = P < max_waves ? P : max_waves;
    size_t bytes = (size_t)waves * (size_t)pl.ws_stride * sizeof(double) + 64;   // + the launch's work counter
    const size_t rows = mb_split_rows_bytes(kind, P, Mc, Nc, dyadic, D, false, false);
    if (rows) {
        const MbPlan ps = mb_plan(kind, Mc, Nc, dyadic, D, false, false, true);
        const int64_t items = P * ps.nb, w2 = items < max_waves ? items : max_waves;
        const size_t split_bytes = (size_t)w2 * (size_t)ps.ws_stride * sizeof(double) + rows + ((size_t)items * sizeof(unsigned) + 63) / 64 * 64 + 64;
        if (split_bytes > bytes) bytes = split_bytes;
    }
    return bytes;
}
// bands per pair when a launch of P pairs (fp64 staging, no edges) runs the bands of a pair on several waves; 0: one wave per pair
int fused_mb_split(int kind, int64_t P, int Mc, int Nc, int dyadic, int D) {
    if (!mb_split_rows_bytes(kind, P, Mc, Nc, dyadic, D, false, false)) return 0;
    return mb_plan(kind, Mc, Nc, dyadic, D, false, false, true).nb;
}
// columns (Ncp = 2 NUp) the caller must provide per dimension row of Yt: the units of a band incl. the padding of short second paths
int fused_mb_cols(int kind, int Nc) {
    const int NU = kind == 1 ? (Nc + 2) / 2 : (Nc + 1) / 2;
    int NUp = (NU + LINE_UNITS - 1) / LINE_UNITS * LINE_UNITS;
    if (NUp < MB_L + 16) NUp = MB_L + 16;
    if (kind == 1 && NUp < (Nc - 1) / 2 + 2) NUp += LINE_UNITS;   // (the padding unit split mode sweeps behind a pair's last real one)
    return 2 * NUp;
}
// rows the caller must provide per path in Xr (node / difference rows incl. the padding the last band reads)
int fused_mb_rows(int kind, int Mc, int dyadic, bool edges) {
    const int RC = dyadic == 0 ? (kind == 1 && edges ? 2 : 4) : dyadic == 1 ? 2 : 1;
    const int nb = (Mc + (edges && kind == 1 ? 1 : 0) + MB_L * RC - 1) / (MB_L * RC);
    return nb * MB_L * RC + 8;
}

template <typename TO>
int launch_fwd_fused_mb(int kind, const double *Xr, const void *Yt_any, int yt_f32, int64_t A, int64_t B, int Mrows, int Ncp, int D,
                        int fd, const Geom &g, double inv_sigma, TO *out, double *edges, void *ws, size_t ws_bytes, hipStream_t s) {
    // yt_f32: Yt holds fp32 values packed two dimensions per 16-byte unit (sk_prep_paths_* layout 2): byte for byte a
    // dimension-major fp64 array of fd / 2 rows, which is how the kernel addresses it
    const double *Yt = static_cast<const double *>(Yt_any);
    const bool y32 = yt_f32 != 0;
    MbPlan pl = mb_plan(kind, g.Mc, g.Nc, g.dyadic, D, y32, edges != nullptr);
    if (!pl.ok || fd != pl.fd) return SK_ERR_UNSUPPORTED;
    if (Ncp < pl.NUp * 2 || (Ncp & 1) || Mrows < fused_mb_rows(kind, g.Mc, g.dyadic, edges != nullptr)) return SK_ERR_UNSUPPORTED;
    FusedMbParams prm{};
    prm.Xr = Xr; prm.Yt = Yt; prm.out = out; prm.edges = edges; prm.P = g.P; prm.B = B; prm.Mrows = Mrows; prm.Ncp = Ncp;
    prm.Mc = g.Mc; prm.Nc = g.Nc; prm.NUp = pl.NUp; prm.nb = pl.nb; prm.inv_sigma = inv_sigma; prm.ws_stride = pl.ws_stride;
    prm.naive = g.naive;
    prm.split = 0; prm.Pn = g.P; prm.nb_true = pl.nb; prm.rows = nullptr; prm.row_stride = 0; prm.prog = nullptr; prm.status = nullptr;
    int64_t P_launch = g.P;
    const size_t split_rows = mb_split_rows_bytes(kind, g.P, g.Mc, g.Nc, g.dyadic, D, y32, edges != nullptr);
    if (split_rows) {
        const MbPlan ps = mb_plan(kind, g.Mc, g.Nc, g.dyadic, D, y32, false, true);
        const int64_t items = g.P * ps.nb, max_waves = (int64_t)device_cu_count() * 16, w2 = items < max_waves ? items : max_waves;
        const size_t need = (size_t)w2 * (size_t)ps.ws_stride * sizeof(double) + split_rows + ((size_t)items * sizeof(unsigned) + 63) / 64 * 64 + 64;
        if (Ncp >= ps.NUp * 2 && ws_bytes >= need) {   // (a caller that sized Yt / the workspace for the one-wave sweep gets that sweep)
            pl = ps;
            prm.split = 1;
            prm.lead = knobs().fusedmb_lead > 0 ? (knobs().fusedmb_lead > MB_LEAD ? MB_LEAD : knobs().fusedmb_lead) : 4;
            prm.NUp = ps.NUp;
            prm.nb = 1;                      // every stream position is one row unit: an item (band, pair)
            prm.ws_stride = ps.ws_stride;
            prm.row_stride = (int64_t)ps.NUp * (ps.S + (kind == 1 ? 2 : 0));
            prm.P = items;
            P_launch = items;
        }
    }
    const int row_unit = (g.Mc - 1) / pl.RC;      // lane-row that holds the last coarse row
    prm.u_f = (g.Nc - 1) / 2;
    prm.lam_f = row_unit % MB_L;
    prm.band_f = row_unit / MB_L;
    prm.sel_f = ((g.Mc - 1) % pl.RC) * 2 + (g.Nc - 1) % 2;
    (void)A;
    double *w = static_cast<double *>(ws);
    if (kind == 0) {
        switch (g.dyadic) {
            case 0: return launch_mb_dy<TO, 0, 0>(prm, pl, g.naive, y32, P_launch, w, ws_bytes, s);
            case 1: return launch_mb_dy<TO, 1, 0>(prm, pl, g.naive, y32, P_launch, w, ws_bytes, s);
            default: return launch_mb_dy<TO, 2, 0>(prm, pl, g.naive, y32, P_launch, w, ws_bytes, s);
        }
    }
    switch (g.dyadic) {
        case 0: return launch_mb_dy<TO, 0, 1>(prm, pl, g.naive, y32, P_launch, w, ws_bytes, s);
        case 1: return launch_mb_dy<TO, 1, 1>(prm, pl, g.naive, y32, P_launch, w, ws_bytes, s);
        default: return launch_mb_dy<TO, 2, 1>(prm, pl, g.naive, y32, P_launch, w, ws_bytes, s);
    }
}

template int launch_fwd_fused_mb<double>(int, const double *, const void *, int, int64_t, int64_t, int, int, int, int, const Geom &, double,
                                         double *, double *, void *, size_t, hipStream_t);
template int launch_fwd_fused_mb<float>(int, const double *, const void *, int, int64_t, int64_t, int, int, int, int, const Geom &, double,
                                        float *, double *, void *, size_t, hipStream_t);

}  // namespace sk
