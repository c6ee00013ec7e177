// sk_adj_fused_rescue.hip -- device-side rescue of the FUSED adjoints (sk_wave_adj_fused.hip, sk_wave_adj_fused_rbf.hip).
//
// The fused adjoints recompute the forward solution backwards from its terminal edges; that recurrence loses accuracy like
// 1e-16 K_max^2 and is useless for exploding kernels (|K| > ~1e4, where the explicit scheme itself means little -- but
// unnormalised paths in a training loop get there).  The unfused route re-solves such pairs with stored grids
// (sk_adj_rescue_*); a fused adjoint cannot simply be patched, because a pair's contribution is summed, in registers, with
// those of the other pairs of its lane group.  So:
//   * k_screen marks, BEFORE the sweep, every pair whose forward value |K[MM][NN]| exceeds `screen` (the caller has the
//     forward values; the backward recurrence is safe below ~1e3): its upstream gradient becomes NaN -- the fused kernels skip
//     such a pair entirely -- and its residual entry -1;
//   * after the sweep k_fused_rescue walks the lane groups' chunks (the same ChunkSplit the sweep used).  A chunk with marked
//     pairs only gets their EXACT contributions added to its partial sums (stored-grid adjoint of sk_pair_sweep.h +
//     static-kernel chain rule, one wavefront per chunk, pairs in ascending order: reproducible); a chunk in which a pair that
//     was not marked failed its self-check after the fact (residual > tol: rare, needs a kernel that peaks inside the grid)
//     is recomputed from scratch, all pairs exactly, and its partial sums replaced.
// Nothing comes back to the host: a backward pass has no synchronisation, and when no pair is marked or failed (the normal
// case) the rescue reads P residuals and exits.  Reference behaviour covered: sigkernel.py:419-502 stores both grids for
// every pair, whatever K's size.
#include "sk_pair_sweep.h"
#include "sk_wave_common.h"

namespace sk {
namespace {

__global__ void k_screen(const double *__restrict__ kfinal, const double *__restrict__ scale, int64_t P, double screen,
                         double *__restrict__ scale_eff, double *__restrict__ err) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const bool wild = fabs(kfinal[p]) > screen;      // (a NaN value -- poisoned inputs -- is left to the sweep: NaN either way)
    scale_eff[p] = wild ? __longlong_as_double(0x7ff8000000000000LL) : (scale ? scale[p] : 1.0);
    err[p] = wild ? -1.0 : 0.0;
}

struct FusedRescueParams {
    int kind;                  // 0: linear (Xs = s^2 dx [A][Mrows][8], Ys = dy [B][8][Ncp]); 1: rbf (the same layouts hold points)
    const double *Xs, *Ys;
    const double *scale;       // the ORIGINAL upstream gradient [P], nullable (= 1)
    const double *err;         // residuals after the sweep: -1 marked by k_screen, > tol failed
    double tol;
    double *part;              // Tpart [groups][rows][8] (linear, coarse rows flipped) / Gpart [groups][rows][outw] (rbf, node rows)
    double *Ypart;             // nullable: rbf [P][ycols][outw] per node column (outw = 6 / 10 for dims <= 4 / 5..8); linear [P][ycols][8] per increment column, and `part` null
                               // (the second-argument form of sk_wave_adj_fused.hip has no first-argument sums to patch)
    int64_t A, B, P, n_groups;
    int Mrows, Ncp, Mc, Nc, D, dyadic, rows, outw, ycols;
    int naive;                 // _naive_solver stencil (cython_backend.pyx:114)
    const double *kfinal;      // forward values [P], nullable: tells a NaN residual of poisoned inputs (K not finite either: left alone,
                               // the re-solve could only reproduce the NaN) from one of a recompute that overflowed on finite inputs
                               // (K finite: the chunk is recomputed exactly like any other failed self-check)
    int fd;                    // dims carried by Xs / Ys (8; 8 or 16 for sk_wave_adj_fused_mb.hip)
    double *N0;                // sk_wave_adj_fused_mb.hip, nullable: [P][n0cols] node row 0 weights of the sweep, cleared for a failed pair
    int n0cols;
    double inv_sigma;
    ChunkSplit cs;
    double *ws;                // per block: inc [Mc][Nc], W [Mc][Nc], G [M][N] (rbf), Kf, Kr
    int64_t ws_block;          // doubles per block
};

// exact contribution of pair p, added to the chunk's partial sums `slot` (null: none kept) and written to Ypart (if kept)
__device__ void rescue_pair(const FusedRescueParams &prm, int64_t p, double *slot, double *lds, double *wsb) {
    const int Mc = prm.Mc, Nc = prm.Nc, M = Mc + 1, N = Nc + 1, d = prm.dyadic;
    const int64_t a = prm.B > 0 ? p / prm.B : p, b = prm.B > 0 ? p % prm.B : p;
    const int fd = prm.fd;
    const double *xs = prm.Xs + a * (int64_t)prm.Mrows * fd;     // row r: xs[r * fd + k]
    const double *ys = prm.Ys + b * (int64_t)fd * prm.Ncp;       // column c: ys[k * Ncp + c]
    double *inc = wsb, *W = inc + (int64_t)Mc * Nc, *G = W + (int64_t)Mc * Nc;
    double *Kf = G + (prm.kind == 1 ? (int64_t)M * N : 0), *Kr = Kf + (int64_t)((Mc << d) + 1) * ((Nc << d) + 1);
    const double s = prm.scale ? prm.scale[p] : 1.0;
    const int lane = threadIdx.x;
    if (prm.kind == 0) {
        for (int c = lane; c < Mc * Nc; c += WAVE) {
            const int pp = c / Nc, q = c - pp * Nc;
            double g = 0.0;
            for (int k = 0; k < fd; ++k) g = fma(xs[pp * fd + k], ys[(int64_t)k * prm.Ncp + q], g);
            inc[c] = g;
        }
    } else {
        for (int c = lane; c < M * N; c += WAVE) {
            const int r = c / N, q = c - r * N;
            double d2 = 0.0;
            for (int k = 0; k < fd; ++k) {
                const double df = xs[r * fd + k] - ys[(int64_t)k * prm.Ncp + q];
                d2 = fma(df, df, d2);
            }
            G[c] = exp(-d2 * prm.inv_sigma);
        }
        __syncthreads();
        for (int c = lane; c < Mc * Nc; c += WAVE) {
            const int pp = c / Nc, q = c - pp * Nc;
            inc[c] = ((G[(pp + 1) * N + q + 1] + G[pp * N + q]) - G[(pp + 1) * N + q]) - G[pp * N + q + 1];   // sigkernel.py:362-363
        }
    }
    __syncthreads();
    adj_pair<double>(inc, Nc, Mc, Nc, d, prm.naive, lds, Kf, Kr, nullptr, W, Nc);
    if (prm.kind == 0) {
        // T[a][pp][k] += s sum_q W[pp][q] dy[q][k], kept at flipped row rows - 1 - pp (sk_wave_adj_fused.hip)
        for (int c = lane; slot && c < Mc * fd; c += WAVE) {
            const int pp = c / fd, k = c - pp * fd;
            double t = 0.0;
            for (int q = 0; q < Nc; ++q) t = fma(W[(int64_t)pp * Nc + q], ys[(int64_t)k * prm.Ncp + q], t);
            slot[(int64_t)(prm.rows - 1 - pp) * fd + k] += s * t;
        }
        // second argument: per increment column q, sum_pp W[pp][q] s^2 dx[pp][:], WITHOUT the upstream gradient
        for (int c = lane; prm.Ypart && c < Nc * 8; c += WAVE) {
            const int q = c >> 3, k = c & 7;
            double t = 0.0;
            for (int pp = 0; pp < Mc; ++pp) t = fma(W[(int64_t)pp * Nc + q], xs[pp * fd + k], t);
            prm.Ypart[(p * prm.ycols + q) * 8 + k] = t;
        }
    } else {
        // V[r][c] = w[r-1][c-1] + w[r][c] - w[r-1][c] - w[r][c-1] (w = W inside the grid, 0 outside): d k / d G[r][c]
        auto w_at = [&](int r, int c) -> double { return (r >= 0 && r < Mc && c >= 0 && c < Nc) ? W[(int64_t)r * Nc + c] : 0.0; };
        auto VG = [&](int r, int c) -> double {
            return (((w_at(r - 1, c - 1) + w_at(r, c)) - w_at(r - 1, c)) - w_at(r, c - 1)) * G[r * N + c];
        };
        for (int r = lane; slot && r < M; r += WAVE) {      // first argument: per node row, cs = sum_c V G, accd = sum_c V G y_c
            double cs = 0.0, acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int c = 0; c < N; ++c) {
                const double v = VG(r, c);
                cs += v;
                for (int k = 0; k < prm.outw - 2; ++k) acc[k] = fma(v, ys[(int64_t)k * prm.Ncp + c], acc[k]);
            }
            double *dst = slot + (int64_t)r * prm.outw;
            dst[0] += s * cs;
            for (int k = 0; k < prm.outw - 2; ++k) dst[2 + k] += s * acc[k];
        }
        if (prm.Ypart) {                             // second argument: per node column, WITHOUT the upstream gradient
            const int yd = prm.outw - 2;             // (the stride of the sums is the partial sums': 2 + 4 or 2 + 8 doubles)
            for (int c = lane; c < N; c += WAVE) {
                double s0 = 0.0, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int r = 0; r < M; ++r) {
                    const double v = VG(r, c);
                    s0 += v;
                    for (int k = 0; k < yd; ++k) s1[k] = fma(v, xs[r * fd + k], s1[k]);
                }
                double *dst = prm.Ypart + (p * prm.ycols + c) * prm.outw;
                dst[0] = s0; dst[1] = 0.0;
                for (int k = 0; k < yd; ++k) dst[2 + k] = s1[k];
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(WAVE) void k_fused_rescue(const FusedRescueParams prm) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *wsb = prm.ws + (int64_t)blockIdx.x * prm.ws_block;
    const int64_t A = prm.B > 0 ? prm.P / prm.B : prm.P;
    if (prm.cs.nr == 1 && prm.cs.size[0] == 1 && prm.n_groups == prm.P) {
        // one pair per chunk, slot = pair (sk_wave_adj_fused_mb.hip: tens of thousands of chunks): the lanes look at 64 residuals at a
        // time, the wave stops only for the flagged ones
        for (int64_t base = (int64_t)blockIdx.x * WAVE; base < prm.P; base += (int64_t)gridDim.x * WAVE) {
            const int64_t p = base + threadIdx.x;
            const double e = p < prm.P ? prm.err[p] : 0.0;
            unsigned long long todo = __ballot(e > prm.tol || e < 0.0);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int64_t pp = base + l;
                const bool failed = prm.err[pp] > prm.tol;
                double *slot = prm.part ? prm.part + pp * (int64_t)prm.rows * prm.outw : nullptr;
                if (failed) {
                    for (int c = threadIdx.x; slot && c < prm.rows * prm.outw; c += WAVE) slot[c] = 0.0;
                    if (prm.N0)
                        for (int c = threadIdx.x; c < prm.n0cols; c += WAVE) prm.N0[pp * prm.n0cols + c] = 0.0;
                    __syncthreads();
                }
                rescue_pair(prm, pp, slot, lds, wsb);
            }
        }
        return;
    }
    // The lanes look at 64 chunks at a time -- each its own chunk's residuals -- and the wave stops only for the flagged ones (one
    // chunk per iteration with the lanes across its pairs cost 1.1 us per chunk in address arithmetic and one dependent load:
    // 71 us for the 4096 chunks of 128 x 128 pairs, 8 % of a training step of that size).  A chunk is still handled by exactly
    // one wave, its pairs in ascending order: reproducible.
    auto bcast64 = [](int64_t v, int l) -> int64_t {
        const unsigned lo = (unsigned)__shfl((int)(unsigned)(v & 0xffffffffLL), l), hi = (unsigned)__shfl((int)(v >> 32), l);
        return (int64_t)(((unsigned long long)hi << 32) | lo);
    };
    const bool short_chunks = prm.cs.size[0] < 32;   // (long chunks: the lanes across the pairs of one chunk, coalesced, below)
    for (int64_t g0 = (int64_t)blockIdx.x * WAVE; short_chunks && g0 < prm.n_groups; g0 += (int64_t)gridDim.x * WAVE) {
        const int64_t gi = g0 + threadIdx.x;
        int64_t first = prm.P, slot_i = 0;
        int ppg = 0;
        bool failed = false, marked = false;
        if (gi < prm.n_groups) {
            chunk_share(prm.cs, gi, A, prm.B, prm.P, first, slot_i, ppg);
            for (int i = 0; i < ppg && first + i < prm.P; ++i) {
                const double e = prm.err[first + i];
                failed |= e > prm.tol || (e != e && prm.kfinal && isfinite(prm.kfinal[first + i]));   // (NaN with K not finite: poisoned inputs, left alone)
                marked |= e < 0.0;
            }
        }
        unsigned long long todo = __ballot(failed || marked);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int64_t first_l = bcast64(first, l), slot_l = bcast64(slot_i, l);
            const int ppg_l = __shfl(ppg, l);
            const bool failed_l = __shfl((int)failed, l) != 0;
            double *slot = prm.part ? prm.part + slot_l * (int64_t)prm.rows * prm.outw : nullptr;
            if (failed_l) {      // a pair the screen let through failed after the fact: the whole chunk again, exactly
                for (int c = threadIdx.x; slot && c < prm.rows * prm.outw; c += WAVE) slot[c] = 0.0;
                if (prm.N0)    // (one pair per chunk there)
                    for (int c = threadIdx.x; c < prm.n0cols; c += WAVE) prm.N0[first_l * prm.n0cols + c] = 0.0;
                __syncthreads();
            }
            for (int i = 0; i < ppg_l && first_l + i < prm.P; ++i) {
                const double e = prm.err[first_l + i];
                if (failed_l || e < 0.0) rescue_pair(prm, first_l + i, slot, lds, wsb);
            }
        }
    }
    for (int64_t gi = blockIdx.x; !short_chunks && gi < prm.n_groups; gi += gridDim.x) {
        int64_t first, slot_i;
        int ppg;
        chunk_share(prm.cs, gi, A, prm.B, prm.P, first, slot_i, ppg);
        if (first >= prm.P) continue;
        bool failed = false, marked = false;
        for (int i = threadIdx.x; i < ppg; i += WAVE) {
            if (first + i >= prm.P) break;
            const double e = prm.err[first + i];
            failed |= e > prm.tol || (e != e && prm.kfinal && isfinite(prm.kfinal[first + i]));   // (NaN with K not finite: poisoned inputs, left alone)
            marked |= e < 0.0;
        }
        failed = __any(failed);
        marked = __any(marked);
        if (!failed && !marked) continue;
        double *slot = prm.part ? prm.part + slot_i * (int64_t)prm.rows * prm.outw : nullptr;
        if (failed) {
            for (int c = threadIdx.x; slot && c < prm.rows * prm.outw; c += WAVE) slot[c] = 0.0;
            if (prm.N0)
                for (int c = threadIdx.x; c < prm.n0cols; c += WAVE) prm.N0[first * prm.n0cols + c] = 0.0;
            __syncthreads();
        }
        for (int i = 0; i < ppg && first + i < prm.P; ++i) {
            const double e = prm.err[first + i];
            if (failed || e < 0.0) rescue_pair(prm, first + i, slot, lds, wsb);
        }
    }
}

}  // namespace

size_t fused_rescue_block_doubles(int kind, int Mc, int Nc, int dyadic) {
    const size_t grid = (size_t)((Mc << dyadic) + 1) * ((Nc << dyadic) + 1);
    return 2 * (size_t)Mc * Nc + (kind == 1 ? (size_t)(Mc + 1) * (Nc + 1) : 0) + 2 * grid;
}

// workspace: [P doubles: the upstream gradient with the marked pairs set to NaN][blocks x fused_rescue_block_doubles]
size_t fused_rescue_workspace_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int blocks) {
    if (simple_lds_bytes(make_geom(1, Mc, Nc, dyadic, SK_SCHEME_DEFAULT)) > 160 * 1024) return 0;   // no stored-grid kernel for this grid
    return sizeof(double) * ((size_t)(P + 1) / 2 * 2 + (size_t)(blocks < 1 ? 1 : blocks) * fused_rescue_block_doubles(kind, Mc, Nc, dyadic));
}

int launch_fused_screen(const double *kfinal, const double *scale, int64_t P, double screen, double *scale_eff, double *err, hipStream_t s) {
    SK_LAUNCH(k_screen, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, kfinal, scale, P, screen, scale_eff, err);
    return check_launch();
}

int launch_fused_rescue(int kind, const double *Xs, const double *Ys, const double *scale, const double *err, double tol, double *part,
                        double *ypart, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g, int rows, int outw, int ycols,
                        double inv_sigma, const ChunkSplit &cs, int64_t n_groups, void *ws, size_t ws_bytes, hipStream_t s, int fd, double *n0,
                        int n0cols, const double *kfinal) {
    const size_t lds = simple_lds_bytes(g);
    if (lds > 160 * 1024) return SK_ERR_UNSUPPORTED;
    const size_t per_block = sizeof(double) * fused_rescue_block_doubles(kind, g.Mc, g.Nc, g.dyadic);
    if (!ws || ws_bytes < per_block) return SK_ERR_WORKSPACE;
    int64_t blocks = (int64_t)(ws_bytes / per_block);
    if (blocks > n_groups) blocks = n_groups;
    if (blocks > 1024) blocks = 1024;
    FusedRescueParams prm;
    prm.kind = kind; prm.Xs = Xs; prm.Ys = Ys; prm.scale = scale; prm.err = err; prm.tol = tol; prm.part = part; prm.Ypart = ypart;
    prm.A = A; prm.B = B; prm.P = g.P; prm.n_groups = n_groups; prm.Mrows = Mrows; prm.Ncp = Ncp; prm.Mc = g.Mc; prm.Nc = g.Nc; prm.D = D;
    prm.naive = g.naive;
    prm.kfinal = kfinal;
    prm.dyadic = g.dyadic; prm.rows = rows; prm.outw = outw; prm.ycols = ycols; prm.inv_sigma = inv_sigma; prm.cs = cs;
    prm.ws = (double *)ws; prm.ws_block = (int64_t)(per_block / sizeof(double));
    prm.fd = fd; prm.N0 = n0; prm.n0cols = n0cols;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_fused_rescue, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SK_LAUNCH(k_fused_rescue, dim3((unsigned)blocks), dim3(WAVE), lds, s, prm);
    return check_launch();
}

}  // namespace sk
