// sk_route.hip -- ONE place that says which kernel family serves a call (host code only; no device work, no HIP call).
//
// The host layer (sigkernel_amd/sigkernel.py), the launchers in this library and DESIGN.md used to carry three copies of the
// scope rules; sk_route_query is now the only one the host layer consults, tests/test_routes.py pins it against a table, and its
// GPU half checks that every launcher honours what it says: a FUSED / FUSED_MB answer means the launch succeeds and nothing of
// size pairs x M x N is ever allocated.
//
// SCOPE (kind 0 = exactly LinearKernel, 1 = exactly RBFKernel with sigma > 0; M, N = points of the two paths; D = path dimension;
// d = dyadic order; either stencil -- the _naive_solver one is a launch-time constant of every fused kernel): every call with
// D <= 16 and d <= 2 CAN run fused, forward and adjoint.  With SK_ROUTE_NO_STREAM the answer is never STREAM there.
//
//   forward (SK_OP_FORWARD: sigkernel.py:216-234, :362-382)
//     FUSED      one band per pair, path dim <= 8 (csrc/sk_wave_fused.hip):  rows <= 64 RC (RC = 4 / 2 / 1 coarse rows per lane at
//                d = 0 / 1 / 2; rows = M - 1 linear, M rbf);  rbf at d = 0 beyond dim 4 / the default stencil / fp64 paths: two rows per
//                lane (rows <= 128; the four-row 8-dim variants spill registers), and no edges kept
//     FUSED_SWAP the same kernel on (y, x) when only the SECOND paths fit one band (k(x, y) = k(y, x); a Gram caller transposes)
//     FUSED_MB   any number of bands, path dim <= 16, any M, N (csrc/sk_wave_fused_mb.hip); _SWAP: solved as k(y, x) -- the kernel
//                and both static kernels are symmetric -- when that orientation sweeps at most 80 % of the macro-steps
//   adjoint (SK_OP_ADJOINT: sigkernel.py:257-343, :404-502; the forward of a call with a gradient pending keeps the edges of the
//   SAME family: sk_solve_fwd_{linear,rbf}_edges_f64 for FUSED, sk_solve_fwd_static_* with edges for FUSED_MB)
//     FUSED      linear: dim <= 8, M - 1 <= 128 (64 at d = 2) (csrc/sk_wave_adj_fused.hip);
//                rbf: dim <= 4, d = 1..2, M <= 128 / 64; dim 5..8 at d = 1: M <= 64 (one row per lane); d = 0: dim <= 8, default stencil, M <= 128, two rows per lane (csrc/sk_wave_adj_fused_rbf.hip;
//                its 8-dim variants spill and lose)
//     FUSED_MB   dim <= 16, d = 0..2, any M, N (csrc/sk_wave_adj_fused_mb.hip; rbf at d = 0: two coarse rows per lane); never swapped
//                (the gradient is the first argument's)
//   STREAM       everything else (dim > 16, d > 2, other static kernels): static kernel -> increments in HBM -> sk_solve_fwd_* /
//                sk_static_increments -> sk_solve_adj -> sk_static_adjoint (or the generic vector-Jacobian route)
//
// CHOICE (the default, without SK_ROUTE_NO_STREAM).  The multi-band kernels sweep a pair with all 64 lanes over bands of 64 RC rows
// and at least 80 units of two columns: on SHORT paths most of that sweep is padding, and the streaming route -- whose transient
// memory the host layer bounds by tiling over rows anyway -- is several times faster (measured, same box, 256 x 256 pairs,
// profiles/r04_ab_routes.txt: linear dim 12, 40 x 40 points: 0.88 ms streamed against 3.9 ms multi-band; rbf dim 7, 64 x 64, d = 1
// with a gradient: 4.0 against 12.2 ms; at 128 x 128 points the multi-band route wins: 12.2 against 14.0 ms).  So FUSED_MB is only
// answered when the sweep's efficiency  rows / (bands 64 RC) x units / max(80, units)  reaches 0.45 (forward with the rbf kernel:
// 0.5; rbf with 9..16 dims of fp64 paths, whose 16 staged fp64 dimensions leave one wave per SIMD: never for the forward, 0.9 for
// the adjoint -- mb_min_eff has the numbers); below that the answer is STREAM.  The crossover sits at efficiency ~0.5 for every
// other adjoint measured.  A deployment that must not hold increments at all (memory
// first) passes SK_ROUTE_NO_STREAM (host layer: sigkernel_amd.routes.no_stream / SK_NO_STREAM=1).
#include "sk_internal.h"

namespace sk {

namespace {
constexpr int MB_MIN_UNITS = 80;   // units per band of the multi-band kernels (sk_wave_fused_mb.hip: MB_L + 16)

inline int rc_of(int d) { return d == 0 ? 4 : d == 1 ? 2 : 1; }
// macro-steps per pair of the multi-band forward: bands x units per band
inline long mb_steps(int kind, int Mc, int Nc, int d) {
    const int nb = (Mc + 64 * rc_of(d) - 1) / (64 * rc_of(d));
    int nup = ((kind == 1 ? (Nc + 2) / 2 : (Nc + 1) / 2) + 7) / 8 * 8;
    if (nup < MB_MIN_UNITS) nup = MB_MIN_UNITS;
    return (long)nb * nup;
}
// share of a multi-band sweep that is not padding: rows / (bands x 64 RC) x units / padded units
inline double mb_efficiency(int kind, int Mc, int Nc, int d, int rc) {
    const int rows = Mc + (kind == 1 ? 1 : 0);
    const int nb = (rows + 64 * rc - 1) / (64 * rc);
    const int nu = kind == 1 ? (Nc + 2) / 2 : (Nc + 1) / 2;
    int nup = (nu + 7) / 8 * 8;
    if (nup < MB_MIN_UNITS) nup = MB_MIN_UNITS;
    return (double)rows / (double)(nb * 64 * rc) * (double)nu / (double)nup;
}
// least sweep efficiency at which the multi-band kernels are the default (measured crossovers, profiles/r04_ab_routes.txt)
inline double mb_min_eff(int op, int kind, int D, int elem_size) {
    if (kind == 1 && D > 8 && elem_size == 8)   // 16 staged fp64 dims: one wave per SIMD (the fp32 ring of fp32 inputs holds two) --
        return op == SK_OP_FORWARD ? 1.01 : 0.9;   // forward 7.4 against 5.1 ms streamed at 0.8; with a gradient 35.7 against 27.5 at 0.8
    if (kind == 1 && op == SK_OP_FORWARD) return 0.5;   // at 0.4: 5.4 against 3.8 ms (dim 7, d = 0, 128 points)
    return 0.45;
}
}  // namespace

int route_query(int op, int kind, int D, int M, int N, int d, int naive, int elem_size, int flags) {
    const bool may_stream = !(flags & SK_ROUTE_NO_STREAM);
    if ((kind != 0 && kind != 1) || D < 1 || D > 16 || M < 2 || N < 2 || d < 0 || d > 2) return SK_ROUTE_STREAM;
    if (elem_size != 8 && elem_size != 4) return SK_ROUTE_STREAM;
    const int Mc = M - 1, Nc = N - 1;
    if (op == SK_OP_FORWARD) {
        const int rows = kind == 1 ? M : Mc;
        // (rbf at d = 0 beyond dim 4 / the default stencil / fp64 paths: the variant with TWO coarse rows per lane, 128 node rows)
        const bool two_rows = kind == 1 && d == 0 && (D > 4 || naive || elem_size != 8);
        bool one_band = D <= 8 && rows <= 64 * (two_rows ? 2 : rc_of(d));
        if (one_band) return SK_ROUTE_FUSED;
        {   // the same kernel on (y, x): k and both static kernels are symmetric; rows then come from the SECOND path
            const int rows_s = kind == 1 ? N : Nc;
            const bool one_band_s = D <= 8 && rows_s <= 64 * (two_rows ? 2 : rc_of(d));
            if (one_band_s) return SK_ROUTE_FUSED_SWAP;
        }
        const bool swap = 5 * mb_steps(kind, Nc, Mc, d) <= 4 * mb_steps(kind, Mc, Nc, d);
        const double eff = swap ? mb_efficiency(kind, Nc, Mc, d, rc_of(d)) : mb_efficiency(kind, Mc, Nc, d, rc_of(d));
        if (may_stream && eff < mb_min_eff(op, kind, D, elem_size)) return SK_ROUTE_STREAM;
        return swap ? SK_ROUTE_FUSED_MB_SWAP : SK_ROUTE_FUSED_MB;
    }
    if (op == SK_OP_ADJOINT) {
        if (kind == 0 && D <= 8 && Mc <= (d == 2 ? 64 : 128)) return SK_ROUTE_FUSED;
        if (kind == 1 && D <= 4 && d >= 1 && M <= 64 * rc_of(d)) return SK_ROUTE_FUSED;
        if (kind == 1 && D <= 8 && d == 0 && !naive && M <= 128) return SK_ROUTE_FUSED;   // two coarse rows per lane
        if (kind == 1 && D <= 8 && d == 1 && M <= 64) return SK_ROUTE_FUSED;               // dim 5..8: one coarse row per lane
        if (may_stream && mb_efficiency(kind, Mc, Nc, d, kind == 1 && d == 0 ? 2 : rc_of(d)) < mb_min_eff(op, kind, D, elem_size)) return SK_ROUTE_STREAM;
        return SK_ROUTE_FUSED_MB;
    }
    return SK_ROUTE_STREAM;
}

}  // namespace sk
