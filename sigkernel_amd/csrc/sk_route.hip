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
//     FUSED_SWAP rbf, dim <= 4, fp64, Gram calls: the one-band adjoint on (y, x) with the second-argument sums when only the SECOND paths
//                fit its lanes (64 points at d = 1..2, 128 at d = 0): d k(x, y) / dx = d2 k(y, x)
//     FUSED_MB   dim <= 16, d = 0..2, any M, N (csrc/sk_wave_adj_fused_mb.hip; rbf at d = 0: two coarse rows per lane); never swapped
//   STREAM       everything else (dim > 16, d > 2, other static kernels): static kernel -> increments in HBM -> sk_solve_fwd_* /
//                sk_static_increments -> sk_solve_adj -> sk_static_adjoint (or the generic vector-Jacobian route)
//
// CHOICE (the default, without SK_ROUTE_NO_STREAM).  The multi-band kernels sweep a pair with all 64 lanes over bands of 64 RC rows
// and at least 80 units of two columns: on SHORT paths most of that sweep is padding, and the streaming route -- whose transient
// memory the host layer bounds by tiling over rows anyway -- is faster.  Two numbers of the cost table below decide (measured,
// same box, 128 x 128 pairs, profiles/r05_thresholds.txt): STREAM while both paths have at most `stream_one_strip_cells` increments
// (the streaming kernels' one-strip regime: they win there at ANY efficiency, up to 2x), and beyond that while the sweep's efficiency
//   rows / (bands 64 RC) x units / max(80, units)   stays below `mb_min_eff`; FUSED_MB otherwise (0.5..0.9x the streamed time from
// ~140 points on, forward and adjoint, both static kernels, 8 or 16 staged dims).  A deployment that must not hold increments at all
// (memory first) passes SK_ROUTE_NO_STREAM (host layer: sigkernel_amd.routes.no_stream / SK_NO_STREAM=1).
#include "sk_internal.h"
#include <cstring>

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
// ---- the COST table: every measured crossover the library and the host layer decide by, in one place ----------------------------
// (scope rules say what a kernel CAN do; these say when it is the faster choice.  Each entry carries the same-box A/B measurement it
// came from; tools/crossovers.py re-measures them on the box at hand and says which sit within noise of flipping.  Box-to-box
// spread is +-5 %, so an entry whose two sides differ by less is a preference, not a cliff.)
struct CostEntry { const char *name; double value; const char *note; };
const CostEntry COSTS[] = {
    {"mb_min_eff", 0.45,
     "least sweep efficiency rows / (bands 64 RC) x units / max(80, units) at which the multi-band fused kernels are the default instead of the "
     "streaming route (forward and adjoint, LinearKernel and RBFKernel alike); r05_thresholds, 128 x 128 pairs, same box: linear dim 12 d = 0: 128 points "
     "(0.40) 1.38 ms multi-band vs 1.18 streamed, 140 points (0.48) 1.37 vs 2.04; rbf dim 7 d = 0, 140 points (0.48) 1.64 vs 2.06; with a "
     "gradient linear dim 12 d = 1, 140 points (0.48) 7.7 vs 9.0 ms"},
    {"stream_one_strip_cells", 128,
     "... but never while BOTH paths have at most this many increments: there the streaming kernels sweep a pair in one strip and win whatever "
     "the efficiency (r05_thresholds: linear dim 12 d = 1 with a gradient, 128 points (0.79): 4.14 ms multi-band vs 3.52 streamed, 100 points "
     "(0.48) 4.14 vs 2.75; rbf dim 12 d = 1, 128 points 5.99 vs 5.36; rbf dim 7 d = 1, 110 points 3.59 vs 3.27) -- round 4's per-kernel "
     "thresholds (0.5 / 0.9 / never) were this cliff seen through the efficiency"},
    {"mb_min_eff_rbf16_forward", 0.85,
     "the rbf FORWARD with 9..16 dims of fp64 paths (16 staged fp64 dims leave the multi-band kernel one wave per SIMD) needs fuller bands: "
     "r05_thresholds2, 64 x 64 pairs, dim 16: d = 2: 140 points (0.64) 1.25 ms multi-band vs 1.13 streamed, 200 (0.75) 1.95 vs 1.82, 300 (0.93) "
     "3.30 vs 3.46, 512 (1.0) 8.2 vs 11.1; d = 0: 140 (0.48) 0.95 vs 0.79, 300 (0.77) 2.59 vs 2.79; with a gradient the multi-band route wins "
     "from 140 points on (0.77 .. 0.39x) and takes the common threshold"},
    {"mb_swap_steps_ratio", 0.8, "forward: solve k(y, x) when that orientation sweeps at most this share of the macro-steps of k(x, y)"},
    {"sym_tiles", 8, "row blocks of compute_Gram(X, X, sym=True) beyond the one-band triangle launch: (T + 1) / (2 T) of the square is solved; "
     "16 with a gradient on 64 T rows and more (C4: -1 %)"},
    {"sym_min_cells", 5e9,
     "grid cells of K_XX from which the blocked triangle (and, for the loss wrappers, the composition of three Gram calls) beats one launch over "
     "the square / the merged block K(X, [X; Y]): 128 x 128 pairs of 64 points take 0.4 ms in one launch, 0.8 ms in 8 blocks; merged loss "
     "15.1 vs 15.5 ms composed at 512 x 512 (4e9 cells, r04_merged_loss), C4 (2.7e11) 349 ms composed"},
    {"sym_min_rows", 32, "least rows per block of the triangular adjoint: 64 paths of length 700 in 8 blocks of 8 rows: backward 60 -> 74 ms"},
    {"sym_stream_min_paths", 224,
     "compute_Gram(X, X, sym=True) without a gradient on the STREAMING route (wide paths, dyadic >= 3, user-defined kernels): paths from "
     "which the blocked triangle replaces the one block of all pairs whatever the grid -- the node evaluation and the increments are most of a "
     "pair's cost there (r06_sym_stream, rbf dim 20, 64 points, d = 1: 192 paths 0.91x, 256 paths 1.52 -> 1.09 ms, 512 paths 5.76 -> 3.48; "
     "rbf dim 3 d = 3: 192 paths 1.15x, 256 paths 0.84x)"},
    {"paired_merge_cells", 2e9,
     "grid cells of a paired batch below which compute_distance solves k(X, X) and k(X, Y) as ONE batch of 2n pairs (launch- and fill-bound there)"},
    {"mmd_streams_max_pairs", 16384,
     "pairs per Gram matrix up to which a CAPTURED composed compute_mmd forks its three matrices onto three streams (replays 10-15 % faster at 16..64 paths)"},
    {"keep_edges_fraction", 0.5, "share of the transient budget the edges kept between forward and backward may take"},
    {"keep_increments_fraction", 0.125,
     "share of the transient budget the STREAMING route's increments of a one-tile Gram block may keep between forward and backward (beside the edges): "
     "backward then skips the second evaluation of the node kernel -- r06_keep_inc, 256 x 256 pairs of 64 points, forward + backward: rbf dim 20 "
     "5.96 -> 5.02 ms, linear dim 20 3.89 -> 3.23; 2.1 GB held there, 6 GB at most under the default budget"},
    {"loss_launch_free_bytes", 268435456.0,
     "bytes the one-launch loss route (sk_solve_fwd_loss_f64: values, weights, pair table and the rectangle's edges, allocated in one piece and held until "
     "backward) may take without asking the device for its free memory -- hipMemGetInfo costs 20-30 us, a tenth of a 32 + 32-path step; above it the "
     "call is held to keep_edges_fraction of the budget like every other route (ADVICE r5)"},
    {"fused_mid_min_pairs_per_rank", 4,
     "pairs per lane group and rank from which a no-queue fused forward that fills the chip deals out shares by wave age rank: 64-row shard of the "
     "headline Gram 0.695 -> 0.640 ms, 128 + 128 path mmd step 1.237 -> 1.171 ms (r05, same box)"},
    {"fused_static_share_linear", 35,
     "per cent of a lane group's equal share the one-band fused forward deals out before its work queue (SK_FUSED_Q_STATIC overrides): the headline "
     "Gram 4.06 ms at 35, 4.09 at 20, 4.10 at 3, 4.42 at 65, 4.80 at 92 -- waves of a SIMD run at different speeds, the queue evens them out"},
    {"fused_static_share_rbf", 20,
     "... the RBF variants (longer macro-steps, a draw costs less of one): 2048 x 2048 pairs, dyadic 2: 52.0 ms at 35, 50.9 at 20, 50.6 at 3; "
     "512 / 1024 paths: 3.41 -> 3.34, 13.0 -> 12.8 ms; C4 335.6 -> 334.5 (profiles/r05_qstatic.txt)"},
    {"mb_split_max_resident_share", 0.5,
     "few pairs of long paths: bands of a pair on several waves when the pairs fill at most this share of the resident waves (r05: 16 x 16 pairs "
     "of 4096 points 18.7 -> 4.7 ms; 32 x 32 pairs of 700 points 2.18 -> 1.96 ms)"},
    {"mb_split_min_units", 256, "... and a band has at least this many two-column units (second paths of ~512 points): the trailing is cheap, and "
     "a band's first windows never reach into the last chunks of the band above"},
};
constexpr int N_COSTS = (int)(sizeof(COSTS) / sizeof(COSTS[0]));
inline double cost(const char *name) {
    for (int i = 0; i < N_COSTS; ++i)
        if (!strcmp(COSTS[i].name, name)) return COSTS[i].value;
    return 0.0;
}
// whether the streaming route is the default over the multi-band kernels at this shape
inline bool prefer_stream(int Mc, int Nc, double eff, bool rbf16_forward = false) {
    const int strip = (int)cost("stream_one_strip_cells");
    if (Mc <= strip && Nc <= strip) return true;
    // the streaming kernels sweep the rows in strips of `strip` too: first paths just over a multiple of it (129 .. ~145 increments) pad
    // THEIR last strip as the multi-band kernels pad their last band, so the multi-band efficiency is held against the streamed one
    // (round 6, profiles/r06_mb_threshold.txt: 128 x 128 pairs of 130 points, linear dim 8 d = 1 -- efficiency 0.41 -- forward 2.38 ms
    // streamed against 1.07 multi-band, with a gradient 7.30 against 3.75; every measurement behind mb_min_eff had one-strip paths)
    const double eff_stream = Mc > strip ? (double)Mc / (double)(((Mc + strip - 1) / strip) * strip) : 1.0;   // (one strip: as measured before)
    return eff < cost(rbf16_forward ? "mb_min_eff_rbf16_forward" : "mb_min_eff") * (rbf16_forward ? 1.0 : eff_stream);
}
}  // namespace

double cost_value(int which) { return which >= 0 && which < N_COSTS ? COSTS[which].value : 0.0; }
const char *cost_name(int which) { return which >= 0 && which < N_COSTS ? COSTS[which].name : nullptr; }
const char *cost_note(int which) { return which >= 0 && which < N_COSTS ? COSTS[which].note : nullptr; }
double cost_by_name(const char *name) { return cost(name); }

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
        const bool swap = (double)mb_steps(kind, Nc, Mc, d) <= cost("mb_swap_steps_ratio") * (double)mb_steps(kind, Mc, Nc, d);
        const double eff = swap ? mb_efficiency(kind, Nc, Mc, d, rc_of(d)) : mb_efficiency(kind, Mc, Nc, d, rc_of(d));
        if (may_stream && prefer_stream(Mc, Nc, eff, kind == 1 && D > 8 && elem_size == 8)) return SK_ROUTE_STREAM;
        return swap ? SK_ROUTE_FUSED_MB_SWAP : SK_ROUTE_FUSED_MB;
    }
    if (op == SK_OP_ADJOINT_SYM) {
        // compute_Gram(X, X, sym=True) with a gradient: FUSED = the triangle through the one-band rbf adjoint with the second-argument
        // sums (a pair above the diagonal also stands for its mirror image); anything else = all pairs / the caller's other routes.
        // fp64 paths, dim <= 4; one coarse row per lane at dyadic 1 and 2 (M <= 64), two at dyadic 0 (M <= 128) -- the two-row form
        // at dyadic 1 spilled and lost to all pairs, profiles/r05_yside_ab.txt
        if (kind == 1 && D <= 4 && elem_size == 8 && route_query(SK_OP_ADJOINT, kind, D, M, N, d, naive, elem_size, flags) == SK_ROUTE_FUSED &&
            M == N && M <= (d == 0 ? 128 : 64))
            return SK_ROUTE_FUSED;
        return SK_ROUTE_STREAM;
    }
    if (op == SK_OP_ADJOINT) {
        if (kind == 0 && D <= 8 && Mc <= (d == 2 ? 64 : 128)) return SK_ROUTE_FUSED;
        if (kind == 1 && D <= 4 && d >= 1 && M <= 64 * rc_of(d)) return SK_ROUTE_FUSED;
        if (kind == 1 && D <= 8 && d == 0 && !naive && M <= 128) return SK_ROUTE_FUSED;   // two coarse rows per lane
        if (kind == 1 && D <= 8 && d == 1 && M <= 64) return SK_ROUTE_FUSED;               // dim 5..8: one coarse row per lane
        // (fp32 paths: the host layer up-casts them, as it does for SK_ROUTE_FUSED -- the one-band kernels sweep in fp64 whatever the dtype)
        // long first paths against short second ones (rbf, dim <= 4; Gram calls -- the host passes SK_ROUTE_NO_SWAP for
        // paired batches): the one-band adjoint on (y, x) with the SECOND-argument sums, d k(x, y) / dx = d2 k(y, x) (k and the static
        // kernel are symmetric), where the second paths fit its lanes -- 0.55-0.65x the streamed time, profiles/r05_asym.txt
        // (the second-argument sums ALONE -- no first-argument accumulators -- fit with two rows per lane at dyadic 1: 128 points)
        if (!(flags & SK_ROUTE_NO_SWAP) && kind == 1 && D <= 4 && (d == 0 ? (!naive && N <= 128) : N <= (d == 1 ? 128 : 64)))
            return SK_ROUTE_FUSED_SWAP;
        // dim 5..8 (dyadic 0 and 1: the one-band adjoint of that width exists there): the second-argument sums INSTEAD of the first-argument
        // ones, which is all the swapped call needs (sk_wave_adj_fused_rbf.hip, YONLY; round 6)
        if (!(flags & SK_ROUTE_NO_SWAP) && kind == 1 && D <= 8 && (d == 0 ? (!naive && N <= 128) : (d == 1 && N <= 64)))
            return SK_ROUTE_FUSED_SWAP;
        // ... and the linear one-band adjoint on (y, x) (dim <= 8, fp64 paths): its second-argument form hands the sums over a lane's rows
        // down the wave by DPP instead of keeping first-argument sums in registers (sk_wave_adj_fused.hip, round 6) -- 0.40-0.87x the time
        // of the routes below at 128 x 128 pairs, within 1.16-1.33x of the other orientation (profiles/r06_asym.txt, r06_asym_xy.txt)
        if (!(flags & SK_ROUTE_NO_SWAP) && kind == 0 && D <= 8 && Nc <= (d == 2 ? 64 : 128)) return SK_ROUTE_FUSED_SWAP;
        if (may_stream && prefer_stream(Mc, Nc, mb_efficiency(kind, Mc, Nc, d, kind == 1 && d == 0 ? 2 : rc_of(d)))) return SK_ROUTE_STREAM;
        return SK_ROUTE_FUSED_MB;
    }
    return SK_ROUTE_STREAM;
}

}  // namespace sk
