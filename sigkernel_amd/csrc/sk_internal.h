// sk_internal.h -- declarations shared by the translation units of libsigkernel_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sigkernel_amd.h"

namespace sk {

// Tuning knobs: the SK_* environment variables are parsed ONCE, when the library is loaded (sk_abi.hip), into this immutable
// struct; no launch path reads the environment (tests/test_abi.py greps the launch translation units for getenv).
// 0 / n = 0 mean "the built-in default".  Tools that sweep a knob inside one process call sk_reload_knobs() after changing it.
struct RankW { int n; double w[4]; };      // shares of the pairs by wave age rank, per cent; n = number of ranks given
struct Knobs {
    int wave_wpc, wave_wpb;
    int adj_wpc, adj_wpb;
    int adjf_wpc, adjf_wpb;
    int adjr_wpc, adjr_wpb;
    int adjmb_wpc, adjmb_wpb, adjmb_q_static;     // sk_wave_adj_fused_mb.hip
    int derivf_wpc, derivf_wpb, derivf_noshift;   // sk_wave_deriv_fused.hip
    int deriv_wpc, deriv_wpb;
    int fused_wpc, fused_wpb, fused_q_static, fused_mid;
    int fusedmb_wpc, fusedmb_wpb, fusedmb_q_static, fusedmb_split, fusedmb_lead;
    RankW rank_w, wave_rank_w, adj_rank_w, adjf_rank_w, adjr_rank_w, deriv_rank_w, fused_rank_w, fusedmb_rank_w;
};
const Knobs &knobs();                      // sk_abi.hip
// sk_route.hip: the cost table (measured crossovers the library and the host layer decide by)
double cost_value(int which);
const char *cost_name(int which);
const char *cost_note(int which);
double cost_by_name(const char *name);
int device_cu_count();                     // sk_abi.hip: compute units of the current device (256 on MI355X), cached per device

// ---- launch trace (diagnostics, sk_abi.hip) -------------------------------------------------------------------------------------------
// EVERY kernel launch of the library goes through SK_LAUNCH.  With tracing on (sk_launch_trace(1), or SK_TRACE_LAUNCHES=1 in the
// environment at load) the launches are counted per kernel instance -- by the instance's host stub, named by its device symbol when the
// counts are dumped (sk_launch_trace_dump) -- so that a sweep of the public API can be checked against the list of instances the build
// contains without a profiler (tools/reach_sweep.py, tests/test_abi.py: no unreachable instance).  Off: one relaxed load per launch.
bool trace_on();
void trace_launch(const void *host_stub);
#define SK_LAUNCH(kern, ...)                                                   \
    do {                                                                        \
        if (::sk::trace_on()) ::sk::trace_launch((const void *)(kern));         \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                                  \
    } while (0)

// The fused adjoints' decomposition (sk_wave_common.h: chunk_split / chunk_share): a lane group sweeps one CHUNK of the B pairs
// of one path x_a and leaves a partial sum in slot a * nch + c; chunks swept by the oldest waves are longer.
struct ChunkSplit {
    int nr;              // 1: nch chunks of (at most) size[0] pairs, group gi = a * nch + c
    int cpr, nch;        // chunks of one a per rank; chunks of one a
    int64_t gpr;         // lane groups per rank (= waves per rank * G)
    int size[4];         // pairs in a chunk of rank r
    int off[4];          // first pair (within the B of an a) of rank r's chunks
    int uneven;          // nr == 1 and nch does not divide B: chunk c of an a is [c B / nch, (c + 1) B / nch) -- lengths differ by one.
    int ppp;             // paired batches (B == 0): pairs per lane group (0 / 1: one); every pair keeps its own slot
    int B;               // (round 6: until then the chunk length had to DIVIDE B, and a batch without a suitable divisor -- a prime
                         //  number of paths, 640, 768, 1536 -- left most lane groups idle: 127 x 127 pairs 10x slower than 128 x 128)
};

// Device-side rescue of the fused adjoints (sk_adj_fused_rescue.hip).
// What the fused adjoint launchers need for it (nullptr: no rescue, the residuals are the caller's to look at):
struct FusedRescue {
    const double *kfinal;   // [P] forward values K[MM][NN], nullable: pairs with |K| > screen are skipped by the sweep and solved exactly
    double screen, tol;     // tol: self-check residual above which a pair that was NOT screened makes its chunk be recomputed
    void *ws;               // fused_rescue_workspace_bytes(kind, P, Mc, Nc, dyadic, blocks)
    size_t ws_bytes;
};

// Geometry of one solve call; MM/NN are fine-grid cell counts.
struct Geom {
    int64_t P;
    int Mc, Nc, dyadic;
    int MM, NN;
    int naive;
    int64_t ld;   // row stride of inc_c in elements (>= Nc)
};

inline Geom make_geom(int64_t P, int Mc, int Nc, int dyadic, int scheme, int64_t ld = 0) {
    Geom g;
    g.P = P; g.Mc = Mc; g.Nc = Nc; g.dyadic = dyadic;
    g.MM = Mc << dyadic; g.NN = Nc << dyadic;
    g.naive = (scheme == SK_SCHEME_NAIVE);
    g.ld = ld > 0 ? ld : Nc;
    return g;
}

// ---- sk_simple.hip: one wavefront per pair, anti-diagonal sweep, FMA-free arithmetic -------
template <typename T>
int launch_fwd_simple(const T *inc_c, const Geom &g, T *out_final, T *out_grid, double *out_edges, hipStream_t s);
template <typename T>
int launch_adj_simple(const T *inc_c, const Geom &g, T *out_final, T *W, int64_t ldw, void *ws, size_t ws_bytes, hipStream_t s);
// K, K_gamma, K_gamma_gamma (directional derivatives) from three increment arrays of equal layout
template <typename T>
int launch_deriv_simple(const T *inc, const T *inc_d, const T *inc_dd, const Geom &g, T *out_k, T *out_kd, T *out_kdd,
                        hipStream_t s);
template <typename T>
int launch_adj_rescue(const T *inc_c, const Geom &g, const double *err, double tol, T *out_final, T *W, int64_t ldw, void *ws,
                      size_t ws_bytes, hipStream_t s);
size_t adj_simple_workspace_bytes(const Geom &g);
size_t simple_lds_bytes(const Geom &g);

// ---- sk_wave.hip: skewed row-strip wavefront sweep, register-resident state, LDS-DMA staging ----
// SK_ERR_UNSUPPORTED = shape/layout not covered; the caller falls back to the simple kernel.
// `strip_edges` (nullable, internal): the pair's terminal row and column in the padded strip layout
// [P][NNp + MMp] = K[MM][1..NNp], K[1..MMp][NN] that launch_adj_wave reads; strip_edge_doubles() gives NNp + MMp
// (0 when the strip kernels do not cover the shape).
template <typename T>
int launch_fwd_wave(const T *inc_c, int64_t ld, const Geom &g, T *out_final, double *strip_edges, hipStream_t s);

// Strip decomposition shared by the forward and adjoint wave kernels for dyadic 1..2 (where both use the same rows per
// lane): NUp 16-byte units per padded row, L = 1 << logL lanes per pair, nb bands, RC coarse rows per lane.
struct Strip {
    int NUp, logL, nb, RC;
    int NNp, MMp;   // padded fine columns / rows
    bool ok;
};
inline Strip strip_geom(const Geom &g, int elem_size) {
    Strip st{};
    const int CW = 16 / elem_size, DY = g.dyadic;
    st.RC = DY == 0 ? 4 : DY == 1 ? 2 : 1;
    const int NU = (g.Nc + CW - 1) / CW;
    st.NUp = (NU + 7) / 8 * 8;
    st.logL = 3;
    while (st.logL < 6 && (st.RC << st.logL) < g.Mc) ++st.logL;
    int L = 1 << st.logL;
    st.nb = (g.Mc + L * st.RC - 1) / (L * st.RC);
    st.ok = true;
    if (st.nb > 1) {
        // band b+1 reads what band b's bottom lane wrote L-1 macro-steps after the top lane: needs NUp >= L
        while (L > st.NUp && st.logL > 3) { --st.logL; L >>= 1; }
        if (L > st.NUp) st.ok = false;
        st.nb = (g.Mc + L * st.RC - 1) / (L * st.RC);
    }
    st.NNp = (st.NUp * CW) << DY;
    st.MMp = (st.nb * L * st.RC) << DY;
    return st;
}
// The one-band RBF kernels sweep NODE columns: column 2 NUp of the strip must be padding, which the increment layout has not when
// N - 1 is a multiple of 16.  Their edges are then kept in the layout of a strip one column wider (one more line of 8 units);
// sk_strip_edges_bytes(P, Mc, Nc + 1, ...) sizes it.  (The streaming adjoint sees a size it does not expect and sweeps forward itself.)
// Likewise they sweep NODE rows, M = Mc + 1 of them: when the increment rows fill the strip's lanes exactly (Mc = RC 2^k), the node rows
// need the next power of two of lanes, and the edges are kept in the layout of a strip one row taller (its padded rows doubled).
inline Geom rbf_edge_geom(const Geom &g) {
    Geom e = g;
    if (g.Nc % 16 == 0) { e.Nc += 1; e.NN = e.Nc << e.dyadic; }
    const int rc = g.dyadic == 0 ? 4 : g.dyadic == 1 ? 2 : 1;
    int cap = 8 * rc;
    while (cap < g.Mc) cap *= 2;
    if (cap == g.Mc && cap < 64 * rc) { e.Mc += 1; e.MM = e.Mc << e.dyadic; }
    return e;
}
inline size_t strip_edge_doubles(const Geom &g, int elem_size) {
    const Strip st = strip_geom(g, elem_size);
    return st.ok ? (size_t)st.NNp + (size_t)st.MMp : 0;
}

// ---- sk_wave_adj.hip: fused reverse sweep + backward recompute of K (needs the forward kernel's edges) ----
template <typename T>
int launch_adj_wave(const T *inc_c, int64_t ld, const Geom &g, const double *edges, T *W, int64_t ldw, double *err,
                    hipStream_t s);

// ---- sk_wave_deriv.hip: K, K_gamma, K_gamma_gamma in one skewed sweep, three increment streams ----
template <typename T>
int launch_deriv_wave(const T *inc, const T *inc_d, const T *inc_dd, int64_t ld, const Geom &g, T *out_k, T *out_kd,
                      T *out_kdd, hipStream_t s);

// ---- sk_wave_fused.hip: forward solver with the linear static kernel fused in (no increments in HBM) ----
template <typename TO>
int launch_fwd_fused_linear(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                            TO *out, double *strip_edges, void *queue, hipStream_t s, int tri = 0, const int64_t *loss = nullptr);

template <typename TO>
int launch_fwd_fused_rbf(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                         double inv_sigma, TO *out, double *strip_edges, void *queue, hipStream_t s, int tri = 0, const int64_t *loss = nullptr);
// (tri = 2, loss = {tri_n, tri_off}: the LOSS layout -- both staged arrays hold one batch Z of B paths; A B rectangle pairs (rows
// Z[0 .. A)), which keep their edges, then the strict upper triangle of Z[tri_off .. tri_off + tri_n), which does not; out [P] linear)

// ---- sk_loss.hip: the glue of the loss wrappers (compute_mmd / scoring rules) as single launches ----
template <typename T>
int launch_prep_cat(const T *X, int64_t A, const T *Y, int64_t B, int M, int D, int diff, double scale_rows, double scale_rows2, double *out_rows,
                    double *out_rows2, int rows, double *out_cols, int cols, int FDp, int *pair_tab, int64_t tri_n, hipStream_t s);
int launch_loss_value(const double *out, int64_t A, int64_t B, int with_yy, double *value, double *wb, hipStream_t s);
int launch_loss_weights(int64_t A, int64_t B, const double *grad_out, double *go, hipStream_t s);
int launch_rbf_adjoint_finish(const double *gpart, int64_t A, int64_t chunks, int rows, int outw, const double *X, int M, int D, double sigma,
                              const double *gscale, double *grad, hipStream_t s);
int launch_linear_adjoint_finish(const double *tpart, int64_t A, int64_t chunks, int rows, int M, int D, double scale2, const double *gscale,
                                 double *grad, hipStream_t s);

// ---- sk_wave_fused_mb.hip: the same for pairs that need several bands, and path dims up to 16 (kind 0 linear, 1 rbf) ----
template <typename TO>
int launch_fwd_fused_mb(int kind, const double *Xr, const void *Yt, int yt_f32, int64_t A, int64_t B, int Mrows, int Ncp, int D,
                        int fd, const Geom &g, double inv_sigma, TO *out, double *edges, void *ws, size_t ws_bytes, hipStream_t s);
size_t fused_mb_workspace_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int D);
int fused_mb_split(int kind, int64_t P, int Mc, int Nc, int dyadic, int D);
int fused_mb_rows(int kind, int Mc, int dyadic, bool edges = false);
int fused_mb_cols(int kind, int Nc);

// ---- sk_route.hip: which kernel family serves a call (host only) ----
int route_query(int op, int kind, int D, int M, int N, int dyadic, int naive, int elem_size, int flags);

// ---- sk_wave_deriv_fused.hip: k, d/dgamma, d2/dgamma2 with the static kernel fused in (no increment arrays in HBM) ----
size_t deriv_fused_workspace_bytes(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows);
int launch_deriv_fused(int kind, const double *X0r, const double *X1r, const double *X2r, const double *Yt, int64_t A, int64_t B, int Mrows,
                       int Ncp, int D, int fd, const Geom &g, double inv_sigma, double eps, double *out_k, double *out_kd, double *out_kdd,
                       void *ws, size_t ws_bytes, hipStream_t s);

// ---- sk_wave_adj_fused_mb.hip: the fused RBF adjoint for pairs of several bands / path dims up to 16 ----
bool adj_fused_mb_layout(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows, int *rows, int *outw, int *ncols, int64_t *edge_doubles,
                         int *nb, int *nup, size_t *ws_bytes, int kind = 1);
int launch_adj_fused_linear_mb(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Ncp, int D, int fd, const Geom &g,
                               const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *err, void *ws,
                               size_t ws_bytes, const FusedRescue *rescue, hipStream_t s);
int launch_adj_fused_rbf_mb(const double *Xr, const void *Yt, int yt_f32, int64_t A, int64_t B, int Mrows, int Ncp, int D, int fd, const Geom &g,
                            double inv_sigma, const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *n0,
                            size_t n0_doubles, double *err, void *ws, size_t ws_bytes, const FusedRescue *rescue, const double *Yt64,
                            hipStream_t s);

// ---- sk_wave_adj_fused.hip: adjoint with the linear static kernel fused in (no increments, no W in HBM) ----
int launch_adj_fused_linear(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Ncp, const Geom &g,
                            const double *edges, const double *scale, double *tpart, size_t tpart_doubles, double *err, double *ypart,
                            size_t ypart_doubles, int *ppg_out, int *rows_out, int *ycols_out, const FusedRescue *rescue, hipStream_t s);

// ---- sk_prep.hip: fp64, zero-padded, row-major / dimension-major staging of the paths for the fused kernels ----
template <typename T>
int launch_prep_pair(const T *X, int64_t A, int M, const T *Y, int64_t B, int N, int D, int diff, double scale_x, double scale_y, double *out_x,
                     int rows_x, double *out_y, int rows_y, int FDp, hipStream_t s);
template <typename T>
int launch_prep_paths(const T *X, int64_t A, int M, int D, int diff, int dim_major, double scale, double *out, int rows, int FDp,
                      hipStream_t s);

// ---- sk_wave_adj_fused_rbf.hip: adjoint with the RBF static kernel fused in (nodes, increments, contraction in the sweep) ----
int launch_adj_fused_rbf(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g,
                         double inv_sigma, const double *edges, const double *scale, double *gpart, size_t gpart_doubles, double *err,
                         double *ypart, size_t ypart_doubles, int want_yside, int *ppg_out, int *rows_out, int *outw_out, int *ycols_out,
                         const FusedRescue *rescue, hipStream_t s);

// ---- sk_adj_fused_rescue.hip: device-side rescue of the fused adjoints (exploding kernels) ----
size_t fused_rescue_workspace_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int blocks);
int launch_fused_screen(const double *kfinal, const double *scale, int64_t P, double screen, double *scale_eff, double *err, hipStream_t s);
int launch_fused_rescue(int kind, const double *Xs, const double *Ys, const double *scale, const double *err, double tol, double *part,
                        double *ypart, int64_t A, int64_t B, int Mrows, int Ncp, int D, const Geom &g, int rows, int outw, int ycols,
                        double inv_sigma, const ChunkSplit &cs, int64_t n_groups, void *ws, size_t ws_bytes, hipStream_t s, int fd = 8, double *n0 = nullptr,
                        int n0cols = 0, const double *kfinal = nullptr);

// ---- sk_increments.hip ------------------------------------------------------------------
template <typename T>
int launch_increments(const T *G, int64_t P, int M, int N, T *inc_c, int64_t ld, hipStream_t s);
template <typename T>
int launch_increments_adjoint(const T *W, int64_t ldw, const T *scale, int64_t P, int M, int N, T *dG, hipStream_t s);

template <typename T>
int launch_deriv_increments(const T *G0, const T *G1, const T *G2, double eps, int64_t P, int M, int N, T *inc, T *inc_d,
                            T *inc_dd, int64_t ld, hipStream_t s);

// ---- sk_static.hip: static kernel (linear / rbf) + increments in one pass ---------------------------------
template <typename T>
int launch_static_increments(int kind, double param, const T *X, const T *Y, int64_t A, int64_t B, int M, int N, int D,
                             T *inc, int64_t ld, hipStream_t s);

template <typename T>
int launch_static_deriv_increments(int kind, double param, const T *X0, const T *X1, const T *X2, const T *Y, int64_t A,
                                   int64_t B, int M, int N, int D, double eps, T *inc, T *inc_d, T *inc_dd, int64_t ld,
                                   hipStream_t s);

template <typename T>
int launch_static_adjoint(int kind, double param, const T *X, const T *Y, const T *W, int64_t ldw, const T *scale, int64_t A,
                          int64_t B, int M, int N, int D, T *out, hipStream_t s);

template <typename T>
int launch_linear_adjoint_dyt(const double *dYt, int64_t ldy, const T *W, int64_t ldw, const T *scale, int64_t A, int64_t B,
                              int Mc, int Nc, int D, T *out, hipStream_t s);

template <typename T>
int launch_static_adjoint2(int kind, double param, const T *X, const T *Y, const double *dXr, int Mrows, const T *W, int64_t ldw,
                           const T *scale, int64_t A, int64_t B, int b0, int M, int N, int D, T *out, hipStream_t s);

inline int check_launch() {
    return hipGetLastError() == hipSuccess ? SK_OK : SK_ERR_LAUNCH;
}

// ---- device math shared by the RBF kernels (sk_wave_fused.hip, sk_static.hip) ----------------------------------------
// exp(x) for finite x <= 0 (the RBF exponent): n = rint(x log2 e), r = x - n ln 2 (two-term ln 2), a polynomial in |r| <= ln2 / 2,
// result scaled by 2^n; underflow goes through v_ldexp to 0.  Without the range checks and special cases of the library exp this is
// 20 VALU instructions.  The polynomial is 1 + r + r^2 / 2 + sum_{k = 3 .. DEG} c_k r^k with the c_k of the MINIMAX fit of that form
// (relative error; Remez in 80-digit arithmetic, tools/experiments/r06_exp_minimax.py; the three low terms are inline constants and
// stay Taylor's): DEG = 11: 2.4e-17 -- a fifth of an ulp, below the rounding of its own Horner steps; rounds 1-5 ran the Taylor
// polynomial of degree 13 (4e-18) for two more FMAs per node, and a node is 40 % of the fp64 work of a dyadic-1 RBF forward;
// DEG = 9: 1.1e-13 -- for the fp32-ring kernel of sk_wave_fused_mb.hip, whose results are fp32 (rounds 3-5: Taylor degree 10, 2e-13).
// The coefficients that are not inline constants live in VGPRs: as SGPR pairs they push the kernel's scalar state into spills
// (v_readlane in the hot loop).
// x below -800 -> a value in (-800.0000001, -800]: exp of it is already 0 in fp64, and n stays inside the int range for absurdly
// distant points.  A select, not fmax: a NaN exponent (NaN / inf coordinates) must stay NaN, as in the reference -- and a select of the
// HIGH word only (0xC0890000 = the high word of -800.0; whatever the low word holds, the value stays within 1e-7 of -800): one
// v_cndmask instead of two.
__device__ __forceinline__ double exp_clamp(double x) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    return __hiloint2double(x < -800.0 ? (int)0xC0890000u : hi, lo);
}
template <int DEG> struct ExpPoly;
template <> struct ExpPoly<11> {
    static constexpr int N = 9;      // r^11 .. r^3
    static constexpr double k[N] = {2.297635202637367e-08, 2.7622226391501255e-07, 2.756385758315426e-06, 2.4801516346828486e-05, 0.00019841262672752407,
                                    0.001388888892051129, 0.00833333333652807, 0.04166666666661997, 0.16666666666661958};
};
template <> struct ExpPoly<9> {
    static constexpr int N = 7;      // r^9 .. r^3
    static constexpr double k[N] = {2.4813607735638368e-06, 2.4869465890354224e-05, 0.00019848113928851274, 0.0013888836980909258, 0.008333328086765797,
                                    0.041666666786023716, 0.1666666667871994};
};
template <int DEG> struct ExpCoefT {
    double c[ExpPoly<DEG>::N];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < ExpPoly<DEG>::N; ++i) {
            c[i] = ExpPoly<DEG>::k[i];
            asm volatile("" : "+v"(c[i]));
        }
    }
};
typedef ExpCoefT<11> ExpCoef;
template <int DEG>
__device__ __forceinline__ double exp_nonpos(double x, const ExpCoefT<DEG> &e) {
    x = exp_clamp(x);
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = e.c[0];
#pragma unroll
    for (int i = 1; i < ExpPoly<DEG>::N; ++i) p = fma(p, r, e.c[i]);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}
// the same with the coefficients left to the compiler (kernels whose scalar register file has room for them)
__device__ __forceinline__ double exp_nonpos(double x) {
    x = exp_clamp(x);
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = ExpPoly<11>::k[0];
#pragma unroll
    for (int i = 1; i < ExpPoly<11>::N; ++i) p = fma(p, r, ExpPoly<11>::k[i]);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

}  // namespace sk
