/*
 * sigkernel_amd.h -- C ABI of libsigkernel_amd.so, the MI355X (gfx950) engine for the
 * signature-PDE-kernel hot path of crispitagorico/sigkernel.
 *
 * The reference has no FFI: its boundary is the Python signature of the solver
 * back-ends called from sigkernel/sigkernel.py.  Each entry point below names
 * the reference call it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - P independent problems ("pairs"): P = A for compute_kernel, P = A*B for
 *     compute_Gram with pair (a,b) at index a*B+b.
 *   - M, N   : number of points of the two paths; Mc = M-1, Nc = N-1 coarse cells.
 *   - dyadic : dyadic_order d; the fine grid has MM = Mc<<d by NN = Nc<<d cells.
 *     Dyadic refinement is index arithmetic inside the kernels
 *     (fine increment (i,j) = inc_c[i>>d][j>>d] / 4^d); the refined matrix that
 *     the reference materialises with tile() (sigkernel.py:218, :364) never exists.
 *   - all arrays are row-major device pointers owned by the caller; the library keeps no
 *     reference after return and allocates nothing.  Increment matrices carry a row stride
 *     `ld` (in elements, ld >= Nc; 0 means dense, ld = Nc): the fast kernels want 16-byte
 *     aligned rows (ld*sizeof(T) % 16 == 0, base pointer 16-byte aligned, which
 *     sk_increments_* produces when asked); any other layout is served by the simple
 *     kernels.  Everything else is dense.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call only enqueues work on it -- no hidden synchronisation.
 *   - return value: SK_OK or an sk_status error code (see sk_status_string).
 *   - f32 entry points read/write float but carry the PDE state in double.
 */
#ifndef SIGKERNEL_AMD_H
#define SIGKERNEL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sk_status {
    SK_OK = 0,
    SK_ERR_BAD_ARG = 1,      /* null pointer, non-positive size, unknown scheme/flag     */
    SK_ERR_UNSUPPORTED = 2,  /* shape outside what the kernels implement (see message)  */
    SK_ERR_LAUNCH = 3,       /* hipGetLastError() after the launch was not hipSuccess   */
    SK_ERR_WORKSPACE = 4,    /* workspace pointer null or too small                     */
    SK_ERR_NO_DEVICE = 5     /* no HIP device visible                                   */
} sk_status;

/* scheme: which finite-difference stencil (sigkernel/cython_backend.pyx:114-116,
 * sigkernel/cuda_backend.py:150-156). */
#define SK_SCHEME_DEFAULT 0 /* (k10+k01)(1+g/2+g^2/12) - k00(1-g^2/12)  -- _naive_solver=False */
#define SK_SCHEME_NAIVE 1   /* (k10+k01)(1+g/2) - k00                    -- _naive_solver=True  */

/* flags (bit-or) */
#define SK_FLAG_NONE 0
#define SK_FLAG_EXACT 1  /* FMA-free arithmetic in the reference's operand order: results are
                            bit-identical to the reference's Cython CPU solver (slower path)   */
#define SK_FLAG_SIMPLE 2 /* force the simple one-wavefront-per-pair anti-diagonal kernels       */
#define SK_FLAG_EDGES_GIVEN 8 /* sk_solve_adj_* only: `workspace` already holds the strip edges that
                               * sk_solve_fwd_edges_* wrote for these increments -- skip the forward sweep */
#define SK_FLAG_FAST_ONLY 4 /* never fall back: SK_ERR_UNSUPPORTED if the tiled kernels do not
                               cover the shape/layout (used by tests and benchmarks)            */

/* 330.  The number moves whenever an exported signature changes incompatibly: 310 -> 320 gave sk_solve_fwd_{linear,rbf}_sym_* their
 * pair_tab argument (position 3) and added the sk_prep_cat_* / sk_solve_fwd_loss_f64 / sk_loss_* / sk_*_adjoint_finish_f64 family;
 * 320 -> 330 gave sk_linear_adjoint_fused_f64 its ypart / ypart_doubles / ycols_out arguments (the second-argument sums).  A binding
 * written against an older number must not load this library silently (sigkernel_amd/_lib.py checks it at load). */
int sk_version(void);
/* "sigkernel_amd gfx950; sources <hash>; <hipcc --version>; ISA hazard lint passed at build": the sources and the toolchain this
 * binary was built from (static string).  The hand-scheduled kernels are linted at build time against the register allocator of
 * THAT compiler (csrc/Makefile); a deployment that rebuilds with another one gets a different string -- and a fresh lint. */
const char *sk_build_info(void);

/* ---- routing (host only: no device work) --------------------------------------------------------------------------------------
 * Which kernel family serves a call -- the ONE statement of the fused kernels' scope: the host layer asks it instead of keeping
 * its own predicates, and every launcher honours it (csrc/sk_route.hip holds the rules; tests pin them against a table).
 *   op     SK_OP_FORWARD: k_sig values (sigkernel.py:216-234, :362-382);  SK_OP_ADJOINT: the gradient (sigkernel.py:257-343, :404-502)
 *          -- the forward of a call with a gradient pending keeps the edges of the family the ADJOINT query names
 *   kind   0 = LinearKernel, 1 = RBFKernel (sigma > 0); anything else streams
 *   D, M, N, dyadic, scheme (SK_SCHEME_*), elem_size (8 / 4: the dtype of the caller's paths)
 *   flags  SK_ROUTE_NO_STREAM: never answer STREAM where a fused kernel exists (memory first).  Without it the multi-band kernels
 *          are only chosen where they are not mostly sweeping padding (short paths: the streaming route, whose transient memory the
 *          caller bounds by tiling over rows, is several times faster there -- csrc/sk_route.hip has the rule and the measurements)
 * Returns
 *   SK_ROUTE_STREAM         static kernel -> increments in HBM (pairs x M x N) -> sk_solve_fwd_* / sk_solve_adj_*
 *   SK_ROUTE_FUSED          one band per pair: sk_solve_fwd_linear_* / _rbf_* (+ _edges_f64), sk_linear_adjoint_fused_f64,
 *                           sk_rbf_adjoint_fused_f64 -- nothing of size pairs x M x N exists
 *   SK_ROUTE_FUSED_MB       several bands / wide paths: sk_solve_fwd_static_*, sk_linear_adjoint_fused_mb_f64, sk_rbf_adjoint_fused_mb_f64
 *   SK_ROUTE_FUSED_MB_SWAP  (forward only) sk_solve_fwd_static_* on (Y, X): k is symmetric and that orientation is cheaper
 *   SK_ROUTE_FUSED_SWAP     the one-band kernels on (Y, X): the second paths fit one band (rows <= 64 RC), the first do not; for the ADJOINT
 *                           (Gram; dim <= 8 -- rbf of dim 5..8 at dyadic 0 and 1; fp32 paths up-cast by the caller): sk_rbf_adjoint_fused_f64 / sk_linear_adjoint_fused_f64 on
 *                           (Y, X) with the second-argument sums (rbf, 128 x 128 pairs of 700 x 20 points: 0.41 ms against 1.56 ms
 *                           streamed); Gram callers transpose the result
 * For exactly LinearKernel / RBFKernel, D <= 16, dyadic <= 2, either scheme, the answer with SK_ROUTE_NO_STREAM is never
 * SK_ROUTE_STREAM: every such call CAN run with nothing of size pairs x M x N in HBM. */
#define SK_ROUTE_NO_STREAM 1
#define SK_ROUTE_NO_SWAP 2      /* never answer a swapped ADJOINT route (paired batches: the second-argument sums exist for Gram calls) */
/* The COST table: every measured crossover the library (sk_route_query, the launchers) and the host layer (symmetric blocks, merged
 * loss, paired merge) decide by, with the same-box A/B measurement each came from -- entries 0 .. n-1 (sk_cost_name returns NULL past
 * the end).  Scope rules say what a kernel CAN do; these say when it is the faster choice.  tools/crossovers.py re-measures them on
 * the box at hand.  The reference has no counterpart (it has one route per device, sigkernel.py:220-246). */
double sk_cost_query(int which);
const char *sk_cost_name(int which);
const char *sk_cost_note(int which);
#define SK_OP_FORWARD 0
#define SK_OP_ADJOINT 1
#define SK_OP_ADJOINT_SYM 2   /* compute_Gram(X, X, sym=True) with a gradient (sigkernel.py:404-416 on the symmetric call): SK_ROUTE_FUSED =
                                 the TRIANGLE through sk_rbf_adjoint_fused_f64 with the second-argument sums; else all pairs */
#define SK_ROUTE_STREAM 0
#define SK_ROUTE_FUSED 1
#define SK_ROUTE_FUSED_MB 2
#define SK_ROUTE_FUSED_MB_SWAP 3
#define SK_ROUTE_FUSED_SWAP 4
int sk_route_query(int op, int kind, int D, int M, int N, int dyadic, int scheme, int elem_size, int flags);

/* Development hook: the SK_* tuning knobs are parsed from the environment ONCE, when the library is loaded; tools that sweep a
 * knob inside one process call this after changing it.  Not for product code (not thread-safe against concurrent launches). */
void sk_reload_knobs(void);

/* Diagnostics: which kernel instances does a workload launch?  Every launch of the library is counted per kernel instance while
 * tracing is on (enable: 1 on, 0 off, anything else: query; returns the previous state; SK_TRACE_LAUNCHES=1 in the environment
 * switches it on at load).  sk_launch_trace_dump writes "count<TAB>device symbol<NL>" per instance launched since the last reset into
 * buf (NUL-terminated, truncated to n bytes; buf may be NULL) and returns the bytes the whole list needs; reset != 0 clears the counts.
 * tools/reach_sweep.py + tests check the list of instances the build contains against it: no unreachable instance.  No counterpart in
 * the reference (its back-ends are JIT-compiled per call signature, cuda_backend.py:5). */
int sk_launch_trace(int enable);
size_t sk_launch_trace_dump(char *buf, size_t n, int reset);

/* kappa_d = 4^-d / sqrt(12), see sk_solve_fwd_linear_*. */
double sk_linear_prescale(int dyadic);
const char *sk_status_string(int status);
/* Number of HIP devices visible, or a negative sk_status. */
int sk_device_count(void);

/* ---- increments ---------------------------------------------------------------------------
 * inc_c[p][i][j] = G[p][i+1][j+1] + G[p][i][j] - G[p][i+1][j] - G[p][i][j+1]
 * Replaces the 4-corner difference at sigkernel.py:217 (paired) and :363 (Gram); recomputed in
 * backward at :264 and :421.   G [P,M,N] -> inc_c [P,M-1,ld] (columns >= N-1 are zero-filled). */
int sk_increments_f64(const double *G, int64_t P, int M, int N, double *inc_c, int64_t ld, void *stream);
int sk_increments_f32(const float *G, int64_t P, int M, int N, float *inc_c, int64_t ld, void *stream);

/* Static kernel + increments in ONE pass for the reference's two standard static kernels, straight from the paths:
 * replaces static_kernel.Gram_matrix(X, Y) / batch_kernel(X, Y) (static_kernels.py:24,33,56,73) followed by the
 * 4-corner difference (sigkernel.py:216-217, :362-363) -- G_static is never materialised.
 *   kind 0: linear, inc = param^2 <dx_p, dy_q>  (param = 1 reproduces Gram_matrix, which ignores `scale`;
 *           param = scale reproduces batch_kernel).  9 <= D <= 32: on v_mfma_f64_16x16x4_f64, dims summed four at a time --
 *           equal to the D <= 8 form's left-to-right sum to rounding
 *   kind 1: rbf,    G = exp(-|x_p - y_q|^2 / param) (param = sigma), inc = ((G11 + G00) - G10) - G01
 *   X [A,M,D], Y [B,N,D] dense; B > 0: Gram, pair (a,b) at a*B+b; B == 0: paired, pair a = (x_a, y_a), Y [A,N,D].
 *   inc_c [P,M-1,ld] with zero-filled padding columns.  D <= 32 (else SK_ERR_UNSUPPORTED: use the generic path). */
int sk_static_increments_f64(int kind, double param, const double *X, const double *Y, int64_t A, int64_t B, int M, int N,
                             int D, double *inc_c, int64_t ld, void *stream);
int sk_static_increments_f32(int kind, double param, const float *X, const float *Y, int64_t A, int64_t B, int M, int N,
                             int D, float *inc_c, int64_t ld, void *stream);

/* Adjoint of sk_static_increments_*: dL/dX from W = dL/d inc_c and the per-pair upstream gradient, with neither G_static
 * nor dL/dG_static materialised.  Replaces the finite-difference contraction (sigkernel.py:313-341, :472-500) and the
 * `grad_output * grad_points` reduction over the second batch index (:343, :410-416) for these two static kernels.
 *   W [P,M-1,ldw]; scale [P] = upstream gradient per pair (NULL = 1); pairs as in sk_static_increments_*.
 *   kind 1 (rbf):    out = dL/dX [A,M,D].
 *   kind 0 (linear): out = T [A,M-1,D], T[a][p] = sum_b scale_ab sum_q W[a,b,p,q] (y[b,q+1]-y[b,q]) (the caller differences it along
 *                    the path and applies scale^2), for 9 <= D <= 32 and N <= 128 (k_static_linear_adj_tiled: W read once, y_b through
 *                    LDS).  SK_ERR_UNSUPPORTED otherwise: dim <= 8 runs from pre-differenced paths in sk_linear_adjoint_*, wider or
 *                    longer paths as a plain batched GEMM in the caller.  X and param are not read. */
int sk_static_adjoint_f64(int kind, double param, const double *X, const double *Y, const double *W, int64_t ldw,
                          const double *scale, int64_t A, int64_t B, int M, int N, int D, double *out, void *stream);
int sk_static_adjoint_f32(int kind, double param, const float *X, const float *Y, const float *W, int64_t ldw,
                          const float *scale, int64_t A, int64_t B, int M, int N, int D, float *out, void *stream);

/* LinearKernel adjoint from pre-differenced paths (the fast route for path dim <= 8; same contraction as
 * sk_static_adjoint_* with kind 0): dYt [Bn][8][ldy] fp64 = y[q+1]-y[q], dimension-major, zero-padded -- the array
 * sk_solve_fwd_linear_* takes; W [P, Mc, ldw]; scale [P] nullable; out [A, Mc, D] = sum_b scale_ab sum_q W[a,b,p,q] dy[b,q,:].
 * The caller differences `out` along the path (d inc[p,q]/d x[p+1] = +s^2 dy[q], d/d x[p] = -s^2 dy[q]).  B = 0: paired. */
int sk_linear_adjoint_f64(const double *dYt, int64_t ldy, const double *W, int64_t ldw, const double *scale, int64_t A,
                          int64_t B, int Mc, int Nc, int D, double *out, void *stream);
int sk_linear_adjoint_f32(const double *dYt, int64_t ldy, const float *W, int64_t ldw, const float *scale, int64_t A, int64_t B,
                          int Mc, int Nc, int D, float *out, void *stream);

/* Adjoint PDE AND LinearKernel contraction in one kernel (csrc/sk_wave_adj_fused.hip): the reverse sweep forms its increments
 * from the path differences, recomputes K from the terminal edges a forward with edges kept (sk_solve_fwd_linear_edges_f64),
 * and contracts W = d k / d inc with the y differences on the spot -- neither the increments nor W exist in HBM.  Replaces
 * sigkernel.py:438-500 for LinearKernel, i.e. sk_static_increments + sk_solve_adj(EDGES_GIVEN) + sk_linear_adjoint.
 *   dXr [A][Mrows][8] = s^2 (x[p+1]-x[p]) -- the LAYOUT of sk_solve_fwd_linear_*'s array but WITHOUT its kappa = sk_linear_prescale(dyadic)
 *   factor (this kernel forms inc = <dXr, dYt> itself, and so does its rescue: a forward's pre-scaled staging would give gradients
 *   wrong by kappa);  dYt [Bn][8][Ncp] = y[q+1]-y[q], Mrows, Ncp: as sk_solve_fwd_linear_* takes them;
 *   edges: sk_strip_edges_bytes layout;  scale [A*B] nullable;  either scheme.
 *   tpart [tpart_doubles] receives partial sums over b: viewed as [A][ceil(B / *ppg_out)][*rows_out][8] (the B pairs of an x_a are split
 *   into that many chunks, lengths differing by one where the number does not divide B; *ppg_out is the longest), sum over the chunk axis,
 *   then T[a][p][:] = that[a][*rows_out - 1 - p][:] for p < Mc is what sk_linear_adjoint_* returns.  tpart == NULL: only
 *   *ppg_out and *rows_out are set (size query: A * ceil(B / ppg) * rows * 8 doubles).  err [P] zero-initialised: per-pair
 *   self-check residual as for sk_solve_adj_*.  B == 0: paired batch (P = A, Bn = A, one chunk).  fp64, dyadic <= 2, Mc <= 128 (64 at dyadic 2),
 *   path dim <= 8; otherwise SK_ERR_UNSUPPORTED (sk_route_query(SK_OP_ADJOINT, 0, ...) == SK_ROUTE_FUSED says when it applies).
 *   ypart / ycols_out (either non-NULL; Gram only, B > 0): the SECOND-argument sums INSTEAD of tpart (which is not touched) --
 *   ypart viewed as [A][B][*ycols_out][8]: per pair and increment column q < Nc of y_b, sum_p W[a,b,p,q] dXr[a][p][:], WITHOUT the
 *   upstream gradient (`scale` then only marks screened pairs): d k(x_a, y_b) / d (y_b[q+1] - y_b[q]); the caller weights the pairs of
 *   a y_b, adds them and differences along the path.  Columns q >= Nc are padding and hold nothing meaningful.  ypart == NULL with
 *   ycols_out set: size query (A B *ycols_out 8 doubles).  This is what SK_ROUTE_FUSED_SWAP runs on (Y, X) for LinearKernel. */
int sk_linear_adjoint_fused_f64(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp,
                                int dyadic, int scheme, const double *edges, const double *scale, double *tpart,
                                size_t tpart_doubles, double *err, double *ypart, size_t ypart_doubles, int *ppg_out, int *rows_out,
                                int *ycols_out, const double *kfinal, double screen, double tol, void *rescue_ws,
                                size_t rescue_ws_bytes, void *stream);

/* Adjoint PDE AND RBFKernel chain rule in one kernel (csrc/sk_wave_adj_fused_rbf.hip): the reverse sweep evaluates the nodes
 * G = exp(-|x - y|^2 / sigma) itself, forms its increments as their 4-corner differences, recomputes K from the terminal edges a
 * forward with edges kept (sk_solve_fwd_rbf_edges_f64), and pushes W = d k / d inc through the difference and the exponential on
 * the spot -- neither the increments nor W nor G_static exist in HBM.  Replaces sigkernel.py:419-502 + :404-416 for RBFKernel,
 * i.e. sk_static_increments + sk_solve_adj(EDGES_GIVEN) + sk_static_adjoint.
 *   Xr [A][Mrows][8], Yt [Bn][8][Ncp]: the POINT arrays sk_solve_fwd_rbf_* takes;  edges: sk_strip_edges_bytes layout;
 *   scale [P] nullable (upstream gradient per pair).  gpart [gpart_doubles] receives partial sums over b, viewed as
 *   [A][ceil(B / *ppg_out)][*rows_out][*outw_out]: summed over the chunk axis, row r < M holds cs = [..][0] and accd = [..][2 .. 2+D), and
 *   dL/dx_a[r] = (-2 / sigma) (x_a[r] cs - accd).  gpart == NULL: size query only.  err [P] zero-initialised: self-check residual
 *   as for sk_solve_adj_*.  B == 0: paired batch.  fp64, dyadic 1..2 (either scheme, path dim <= 8; dim 5..8 at dyadic 1: one coarse row per lane, M <= 64) and dyadic 0 (default scheme, path
 *   dim <= 8, M <= 128: two coarse rows per lane on the strip kernels' edge layout), one band per pair with
 *   M <= lanes x rows per lane (edges: sk_strip_edges_bytes(P, Mc, Nc, ...) -- with Nc + 1 in place of Nc when Nc is a multiple
 *   of 16: the sweep needs a padding NODE column behind the last unit, as sk_solve_fwd_rbf_edges_f64 keeps them); otherwise SK_ERR_UNSUPPORTED
 *   (sk_route_query(SK_OP_ADJOINT, 1, ...) == SK_ROUTE_FUSED says when the host layer takes it: path dim <= 4).
 *   ypart (nullable; Gram): the SECOND-argument sums of the same sweep, for compute_Gram(X, X, sym=True) with a
 *   gradient (compute_mmd's K_XX, sigkernel.py:190), where only the pairs on and above the diagonal are solved and a pair (a, b)
 *   also owes d1 k(x_b, x_a) = d2 k(x_a, x_b) to row b -- and for SK_ROUTE_FUSED_SWAP.  Viewed as [A*B][*ycols_out][W], W = 6 for
 *   path dim <= 4 and 10 for dim 5..8, node column c < N of pair (a, b) holds S0 = [..][0] and S1 = [..][2 .. 2+D), WITHOUT the
 *   upstream gradient: d k(x_a, y_b) / d y_b[c] = (-2 / sigma) (y_b[c] S0 - S1); the caller weights the pairs and folds them over a.
 *   gpart == NULL with ypart given, and always for path dim 5..8 (dyadic 0 and 1): the sums INSTEAD of gpart, which is not written
 *   then (all a swapped call needs -- fewer instructions per macro-step; the variant of dims 5..8 has registers for one of the two).
 *   ycols_out != NULL in the size query asks for that variant's sizes. */
int sk_rbf_adjoint_fused_f64(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                             int dyadic, int scheme, double sigma, const double *edges, const double *scale, double *gpart,
                             size_t gpart_doubles, double *err, double *ypart, size_t ypart_doubles, int *ppg_out, int *rows_out,
                             int *outw_out, int *ycols_out, const double *kfinal, double screen, double tol, void *rescue_ws,
                             size_t rescue_ws_bytes, void *stream);

/* k, d/dgamma k, d2/dgamma2 k with the static kernel fused in (csrc/sk_wave_deriv_fused.hip): the three increment arrays of k_kgrad
 * (sigkernel.py:526-541) are formed inside the solver from the point arrays of x, x + eps gamma, x + 2 eps gamma and y, in the
 * operand order of sk_static_deriv_increments_* (bit-identical increments), and never exist in HBM.  Replaces
 * sk_static_deriv_increments_* + sk_solve_deriv_* (cuda_backend.py:165-223, sigkernel.py:526-566) for LinearKernel (kind 0) and
 * RBFKernel (kind 1, param = sigma).
 *   X0r, X1r, X2r [A][Mrows][fd], Yt [Bn][fd][Ncp]: fp64 POINT arrays as sk_prep_paths_* builds them (fd = 8; D <= 8;
 *   Mrows >= *mrows of sk_solve_deriv_static_workspace_bytes; Ncp >= 2 NUp, NUp = ceil8((Nc + 2) / 2) >= 64);  B > 0: Gram, B == 0:
 *   paired;  out_* [P].  SK_ERR_UNSUPPORTED (workspace_bytes 0): dyadic > 2, D > 8, second path shorter than 126 points. */
size_t sk_solve_deriv_static_workspace_bytes(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows);
int sk_solve_deriv_static_f64(int kind, double param, const double *X0r, const double *X1r, const double *X2r, const double *Yt, int64_t A,
                              int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D, int fd, int dyadic, int scheme, double eps, double *out_k,
                              double *out_kd, double *out_kdd, void *workspace, size_t workspace_bytes, void *stream);

/* The fused RBF adjoint for LONG or WIDE paths (csrc/sk_wave_adj_fused_mb.hip): any number of bands per pair, path dimensions up to
 * 16 -- BASELINE configs[4]'s shape with a gradient runs in sk_solve_fwd_static_* (edges) + this kernel with nothing of size P*M*N
 * in HBM.  Replaces sigkernel.py:419-502 + :404-416 for RBFKernel there, i.e. sk_static_increments + sk_solve_fwd + sk_solve_adj +
 * sk_static_adjoint.
 *   sk_rbf_adjoint_fused_mb_layout: *mrows = rows of Xr per path, gpart = [P][*rows][*outw] doubles, n0 = [P][*ncols] doubles,
 *   *edge_doubles per pair, *workspace_bytes (one band-boundary row per resident wave); SK_ERR_UNSUPPORTED outside the scope
 *   (dyadic 0..2, D <= 16; any M, N: second paths shorter than ~160 points are swept with masked padding units, *ncols = 2 NUp = Ncp;
 *   at dyadic 0 the kernel gives a lane two coarse rows -- bands of 128 rows -- and sk_solve_fwd_static_* keeps its edges likewise).
 *   Xr [A][Mrows][fd] / Yt [Bn][fd][Ncp]: the fp64 POINT arrays of sk_solve_fwd_static_* (kind 1), fd = 8 or 16; yt_f32 = 1 (fd = 16,
 *   fp32 inputs): Yt is the packed fp32 array sk_solve_fwd_static_f32 takes -- half the LDS ring, two waves per SIMD at 16 dimensions;
 *   edges: what sk_solve_fwd_static_* kept;  scale [P] nullable.  gpart receives, per PAIR and node row 1 <= r < M, cs = [..][0]
 *   and accd = [..][2 .. 2+D): summed over the pairs of an x_a, dL/dx_a[r] = (-2 / sigma) (x_a[r] cs - accd).  Node row 0 (gpart's
 *   row 0 is written by the rescue only: zero it) comes as per-column weights n0[pair][c], c < N: cs += sum_c n0, accd += sum_c n0 y_b[c].
 *   rescue_ws / kfinal / screen / tol: the device-side rescue described below (sk_fused_rescue_workspace_bytes(1, ...)), with yt64 =
 *   the fp64 Yt [Bn][fd][Ncp] (the same array as Yt unless yt_f32).  err [P] zero-initialised:
 *   self-check residual as for sk_solve_adj_* (a pair whose scale is NaN is skipped and its sums stored as zeros). */
int sk_rbf_adjoint_fused_mb_layout(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows, int *rows, int *outw, int *ncols,
                                   int64_t *edge_doubles, size_t *workspace_bytes);
int sk_rbf_adjoint_fused_mb_f64(const double *Xr, const void *Yt, int yt_f32, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D, int fd,
                                int dyadic, int scheme, double sigma, const double *edges, const double *scale, double *gpart,
                                size_t gpart_doubles, double *n0, size_t n0_doubles, double *err, void *workspace, size_t workspace_bytes,
                                const double *yt64, const double *kfinal, double screen, double tol, void *rescue_ws, size_t rescue_ws_bytes,
                                void *stream);

/* The same for LinearKernel on long / wide paths (csrc/sk_wave_adj_fused_mb.hip: k_adj_fused_linear_mb): increments from the path
 * differences, W contracted with the y differences on the spot.  Replaces sigkernel.py:419-502 + :404-416 there, i.e.
 * sk_static_increments + sk_solve_fwd + sk_solve_adj + sk_linear_adjoint.
 *   dXr [A][Mrows][fd] = s^2 (x[p+1]-x[p]), dYt [Bn][fd][Ncp] = y[q+1]-y[q] (sk_solve_fwd_static_*'s kind-0 arrays; Ncp >=
 *   sk_solve_fwd_static_cols(0, Nc) = 2 NUp, NUp = max(80, ceil8((Nc + 1) / 2)));  edges: what sk_solve_fwd_static_* (kind 0, edges, Mrows = *mrows) kept, *edge_doubles per pair;
 *   gpart [P][*rows][*outw = fd]: per PAIR and FLIPPED coarse row (row *rows - 1 - p holds coarse row p); summed over the pairs of an
 *   x_a and flipped back it is the T of sk_linear_adjoint_*: dL/dx[m] = s^2 (T[m-1] - T[m]).  Rescue arguments as above.  dyadic 0..2. */
int sk_linear_adjoint_fused_mb_layout(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows, int *rows, int *outw, int64_t *edge_doubles,
                                      size_t *workspace_bytes);
int sk_linear_adjoint_fused_mb_f64(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D, int fd,
                                   int dyadic, int scheme, const double *edges, const double *scale, double *gpart, size_t gpart_doubles,
                                   double *err, void *workspace, size_t workspace_bytes, const double *kfinal, double screen, double tol,
                                   void *rescue_ws, size_t rescue_ws_bytes, void *stream);

/* Device-side rescue of the two fused adjoints above (csrc/sk_adj_fused_rescue.hip) -- what makes a backward pass free of host
 * synchronisation.  The fused adjoints recompute K backwards from its terminal edges, which loses accuracy like 1e-16 K^2 and is
 * useless for exploding kernels; the reference stores both grids for every pair whatever K's size (sigkernel.py:438-470).
 *   rescue_ws (sk_fused_rescue_workspace_bytes; NULL: no rescue -- the residuals in err are then the caller's to act on), with
 *   kfinal [P] = the forward values K[MM][NN] (nullable): pairs with |kfinal| > screen are taken out of the sweep (their err entry
 *   becomes -1) and their EXACT contribution (stored-grid adjoint + static-kernel chain rule) is added to the partial sums after it;
 *   a chunk in which a pair that was not screened still ends with a residual above `tol` -- or with a NaN residual while its
 *   forward value is finite (an overflow of the recompute, not poisoned inputs) -- is recomputed exactly, all its pairs.
 *   With kfinal the library initialises err itself.  When nothing is screened or fails (the normal case) the rescue reads P
 *   doubles and returns.  `blocks` flagged chunks are processed concurrently (1..1024; the workspace grows with it). */
size_t sk_fused_rescue_workspace_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int blocks);

/* Second-argument adjoint (Gram only): dL/dY from W for the pairs (a, b), b >= b0 -- the counterpart of sk_static_adjoint_*
 * that the reference never needs (it returns no gradient for its second argument, sigkernel.py:343, :412).  It exists for
 * compute_Gram(X, X, sym=True) with a gradient: only the blocks on and above the diagonal are solved, and a pair (a, b)
 * above the diagonal also stands for (b, a), whose first-argument gradient is this pair's second-argument one.
 *   W [A*B, M-1, ldw], scale [A*B] nullable (the caller passes the TRANSPOSED upstream gradient block).
 *   kind 0 (linear, D <= 8): dXr [A][Mrows][8] fp64 = param^2 (x[p+1]-x[p]) zero-padded (the layout of sk_solve_fwd_linear_*'s array,
 *                    WITHOUT its kappa = sk_linear_prescale(dyadic) factor), X, Y unused; out = T2 [B-b0, N-1, D], the caller forms dL/dy[b][n] = T2[b][n-1] - T2[b][n].
 *   kind 1 (rbf):    dXr unused; out = dL/dY [B-b0, N, D]. */
int sk_static_adjoint2_f64(int kind, double param, const double *X, const double *Y, const double *dXr, int Mrows,
                           const double *W, int64_t ldw, const double *scale, int64_t A, int64_t B, int b0, int M, int N, int D,
                           double *out, void *stream);
int sk_static_adjoint2_f32(int kind, double param, const float *X, const float *Y, const double *dXr, int Mrows, const float *W,
                           int64_t ldw, const float *scale, int64_t A, int64_t B, int b0, int M, int N, int D, float *out,
                           void *stream);

/* Transpose of sk_increments_*, used by the adjoint: dG[p][m][n] = s_p * (W[m-1][n-1] + W[m][n]
 * - W[m-1][n] - W[m][n-1]) with out-of-range W = 0 and s_p = scale[p] (or 1 if scale == NULL).
 * Replaces the finite-difference contraction at sigkernel.py:313-341 / :472-500 (the reference
 * differentiates the increments numerically with h = 1e-9; here dL/dG_static is formed
 * exactly and the static kernel is differentiated by the caller).
 * W [P,M-1,ldw] (row stride ldw >= N-1, 0 = dense), scale [P] -> dG [P,M,N]. */
int sk_increments_adjoint_f64(const double *W, int64_t ldw, const double *scale, int64_t P, int M, int N, double *dG,
                              void *stream);
int sk_increments_adjoint_f32(const float *W, int64_t ldw, const float *scale, int64_t P, int M, int N, float *dG,
                              void *stream);

/* ---- path staging for the fused solvers -----------------------------------------------------
 * Builds, in one launch, one of the two fp64 zero-padded arrays the fused kernels below read, from the caller's dense
 * X [A,M,D] of either precision (replaces the tensor arithmetic the reference runs before its solver launch,
 * static_kernels.py:26-33 / :58-73):
 *   diff = 1: scale * (x[p+1] - x[p]), p < M-1 (differences of the up-cast points);  diff = 0: scale * x[p], p < M;
 *   dim_major = 0: out [A][rows][fd] (the `dXr` / `Xr` layout);  dim_major = 1: out [A][fd][rows] (`dYt` / `Yt`, rows = Ncp);
 *   dim_major = 2 (f32 entry point, diff = 0, fd and rows even): per path fd/2 rows of FLOAT [rows/2][4], the 4 floats of a unit =
 *   {dim 2j col 2u, dim 2j col 2u+1, dim 2j+1 col 2u, dim 2j+1 col 2u+1}, then one row of DOUBLE |x_p|^2 [rows]: (fd/2 + 1) * rows * 8
 *   bytes per path (the fp32 ring of sk_solve_fwd_static_f32);
 *   rows >= M-1 (diff) or M; fd >= D; everything outside the valid range is written as zero. */
int sk_prep_paths_f64(const double *X, int64_t A, int M, int D, int diff, int dim_major, double scale, double *out, int rows, int fd,
                      void *stream);
int sk_prep_paths_f32(const float *X, int64_t A, int M, int D, int diff, int dim_major, double scale, double *out, int rows, int fd,
                      void *stream);
/* Both arrays of a call in ONE launch: out_x [A][rows_x][fd] (dim_major = 0) from X scaled by scale_x and out_y [B][fd][rows_y]
 * (dim_major = 1) from Y scaled by scale_y; small calls are launch-bound. */
int sk_prep_pair_f64(const double *X, int64_t A, int M, const double *Y, int64_t B, int N, int D, int diff, double scale_x, double scale_y,
                     double *out_x, int rows_x, double *out_y, int rows_y, int fd, void *stream);
int sk_prep_pair_f32(const float *X, int64_t A, int M, const float *Y, int64_t B, int N, int D, int diff, double scale_x, double scale_y,
                     double *out_x, int rows_x, double *out_y, int rows_y, int fd, void *stream);

/* ---- forward solve ------------------------------------------------------------------------
 * Solves the Goursat PDE for every pair and returns K[MM][NN].
 * Replaces sigkernel_cuda[A,T](...) (sigkernel.py:231; kernel cuda_backend.py:6-49),
 * sigkernel_Gram_cuda[(A,B),T](...) (sigkernel.py:378; cuda_backend.py:121-160) and their CPU
 * twins sigkernel_cython / sigkernel_Gram_cython (sigkernel.py:246, :395;
 * cython_backend.pyx:7-33, :64-119).  Unlike the reference there is no 1024-thread limit
 * (sigkernel.py:222, :368) and no out-of-bounds extra row/column.
 *   inc_c     [P,Mc,ld]  coarse increments, row stride ld
 *   out_final [P]        K[MM][NN]
 *   out_grid  nullable   [P,MM+1,NN+1] full solution grid (what the reference returns)
 *   out_edges nullable   [P,MM+NN+2]: K[MM][0..NN] followed by K[0..MM][NN] -- the terminal
 *                        row and column.  Like out_grid it is served by the anti-diagonal kernel
 *                        (bit-identical to the reference); sk_solve_adj_* obtains the edges it needs
 *                        from the strip kernel directly, in that kernel's own padded layout. */
int sk_solve_fwd_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags,
                     double *out_final, double *out_grid, double *out_edges, void *stream);
int sk_solve_fwd_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags,
                     float *out_final, float *out_grid, double *out_edges, void *stream);

/* Forward solve with the LINEAR static kernel fused in (csrc/sk_wave_fused.hip): the increments
 * s^2 <x[p+1]-x[p], y[q+1]-y[q]> are formed inside the sweep, nothing of size P*M*N ever exists in HBM.
 * Replaces, for LinearKernel, the whole of sigkernel.py:362-382 (Gram) / :216-234 (paired).
 *   dXr [A][Mrows][8] fp64: kappa s^2 (x[p+1]-x[p]) for p < Mc, kappa = sk_linear_prescale(dyadic) = 4^-d / sqrt(12) (the
 *                           kernel's stencil coefficients take three operations per coarse cell on the pre-scaled increment);
 *                           zero for the padding rows (Mrows >= 256 is always enough) and padding dims (path dim <= 8);
 *   dYt [Bn][8][Ncp] fp64: y[q+1]-y[q], dimension-major, zero-padded; Ncp = Nc rounded up to a multiple of 16;
 *   B > 0: Gram (Bn = B, pair (a,b) at a*B+b); B == 0: paired (Bn = A).  out_final [P].
 *   D = the path dimension (1..8): dimensions >= D of dXr / dYt must be zero; D <= 4 selects kernels that skip them.
 *   queue: 64 bytes of device scratch for the launch's work counter (contents irrelevant, the library zeroes it on `stream`;
 *          not shared with a launch that may run concurrently).  The waves of these persistent kernels take a fixed first share
 *          of the pairs and draw the rest from that counter, so that they finish together whatever the SIMD arbitration, the
 *          XCDs' clocks or other work on the chip do.  NULL: a static, equal partition (the results are the same bit for bit).
 * SK_ERR_UNSUPPORTED when dyadic > 2 or a pair needs more than one band (M-1 > 256/128/64 for dyadic 0/1/2):
 * use sk_static_increments_* + sk_solve_fwd_*. */
int sk_solve_fwd_linear_f64(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                            int dyadic, int scheme, double *out_final, void *queue, void *stream);
int sk_solve_fwd_linear_f32(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                            int dyadic, int scheme, float *out_final, void *queue, void *stream);

/* Forward solve with the RBF static kernel fused in (csrc/sk_wave_fused.hip, KIND 1): the nodes
 * G[p][q] = exp(-|x_p - y_q|^2 / sigma) are evaluated inside the sweep (one exp per coarse cell; a lane takes the node row
 * under its last coarse row from the lane below by DPP) and differenced in the reference's order
 * ((G11 + G00) - G10) - G01; neither G_static nor the increments exist in HBM.  Replaces, for RBFKernel,
 * static_kernels.py:58-73 + sigkernel.py:362-382 (Gram) / static_kernels.py:43-56 + sigkernel.py:216-234 (paired).
 * |x - y|^2 is summed directly over the dimensions, not as |x|^2 + |y|^2 - 2<x,y>.
 *   Xr [A][Mrows][8] fp64: the path POINTS x_p, p < M = Mc + 1, zero padding rows / dims (path dim <= 8);
 *   Yt [Bn][8][Ncp] fp64: y_q, q < N = Nc + 1, dimension-major, zero-padded; Ncp = N rounded up to a multiple of 16;
 *   inv_sigma = 1 / sigma;  B > 0: Gram, B == 0: paired;  out_final [P];  D as for sk_solve_fwd_linear_*.
 * SK_ERR_UNSUPPORTED when dyadic > 2 or a pair needs more than one band (M > 256/128/64 for dyadic 0/1/2; at dyadic 0 the
 * four-rows-per-lane kernel is built for D <= 4, the default scheme and fp64 output: D 5..8, SK_SCHEME_NAIVE and sk_solve_fwd_rbf_f32
 * take two rows per lane there, M <= 128, and sk_solve_fwd_rbf_edges_f64 does not cover them): use sk_static_increments_* +
 * sk_solve_fwd_*, or sk_solve_fwd_static_*; sk_route_query says which. */
int sk_solve_fwd_rbf_f64(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                         int dyadic, int scheme, double inv_sigma, double *out_final, void *queue, void *stream);
int sk_solve_fwd_rbf_f32(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                         int dyadic, int scheme, double inv_sigma, float *out_final, void *queue, void *stream);
/* Symmetric Gram matrix of ONE path batch with the fused kernels above: only the A (A + 1) / 2 pairs on and above the diagonal are
 * solved (what the reference's CPU solver does for sym=True, cython_backend.pyx:74-97; its GPU path ignores `sym`), in ONE launch,
 * and each value is written to out[a][b] and out[b][a]: out [A][A] is exactly symmetric.  dXr / dXt (Xr / Xt): the row-major and
 * the dimension-major staging of the SAME paths, as sk_solve_fwd_linear_* (sk_solve_fwd_rbf_*) take them; Mc = Nc = M - 1.
 * pair_tab: the triangle's pairs as [A (A + 1) / 2][2] int32 (a, b), written by sk_prep_cat_* (tri_n = -1) RIGHT BEHIND dXt / Xt
 * (pair_tab == (int *)(Xt + A 8 Ncp)): the kernel looks pairs up there instead of inverting the triangular numbering itself. */
int sk_solve_fwd_linear_sym_f64(const double *dXr, const double *dXt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                                int scheme, double *out, void *queue, void *stream);
int sk_solve_fwd_linear_sym_f32(const double *dXr, const double *dXt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                                int scheme, float *out, void *queue, void *stream);
int sk_solve_fwd_rbf_sym_f64(const double *Xr, const double *Xt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                             int scheme, double inv_sigma, double *out, void *queue, void *stream);
int sk_solve_fwd_rbf_sym_f32(const double *Xr, const double *Xt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                             int scheme, double inv_sigma, float *out, void *queue, void *stream);

/* Forward solve with the static kernel fused in for LONG or WIDE paths (csrc/sk_wave_fused_mb.hip): any number of bands per
 * pair (M - 1 beyond 256/128/64 at dyadic 0/1/2) and path dimensions up to 16 -- BASELINE configs[4] (len 512, dim 16,
 * RBF, dyadic 2) runs in this one kernel with nothing of size P*M*N in HBM.  Replaces, like the two families above,
 * static_kernels.py:26-33 / :58-73 + sigkernel.py:362-382 (Gram) / :216-234 (paired).
 *   kind 0 (linear, param unused): Xr [A][Mrows][fd] = s^2 (x[p+1]-x[p]), Yt [Bn][fd][Ncp] = y[q+1]-y[q];
 *   kind 1 (rbf, param = sigma):   Xr = the points x[p], Yt = the points y[q];  both fp64, zero-padded, as sk_prep_paths_* builds
 *   them;  fd = 8 for D <= 8, 16 for D <= 16;  Mrows >= sk_solve_fwd_static_rows(kind, Mc, dyadic);
 *   Ncp >= sk_solve_fwd_static_cols(kind, Nc) = 2 NUp, NUp = max(80, ceil8((Nc + 1 + kind) / 2)) (second paths shorter than ~160 points
 *   are swept with padding units behind them: wasted steps, same result -- swap the arguments when the first path is the longer
 *   one, the kernel is symmetric);  B > 0: Gram, B == 0: paired;  out_final [P];  either scheme;
 *   workspace: sk_solve_fwd_static_workspace_bytes(...) bytes (one band-boundary row per resident wave; 0 = unsupported).
 * SK_ERR_UNSUPPORTED: dyadic > 2, D > 16. */
size_t sk_solve_fwd_static_workspace_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int D);
/* FEW pairs of LONG paths (fewer pairs than half the resident waves, second paths of ~500 points and more, no edges kept): the
 * BANDS of a pair run on different waves -- band b + 1 trails band b through the pair's boundary row in HBM and a progress
 * counter, work handed out band-major by one ticket counter (an item only waits for a smaller ticket: no deadlock whatever is
 * resident) -- instead of one wave sweeping the bands of its pair one after the other; the same arithmetic in the same order, bit
 * for bit.  The workspace above includes its rows.  Returns the bands per pair when a launch of P pairs takes that mode, else 0
 * (cython_backend.pyx:64-119 has no length limit; its one thread per pair is what this replaces).  SK_FUSEDMB_SPLIT=0 disables.
 * Status: a wave's wait for the band above is bounded (~2^22 polls: a stall -- queue preemption, a fault in another wave -- must
 * not hang the device); an item that gives up poisons ITS pair with NaN (the wave's other items are untouched) and counts in the
 * launch's status word, the LAST 8 bytes (uint64) of the workspace as passed: 0 after every normal launch. */
int sk_solve_fwd_static_split(int kind, int64_t P, int Mc, int Nc, int dyadic, int D);
int sk_solve_fwd_static_rows(int kind, int Mc, int dyadic);
/* Ncp = 2 NUp: columns per dimension row of Yt (zero-padded), the same for sk_rbf_adjoint_fused_mb_f64 / sk_linear_adjoint_fused_mb_f64. */
int sk_solve_fwd_static_cols(int kind, int Nc);
/* edges (nullable; dyadic 0..2, either kind): also keep, of the grid PADDED to the bands and units of sk_rbf_adjoint_fused_mb_f64 / sk_linear_adjoint_fused_mb_f64 (padding
 * carries no increments), the bottom row of every band of 64 lanes (the last one is the pair's terminal row) and the terminal column
 * -- *edge_doubles (sk_rbf_adjoint_fused_mb_layout) doubles per pair: what that adjoint recomputes K from, band by band.  Mrows must
 * then be the layout's *mrows. */
int sk_solve_fwd_static_f64(int kind, double param, const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc,
                            int Ncp, int D, int fd, int dyadic, int scheme, double *out_final, double *edges, void *workspace,
                            size_t workspace_bytes, void *stream);
/* f32: yt_f32 = 1 (kind 1, fd = 16 only): Yt holds the fp32 points packed as sk_prep_paths_f32 layout 2 writes them -- half
 * the LDS ring, twice the resident waves at 16 dimensions; arithmetic stays fp64 (dyadic >= 1).  yt_f32 = 0: Yt is the fp64 array above --
 * except for kind 1, fd = 16, dyadic >= 1, which this entry point serves in the packed form only (SK_ERR_UNSUPPORTED otherwise). */
int sk_solve_fwd_static_f32(int kind, double param, const double *Xr, const void *Yt, int yt_f32, int64_t A, int64_t B, int Mrows,
                            int Mc, int Nc, int Ncp, int D, int fd, int dyadic, int scheme, float *out_final, double *edges,
                            void *workspace, size_t workspace_bytes, void *stream);

/* The same, also keeping the terminal row/column of every pair (layout and size: sk_strip_edges_bytes) for a later
 * sk_solve_adj_* with SK_FLAG_EDGES_GIVEN on the increments of the same paths (sk_static_increments_*, kind 1). */
int sk_solve_fwd_rbf_edges_f64(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                               int dyadic, int scheme, double inv_sigma, double *out_final, double *edges, void *queue, void *stream);

/* The same, also keeping the terminal row/column of every pair for a later sk_solve_adj_* with SK_FLAG_EDGES_GIVEN
 * (`edges`: sk_strip_edges_bytes(P, Mc, Nc, dyadic, 8) bytes; fp64, dyadic 0..2). */
int sk_solve_fwd_linear_edges_f64(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                                  int dyadic, int scheme, double *out_final, double *edges, void *queue, void *stream);

/* ---- adjoint solve ------------------------------------------------------------------------
 * W[p][a][b] = d K_p[MM][NN] / d inc_c[p][a][b] by the reference's variation-of-parameters
 * formula:  W = 4^-d * sum over the fine cells (i,j) of coarse cell (a,b) of
 * K[i][j] * Krev[MM-1-i][NN-1-j], Krev = solution on the doubly flipped increments.
 * Replaces the second solver launch and the KK product at sigkernel.py:282-311 (_SigKernel.backward)
 * and :438-470 (prep_backward).
 * Two implementations behind one entry point:
 *   fast   -- forward sweep emitting the terminal row/column of K, then ONE fused sweep that runs the reverse
 *             PDE and recomputes K backwards from those edges (no grid is stored; csrc/sk_wave_adj.hip).
 *             Needs dyadic 0..2 (fp32: 0..1), increment rows zero-padded to whole 128-byte lines
 *             (ld*sizeof(T) % 128 == 0, as sk_increments_* produces when given such an ld), ldw >= that padded
 *             width, and out_err != NULL; any grid size (no 1024-node limit).  out_err[p] receives the self-check residual
 *             max_i |K_recomputed[i][0] - 1| of pair p: the caller re-solves pairs whose residual is too
 *             large with SK_FLAG_SIMPLE (only ever seen when K explodes, |K| >~ 1e4).
 *   simple -- both grids stored in `workspace`, products summed in the reference's order: bit-identical to the
 *             reference formula evaluated by the CPU oracle (SK_FLAG_EXACT / SK_FLAG_SIMPLE, or any shape
 *             the fast path does not cover).  out_err is zero-filled.
 *   workspace: sk_adj_workspace_bytes(...) bytes of device scratch.
 *   inc_c [P,Mc,ld]; out_final nullable [P]; W [P,Mc,ldw] (ldw = 0: dense); out_err nullable [P] doubles. */
size_t sk_adj_workspace_bytes(int64_t P, int Mc, int Nc, int dyadic, int flags, int elem_size);
/*   flags = SK_FLAG_FAST_ONLY: what the fast implementation alone needs (P (edges + 1) doubles; 0 when it does not cover
 *   the shape) -- the stored-grid figure grows with min(P, 1024) whole grids and is only needed for SK_FLAG_SIMPLE /
 *   SK_FLAG_EXACT calls or shapes the fast kernels do not cover. */

/* Device-side rescue of the fast adjoint: re-solves with stored grids exactly the pairs whose self-check residual
 * err[p] (as written by sk_solve_adj_*) exceeds `tol`, overwriting their W (and out_final when non-NULL); a NaN residual is
 * left alone when the pair's increments are themselves not finite (NaN / inf coordinates: the re-solve could only reproduce the
 * NaN) and re-solved when they are finite (the backward recompute overflowed).
 * Enqueue it unconditionally right after sk_solve_adj_*: when no pair is flagged it reads P doubles and returns, so the
 * caller never has to read the residuals back (the reference has no such step: it stores both grids for every pair,
 * sigkernel.py:438-470).  workspace: k >= 1 slots of sk_adj_rescue_slot_bytes(Mc, Nc, dyadic) bytes; k slots re-solve k
 * flagged pairs concurrently.  sk_adj_rescue_slot_bytes is 0, and sk_adj_rescue_* returns SK_ERR_UNSUPPORTED, for grids
 * whose three live diagonals exceed the LDS (more than ~6800 fine rows): no rescue exists there. */
size_t sk_adj_rescue_slot_bytes(int Mc, int Nc, int dyadic);
int sk_adj_rescue_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, const double *err,
                      double tol, double *out_final, double *W, int64_t ldw, void *workspace, size_t workspace_bytes, void *stream);
int sk_adj_rescue_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, const double *err,
                      double tol, float *out_final, float *W, int64_t ldw, void *workspace, size_t workspace_bytes, void *stream);

/* Forward solve that also keeps what the fast adjoint needs -- the terminal row and column of K in the strip kernels'
 * padded layout -- so that a later sk_solve_adj_* with SK_FLAG_EDGES_GIVEN (workspace = `edges`) skips its forward sweep.
 * The reference keeps the whole solution grid between forward and backward (sigkernel.py:248, :397-399); this is
 * (MM+NN)/(MM*NN) of that.  sk_strip_edges_bytes: size of `edges` for P pairs, 0 when the strip kernels do not cover the
 * shape (then sk_solve_fwd_edges_* returns SK_ERR_UNSUPPORTED: use sk_solve_fwd_* and a plain sk_solve_adj_*).
 * Requires the layout of the fast adjoint (rows zero-padded to whole 128-byte lines). */
size_t sk_strip_edges_bytes(int64_t P, int Mc, int Nc, int dyadic, int elem_size);
int sk_solve_fwd_edges_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, double *out_final,
                           double *edges, void *stream);
int sk_solve_fwd_edges_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, float *out_final,
                           double *edges, void *stream);
int sk_solve_adj_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags,
                     double *out_final, double *W, int64_t ldw, double *out_err, void *workspace,
                     size_t workspace_bytes, void *stream);
int sk_solve_adj_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags,
                     float *out_final, float *W, int64_t ldw, double *out_err, void *workspace,
                     size_t workspace_bytes, void *stream);

/* ---- directional derivatives --------------------------------------------------------
 * Replaces: sigkernel_derivatives_Gram_cuda[(A,B),T](M_inc, M_inc_diff, M_inc_diffdiff, MM+1, NN+1, n_anti_diagonals,
 *           M_sol, M_sol_diff, M_sol_diffdiff)  (cuda_backend.py:166-223, launched at sigkernel.py:546-566) and the
 *           three tile() refinements before it (sigkernel.py:543-545).
 * inc, inc_d, inc_dd: [P, Mc, ld] COARSE increments of the static kernel, of its first and of its second
 * finite-difference derivative along gamma (what sigkernel.py:526-541 builds before refinement); all three share
 * `ld` (0 = Nc).  out_k / out_kd / out_kdd: [P] = K[MM][NN], K_gamma[MM][NN], K_gamma_gamma[MM][NN] (each
 * nullable, at least one non-null).  The scheme is the reference's (no _naive_solver variant exists for this
 * path).  flags: SK_FLAG_EXACT / SK_FLAG_SIMPLE = anti-diagonal kernel in the reference's operand order
 * (bit-identical to the CPU oracle); SK_FLAG_FAST_ONLY = fail with SK_ERR_UNSUPPORTED instead of falling back. */
/* Replaces: the finite-difference pre-processing of k_kgrad (sigkernel.py:526-541).  G0, G1, G2: [P, M, N] static Gram
 * matrices of (X, Y), (X + eps*gamma, Y), (X + 2*eps*gamma, Y); inc, inc_d, inc_dd: [P, M-1, ld] (ld 0 = N-1, padding
 * columns are zeroed) -- the inputs of sk_solve_deriv_*.  Operand order as in the reference (the sums cancel 1/eps^2). */
int sk_deriv_increments_f64(const double *G0, const double *G1, const double *G2, double eps, int64_t P, int M, int N,
                            double *inc, double *inc_d, double *inc_dd, int64_t ld, void *stream);
int sk_deriv_increments_f32(const float *G0, const float *G1, const float *G2, double eps, int64_t P, int M, int N,
                            float *inc, float *inc_d, float *inc_dd, int64_t ld, void *stream);
/* The same with the static kernel fused in (kind 0 = linear, 1 = rbf with param = sigma, like sk_static_increments_*):
 * replaces the three Gram_matrix calls of k_kgrad (sigkernel.py:526, :530, :537) as well.  X0 = X, X1 = X + eps*gamma,
 * X2 = X + 2*eps*gamma: [A, M, D] each (formed by the caller); Y [B, N, D]; outputs [A*B, M-1, ld].  D <= 32. */
int sk_static_deriv_increments_f64(int kind, double param, const double *X0, const double *X1, const double *X2,
                                   const double *Y, int64_t A, int64_t B, int M, int N, int D, double eps, double *inc,
                                   double *inc_d, double *inc_dd, int64_t ld, void *stream);
int sk_static_deriv_increments_f32(int kind, double param, const float *X0, const float *X1, const float *X2, const float *Y,
                                   int64_t A, int64_t B, int M, int N, int D, double eps, float *inc, float *inc_d,
                                   float *inc_dd, int64_t ld, void *stream);
int sk_solve_deriv_f64(const double *inc, const double *inc_d, const double *inc_dd, int64_t ld, int64_t P, int Mc, int Nc,
                       int dyadic, int flags, double *out_k, double *out_kd, double *out_kdd, void *stream);
int sk_solve_deriv_f32(const float *inc, const float *inc_d, const float *inc_dd, int64_t ld, int64_t P, int Mc, int Nc,
                       int dyadic, int flags, float *out_k, float *out_kd, float *out_kdd, void *stream);

/* ---- one-launch glue of the loss wrappers (csrc/sk_loss.hip) -----------------------------------------------------
 * compute_mmd / compute_scoring_rule / compute_expected_scoring_rule (sigkernel.py:146-197) assemble their value from two or
 * three compute_Gram calls and some twenty elementwise / reduction kernels; on MI355X every dispatch costs 4-5 us, a third of a
 * training-sized step.  These entry points make each stage ONE launch; all are deterministic (no atomics, fixed reduction order).
 *
 * sk_prep_cat_*: Z = [X; Y] (X [A,M,D], Y [B,M,D], same length) staged in BOTH layouts from the two batches -- no concatenated
 *   copy: out_rows [A+B][rows][fd] (scaled by scale_rows), optionally out_rows2 (the same scaled by scale_rows2: the linear
 *   adjoint's rows carry s^2, the forward's kappa s^2), out_cols [A+B][fd][cols] (unscaled); diff as sk_prep_paths_*.  pair_tab
 *   (nullable): the pairs of a triangular layout as [P][2] int32 (a, b), written by the same launch -- tri_n >= 0: the loss layout
 *   below; tri_n = -1: the inclusive upper triangle of all A + B paths (sk_solve_fwd_*_sym_*); sk_solve_fwd_loss_f64 wants it
 *   RIGHT BEHIND out_cols (pair_tab == (int *)(out_cols + (A+B) fd cols), fd = 8): its kernel finds the table from the columns' address.
 * sk_solve_fwd_loss_f64: ONE fused forward launch over the LOSS LAYOUT of pairs: the rectangle K(Z[0..A), Z) -- A (A+B) pairs,
 *   pair (a, b) at a (A+B) + b, whose column blocks are K(X, X) and K(X, Y) -- followed by the STRICT upper triangle (i < j,
 *   row-major) of K(Y, Y), tri_n (tri_n - 1) / 2 pairs (tri_n = B, or 0: no K_YY term; the diagonals never enter the unbiased
 *   statistics, sigkernel.py:194-197).  out [P] in pair order; edges (nullable) receives the terminal edges of the RECTANGLE pairs
 *   only, in the layout sk_solve_fwd_{linear,rbf}_edges_f64 writes (sk_strip_edges_bytes for A (A+B) pairs) -- the triangle
 *   carries no gradient (the second argument of compute_mmd must not require one, sigkernel.py:188).  Replaces the three
 *   _SigKernelGram.forward calls of sigkernel.py:190-192.  kind 0: Zr = kappa s^2 differences (sk_linear_prescale), Zt differences;
 *   kind 1: points, param = 1 / sigma.  SK_ERR_UNSUPPORTED outside the one-band kernels' scope.
 * sk_loss_value_f64: value[0] = sum_{a != b} K_XX[a,b] / (A (A-1)) - 2 mean(K_XY) [+ sum_{i != j} K_YY[i,j] / (B (B-1)) when
 *   with_yy] from that output (sigkernel.py:194-197, :160-161, :177-178); wb (nullable, [A (A+B)]): d value / dK of the rectangle
 *   with the K_XX block doubled (the reference's backward doubles a Gram whose both arguments require a gradient,
 *   sigkernel.py:410-412), written by the same launch -- the constant upstream weights of the adjoint.
 * sk_loss_weights_f64: go [A (A+B)] = grad_out[0] * those weights; grad_out: a DEVICE scalar (nullable = 1).  (The wrappers
 *   pass the scalar to sk_*_adjoint_finish_f64 instead -- the adjoint is linear in it -- and keep this for callers that need go.)
 * sk_rbf_adjoint_finish_f64 / sk_linear_adjoint_finish_f64: the partial sums of sk_rbf_adjoint_fused_f64 (gpart
 *   [A][chunks][rows][outw]) / sk_linear_adjoint_fused_f64 (tpart [A][chunks][rows][8]) -> dL/dX [A,M,D], chunks added in ascending
 *   order; X: the fp64 points (rbf); scale2 = s^4 / s^2-staging factor of the linear rows (the wrapper's `param ** 2`);
 *   gscale: a DEVICE scalar (nullable = 1) multiplied into the result. */
int sk_prep_cat_f64(const double *X, int64_t A, const double *Y, int64_t B, int M, int D, int diff, double scale_rows, double scale_rows2,
                    double *out_rows, double *out_rows2, int rows, double *out_cols, int cols, int fd, int *pair_tab, int64_t tri_n, void *stream);
int sk_prep_cat_f32(const float *X, int64_t A, const float *Y, int64_t B, int M, int D, int diff, double scale_rows, double scale_rows2,
                    double *out_rows, double *out_rows2, int rows, double *out_cols, int cols, int fd, int *pair_tab, int64_t tri_n, void *stream);
int sk_solve_fwd_loss_f64(int kind, double param, const double *Zr, const double *Zt, const int *pair_tab, int64_t A, int64_t B, int64_t tri_n,
                          int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic, int scheme, double *out, double *edges, void *queue, void *stream);
int sk_loss_value_f64(const double *out, int64_t A, int64_t B, int with_yy, double *value, double *wb, void *stream);
int sk_loss_weights_f64(int64_t A, int64_t B, const double *grad_out, double *go, void *stream);
int sk_rbf_adjoint_finish_f64(const double *gpart, int64_t A, int64_t chunks, int rows, int outw, const double *X, int M, int D, double sigma,
                              const double *gscale, double *grad, void *stream);
int sk_linear_adjoint_finish_f64(const double *tpart, int64_t A, int64_t chunks, int rows, int M, int D, double scale2, const double *gscale,
                                 double *grad, void *stream);

/* ---- launch planning, host only (no device work; exposed so that the partition of the pairs can be tested without a GPU) ----
 * The persistent kernels share P pairs among `waves` waves of G lane groups each; when a launch fills the chip with whole
 * workgroups (waves == resident = n_cu * waves per CU) the shares depend on the wave's age rank.
 * sk_plan_wave_shares: for every wave w, first[w] = the first pair of its lane group 0, ppg[w] = pairs per lane group (group g
 * sweeps first + g * ppg ...), end[w] = the end of its rank's range (pairs >= end belong to other waves).  Returns the number
 * of ranks (1 = equal shares) or a negative sk_status.
 * sk_plan_group_chunks: the fused adjoints' split of an A x B Gram into chunks of one row a (PPG = the equal chunk size,
 * max_groups = resident lane groups): for every lane group gi < n_groups its first pair, its slot in the partial-sum array
 * (a * chunks_per_a + c) and its number of pairs.  Returns the number of ranks. */
int sk_plan_wave_shares(int64_t P, int G, int64_t waves, int64_t resident, int wpb, int n_cu, int64_t *first, int64_t *end, int *ppg);
int sk_plan_group_chunks(int64_t A, int64_t B, int64_t PPG, int64_t max_groups, int G, int wpb, int n_cu, int64_t n_groups,
                         int64_t *first, int64_t *slot, int *ppg);

#ifdef __cplusplus
}
#endif
#endif /* SIGKERNEL_AMD_H */
