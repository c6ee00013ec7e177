// sk_abi.hip -- the extern "C" surface declared in include/sigkernel_amd.h.
// Argument checking and kernel selection only; no torch types, no allocation, no synchronisation.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "sk_internal.h"
#include "sk_wave_common.h"

using namespace sk;

namespace {

bool bad_common(const void *inc, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags) {
    if (!inc || P < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return true;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return true;
    if (flags & ~(SK_FLAG_EXACT | SK_FLAG_SIMPLE | SK_FLAG_FAST_ONLY | SK_FLAG_EDGES_GIVEN)) return true;
    if ((flags & SK_FLAG_EDGES_GIVEN) && (flags & (SK_FLAG_EXACT | SK_FLAG_SIMPLE))) return true;
    if ((flags & SK_FLAG_FAST_ONLY) && (flags & (SK_FLAG_EXACT | SK_FLAG_SIMPLE))) return true;
    if (((int64_t)Mc << dyadic) > (1 << 24) || ((int64_t)Nc << dyadic) > (1 << 24)) return true;
    return false;
}

template <typename T>
int solve_fwd(const T *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, T *out_final,
              T *out_grid, double *out_edges, void *stream) {
    if (bad_common(inc_c, P, Mc, Nc, dyadic, scheme, flags) || (ld != 0 && ld < Nc)) return SK_ERR_BAD_ARG;
    if (!out_final && !out_grid && !out_edges) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    const Geom g = make_geom(P, Mc, Nc, dyadic, scheme, ld);
    // full grids and terminal edges come from the anti-diagonal kernel (the strip kernel's edges use its own padded layout)
    if (!(flags & (SK_FLAG_EXACT | SK_FLAG_SIMPLE)) && out_final && !out_grid && !out_edges) {
        const int rc = launch_fwd_wave<T>(inc_c, g.ld, g, out_final, nullptr, (hipStream_t)stream);
        if (rc != SK_ERR_UNSUPPORTED || (flags & SK_FLAG_FAST_ONLY)) return rc;
    } else if (flags & SK_FLAG_FAST_ONLY) {
        return SK_ERR_UNSUPPORTED;
    }
    return launch_fwd_simple<T>(inc_c, g, out_final, out_grid, out_edges, (hipStream_t)stream);
}

// workspace of the fast adjoint: terminal edges in the strip layout [P, NNp + MMp] doubles + a dummy K[MM][NN] vector
// [P] doubles; 0 when the strip kernels do not cover the shape
size_t adj_fast_workspace_bytes(const Geom &g, int elem_size) {
    const size_t e = strip_edge_doubles(g, elem_size);
    return e ? (size_t)g.P * (e + 1) * sizeof(double) : 0;
}

template <typename T>
int solve_adj(const T *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, T *out_final, T *W,
              int64_t ldw, double *out_err, void *ws, size_t ws_bytes, void *stream) {
    if (bad_common(inc_c, P, Mc, Nc, dyadic, scheme, flags) || !W || (ld != 0 && ld < Nc) || (ldw != 0 && ldw < Nc))
        return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    const Geom g = make_geom(P, Mc, Nc, dyadic, scheme, ld);
    if (ldw == 0) ldw = Nc;
    hipStream_t s = (hipStream_t)stream;
    const bool fast_shape = dyadic >= 0 && dyadic <= (sizeof(T) == 8 ? 2 : 1);   // launch_adj_wave's scope
    const size_t fast_ws = fast_shape ? adj_fast_workspace_bytes(g, (int)sizeof(T)) : 0;
    if (flags & SK_FLAG_EDGES_GIVEN) {   // argument checks before any HIP call
        const size_t need = fast_shape ? (size_t)P * strip_edge_doubles(g, (int)sizeof(T)) * sizeof(double) : 0;
        if (!need || !out_err || !ws || ws_bytes < need) return SK_ERR_WORKSPACE;
    }
    if (out_err && hipMemsetAsync(out_err, 0, sizeof(double) * (size_t)P, s) != hipSuccess) return SK_ERR_LAUNCH;
    if (flags & SK_FLAG_EDGES_GIVEN)
        // the caller kept the strip edges of its forward pass (sk_solve_fwd_edges_*): only the fused reverse sweep runs
        return launch_adj_wave<T>(inc_c, g.ld, g, static_cast<const double *>(ws), W, ldw, out_err, s);
    if (!(flags & (SK_FLAG_EXACT | SK_FLAG_SIMPLE)) && fast_ws && out_err && ws && ws_bytes >= fast_ws) {
        // forward sweep that also emits the terminal row/column, then the fused reverse sweep + recompute of K
        double *edges = static_cast<double *>(ws);
        T *kfin = out_final ? out_final : reinterpret_cast<T *>(edges + (size_t)P * strip_edge_doubles(g, (int)sizeof(T)));
        int rc = launch_fwd_wave<T>(inc_c, g.ld, g, kfin, edges, s);
        if (rc == SK_OK) rc = launch_adj_wave<T>(inc_c, g.ld, g, edges, W, ldw, out_err, s);
        if (rc != SK_ERR_UNSUPPORTED || (flags & SK_FLAG_FAST_ONLY)) return rc;
    } else if (flags & SK_FLAG_FAST_ONLY) {
        return SK_ERR_UNSUPPORTED;
    }
    return launch_adj_simple<T>(inc_c, g, out_final, W, ldw, ws, ws_bytes, s);
}

template <typename T>
int solve_deriv(const T *inc, const T *inc_d, const T *inc_dd, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int flags,
                T *out_k, T *out_kd, T *out_kdd, void *stream) {
    if (bad_common(inc, P, Mc, Nc, dyadic, SK_SCHEME_DEFAULT, flags) || !inc_d || !inc_dd || (ld != 0 && ld < Nc))
        return SK_ERR_BAD_ARG;
    if (!out_k && !out_kd && !out_kdd) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    const Geom g = make_geom(P, Mc, Nc, dyadic, SK_SCHEME_DEFAULT, ld);
    if (!(flags & (SK_FLAG_EXACT | SK_FLAG_SIMPLE))) {
        const int rc = launch_deriv_wave<T>(inc, inc_d, inc_dd, g.ld, g, out_k, out_kd, out_kdd, (hipStream_t)stream);
        if (rc != SK_ERR_UNSUPPORTED || (flags & SK_FLAG_FAST_ONLY)) return rc;
    }
    return launch_deriv_simple<T>(inc, inc_d, inc_dd, g, out_k, out_kd, out_kdd, (hipStream_t)stream);
}

template <typename T>
int solve_fwd_edges(const T *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, T *out_final, double *edges,
                    void *stream) {
    if (bad_common(inc_c, P, Mc, Nc, dyadic, scheme, 0) || !out_final || !edges || (ld != 0 && ld < Nc)) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    const Geom g = make_geom(P, Mc, Nc, dyadic, scheme, ld);
    const bool fast_shape = dyadic >= 0 && dyadic <= (sizeof(T) == 8 ? 2 : 1);   // what launch_adj_wave will accept later
    if (!fast_shape || !strip_edge_doubles(g, (int)sizeof(T)) || (g.ld * sizeof(T)) % 128) return SK_ERR_UNSUPPORTED;
    return launch_fwd_wave<T>(inc_c, g.ld, g, out_final, edges, (hipStream_t)stream);
}

template <typename TO>
int solve_fwd_static(int kind, double param, const double *Xr, const void *Yt, int yt_f32, int64_t A, int64_t B, int Mrows, int Mc,
                     int Nc, int Ncp, int D, int fd, int dyadic, int scheme, TO *out_final, double *edges, void *ws, size_t ws_bytes,
                     void *stream) {
    if (D < 1 || !Xr || !Yt || !out_final || A < 0 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if ((kind != 0 && kind != 1) || (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE)) return SK_ERR_BAD_ARG;
    if (kind == 1 && (!(param > 0.0) || !(param < 1e300))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    return launch_fwd_fused_mb<TO>(kind, Xr, Yt, yt_f32, A, B, Mrows, Ncp, D, fd, g, kind == 1 ? 1.0 / param : 0.0, out_final, edges, ws,
                                   ws_bytes, (hipStream_t)stream);
}

// symmetric Gram of ONE path batch: only the A (A + 1) / 2 pairs on and above the diagonal are solved, each written twice
template <typename TO>
int solve_fwd_sym(int kind, const double *Xr, const double *Xt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                  int scheme, double inv_sigma, TO *out, void *queue, void *stream) {
    if (D < 1 || !Xr || !Xt || !out || A < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    // the pair table of sk_prep_cat_* (tri_n = -1) lies right behind the staged columns: the kernel finds it from Xt
    if (reinterpret_cast<const void *>(pair_tab) != reinterpret_cast<const void *>(Xt + A * (int64_t)8 * Ncp)) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (kind == 1 && (!(inv_sigma > 0.0) || !(inv_sigma < 1e300))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(A * (A + 1) / 2, Mc, Nc, dyadic, scheme);
    if (kind == 0) return launch_fwd_fused_linear<TO>(Xr, Xt, A, A, Mrows, Ncp, D, g, out, nullptr, queue, (hipStream_t)stream, 1);
    return launch_fwd_fused_rbf<TO>(Xr, Xt, A, A, Mrows, Ncp, D, g, inv_sigma, out, nullptr, queue, (hipStream_t)stream, 1);
}

}  // namespace

// ---- the one place that reads the environment: SK_* tuning knobs, parsed when the library is loaded ------------------------
namespace sk {
namespace {
int knob_int(const char *name) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : 0;
}
RankW knob_shares(const char *name) {
    RankW r{};
    const char *e = getenv(name);
    if (e && *e)
        for (const char *q = e; *q && r.n < 4; ++r.n) {
            r.w[r.n] = atof(q);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    return r;
}
Knobs parse_knobs() {
    Knobs k{};
    k.wave_wpc = knob_int("SK_WAVE_WPC"); k.wave_wpb = knob_int("SK_WAVE_WPB");
    k.adj_wpc = knob_int("SK_ADJ_WPC"); k.adj_wpb = knob_int("SK_ADJ_WPB");
    k.adjf_wpc = knob_int("SK_ADJF_WPC"); k.adjf_wpb = knob_int("SK_ADJF_WPB");
    k.adjr_wpc = knob_int("SK_ADJR_WPC"); k.adjr_wpb = knob_int("SK_ADJR_WPB");
    k.adjmb_wpc = knob_int("SK_ADJMB_WPC"); k.adjmb_wpb = knob_int("SK_ADJMB_WPB"); k.adjmb_q_static = knob_int("SK_ADJMB_Q_STATIC");
    k.derivf_wpc = knob_int("SK_DERIVF_WPC"); k.derivf_wpb = knob_int("SK_DERIVF_WPB"); k.derivf_noshift = knob_int("SK_DERIVF_NOSHIFT");
    k.deriv_wpc = knob_int("SK_DERIV_WPC"); k.deriv_wpb = knob_int("SK_DERIV_WPB");
    k.fused_wpc = knob_int("SK_FUSED_WPC"); k.fused_wpb = knob_int("SK_FUSED_WPB"); k.fused_q_static = knob_int("SK_FUSED_Q_STATIC");
    k.fused_mid = getenv("SK_FUSED_MID") ? knob_int("SK_FUSED_MID") : 1;
    k.fusedmb_wpc = knob_int("SK_FUSEDMB_WPC"); k.fusedmb_wpb = knob_int("SK_FUSEDMB_WPB"); k.fusedmb_q_static = knob_int("SK_FUSEDMB_Q_STATIC");
    k.fusedmb_split = getenv("SK_FUSEDMB_SPLIT") ? knob_int("SK_FUSEDMB_SPLIT") : 1;
    k.fusedmb_lead = knob_int("SK_FUSEDMB_LEAD");
    k.rank_w = knob_shares("SK_RANK_W"); k.wave_rank_w = knob_shares("SK_WAVE_RANK_W"); k.adj_rank_w = knob_shares("SK_ADJ_RANK_W");
    k.adjf_rank_w = knob_shares("SK_ADJF_RANK_W"); k.adjr_rank_w = knob_shares("SK_ADJR_RANK_W");
    k.deriv_rank_w = knob_shares("SK_DERIV_RANK_W"); k.fused_rank_w = knob_shares("SK_FUSED_RANK_W");
    k.fusedmb_rank_w = knob_shares("SK_FUSEDMB_RANK_W");
    return k;
}
Knobs g_knobs = parse_knobs();          // dynamic initialisation = library load
std::atomic<int> g_cu_count[16];        // 0: not asked yet
// launch trace (sk_internal.h: SK_LAUNCH): counts per host stub, guarded by a mutex (diagnostics: the cost only exists while tracing)
std::atomic<int> g_trace{knob_int("SK_TRACE_LAUNCHES") != 0 ? 1 : 0};
std::mutex g_trace_mu;
std::unordered_map<const void *, unsigned long long> g_trace_counts;
}  // namespace
const Knobs &knobs() { return g_knobs; }
bool trace_on() { return g_trace.load(std::memory_order_relaxed) != 0; }
void trace_launch(const void *host_stub) {
    std::lock_guard<std::mutex> lock(g_trace_mu);
    g_trace_counts[host_stub] += 1;
}
int device_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int n = g_cu_count[dev].load(std::memory_order_relaxed);
    if (n == 0) {       // (racing threads compute the same value)
        int v = 0;
        n = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
        g_cu_count[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
}  // namespace sk

extern "C" {

// 330 (round 6): sk_linear_adjoint_fused_f64 takes ypart / ypart_doubles / ycols_out (the second-argument sums, route FUSED_SWAP)
// 320 (round 5/6): sk_solve_fwd_{linear,rbf}_sym_* take the pair table (argument 3), sk_prep_cat_*, sk_solve_fwd_loss_f64, sk_loss_*,
// sk_*_adjoint_finish_f64, sk_cost_query; the SK_WAVE_PF / SK_DERIV_PF / SK_ADJR_ALL knobs are gone; split mode's status word
// (310: edges argument of sk_solve_fwd_static_*, the multi-band adjoints, the fused derivative solver)
int sk_version(void) { return 330; }

int sk_launch_trace(int enable) {
    const int prev = sk::g_trace.load(std::memory_order_relaxed);
    if (enable == 0 || enable == 1) sk::g_trace.store(enable, std::memory_order_relaxed);
    return prev;
}

size_t sk_launch_trace_dump(char *buf, size_t n, int reset) {
    std::string out;
    {
        std::lock_guard<std::mutex> lock(sk::g_trace_mu);
        std::vector<std::pair<std::string, unsigned long long>> rows;
        for (const auto &kv : sk::g_trace_counts) {
            const char *name = hipKernelNameRefByPtr(kv.first, nullptr);
            rows.emplace_back(name ? name : "?", kv.second);
        }
        std::sort(rows.begin(), rows.end());
        for (const auto &r : rows) out += std::to_string(r.second) + "\t" + r.first + "\n";
        if (reset) sk::g_trace_counts.clear();
    }
    if (buf && n) {
        const size_t c = out.size() < n - 1 ? out.size() : n - 1;
        memcpy(buf, out.data(), c);
        buf[c] = 0;
    }
    return out.size() + 1;
}

/* Development hook: parse the SK_* environment variables again (tools that sweep a knob inside one process).  Not
 * thread-safe against concurrent launches; product code never calls it. */
void sk_reload_knobs(void) { sk::g_knobs = sk::parse_knobs(); }

/* kappa_d = 4^-d / sqrt(12): the factor the x differences handed to sk_solve_fwd_linear_* carry besides s^2 (it turns the
 * stencil coefficients 1 + g/2 + g^2/12 and 1 - g^2/12 of the refined increment into 1 + g'(sqrt 3 + g') and 1 - g'^2). */
double sk_linear_prescale(int dyadic) { return dyadic < 0 || dyadic > 16 ? 0.0 : 1.0 / ((double)(1LL << (2 * dyadic)) * 3.4641016151377544); }

int sk_plan_wave_shares(int64_t P, int G, int64_t waves, int64_t resident, int wpb, int n_cu, int64_t *first, int64_t *end, int *ppg) {
    if (P < 0 || G < 1 || waves < 1 || wpb < 1 || n_cu < 1 || !first || !end || !ppg) return SK_ERR_BAD_ARG;
    const sk::RankSplit rs = sk::rank_split(P, G, waves, resident, wpb, n_cu, sk::RankW{});
    for (int64_t w = 0; w < waves; ++w) sk::rank_share(rs, w, G, P, ppg[w], first[w], end[w]);
    return rs.nranks;
}

int sk_plan_group_chunks(int64_t A, int64_t B, int64_t PPG, int64_t max_groups, int G, int wpb, int n_cu, int64_t n_groups,
                         int64_t *first, int64_t *slot, int *ppg) {
    if (A < 1 || B < 0 || PPG < 1 || G < 1 || wpb < 1 || n_cu < 1 || !first || !slot || !ppg) return SK_ERR_BAD_ARG;
    const sk::ChunkSplit cs = sk::chunk_split(A, B, PPG, max_groups, G, wpb, n_cu, sk::RankW{});
    const int64_t P = B > 0 ? A * B : A;
    for (int64_t gi = 0; gi < n_groups; ++gi) sk::chunk_share(cs, gi, A, B, P, first[gi], slot[gi], ppg[gi]);
    return cs.nr;
}

const char *sk_status_string(int status) {
    switch (status) {
        case SK_OK: return "ok";
        case SK_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size, unknown scheme or flag)";
        case SK_ERR_UNSUPPORTED: return "shape not supported by the available kernels";
        case SK_ERR_LAUNCH: return "HIP kernel launch failed";
        case SK_ERR_WORKSPACE: return "workspace missing or too small";
        case SK_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

int sk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -SK_ERR_NO_DEVICE;
    return n;
}

int sk_increments_f64(const double *G, int64_t P, int M, int N, double *inc_c, int64_t ld, void *stream) {
    if (!G || !inc_c || P < 0 || M < 2 || N < 2 || (ld != 0 && ld < N - 1)) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    return launch_increments<double>(G, P, M, N, inc_c, ld ? ld : N - 1, (hipStream_t)stream);
}
int sk_increments_f32(const float *G, int64_t P, int M, int N, float *inc_c, int64_t ld, void *stream) {
    if (!G || !inc_c || P < 0 || M < 2 || N < 2 || (ld != 0 && ld < N - 1)) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    return launch_increments<float>(G, P, M, N, inc_c, ld ? ld : N - 1, (hipStream_t)stream);
}
int sk_static_increments_f64(int kind, double param, const double *X, const double *Y, int64_t A, int64_t B, int M, int N,
                             int D, double *inc_c, int64_t ld, void *stream) {
    if (!X || !Y || !inc_c || A < 0 || B < 0 || M < 2 || N < 2 || D < 1 || (kind != 0 && kind != 1)) return SK_ERR_BAD_ARG;
    if ((ld != 0 && ld < N - 1) || (kind == 1 && !(param > 0))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_static_increments<double>(kind, param, X, Y, A, B, M, N, D, inc_c, ld ? ld : N - 1, (hipStream_t)stream);
}
int sk_static_increments_f32(int kind, double param, const float *X, const float *Y, int64_t A, int64_t B, int M, int N,
                             int D, float *inc_c, int64_t ld, void *stream) {
    if (!X || !Y || !inc_c || A < 0 || B < 0 || M < 2 || N < 2 || D < 1 || (kind != 0 && kind != 1)) return SK_ERR_BAD_ARG;
    if ((ld != 0 && ld < N - 1) || (kind == 1 && !(param > 0))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_static_increments<float>(kind, param, X, Y, A, B, M, N, D, inc_c, ld ? ld : N - 1, (hipStream_t)stream);
}

int sk_static_adjoint_f64(int kind, double param, const double *X, const double *Y, const double *W, int64_t ldw,
                          const double *scale, int64_t A, int64_t B, int M, int N, int D, double *out, void *stream) {
    if (!X || !Y || !W || !out || A < 0 || B < 0 || M < 2 || N < 2 || D < 1 || (kind != 0 && kind != 1)) return SK_ERR_BAD_ARG;
    if ((ldw != 0 && ldw < N - 1) || (kind == 1 && !(param > 0))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_static_adjoint<double>(kind, param, X, Y, W, ldw ? ldw : N - 1, scale, A, B, M, N, D, out,
                                         (hipStream_t)stream);
}
int sk_static_adjoint_f32(int kind, double param, const float *X, const float *Y, const float *W, int64_t ldw,
                          const float *scale, int64_t A, int64_t B, int M, int N, int D, float *out, void *stream) {
    if (!X || !Y || !W || !out || A < 0 || B < 0 || M < 2 || N < 2 || D < 1 || (kind != 0 && kind != 1)) return SK_ERR_BAD_ARG;
    if ((ldw != 0 && ldw < N - 1) || (kind == 1 && !(param > 0))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_static_adjoint<float>(kind, param, X, Y, W, ldw ? ldw : N - 1, scale, A, B, M, N, D, out,
                                        (hipStream_t)stream);
}

int sk_linear_adjoint_f64(const double *dYt, int64_t ldy, const double *W, int64_t ldw, const double *scale, int64_t A,
                          int64_t B, int Mc, int Nc, int D, double *out, void *stream) {
    if (!dYt || !W || !out || A < 0 || B < 0 || Mc < 1 || Nc < 1 || D < 1 || ldy < Nc || (ldw != 0 && ldw < Nc)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_linear_adjoint_dyt<double>(dYt, ldy, W, ldw ? ldw : Nc, scale, A, B, Mc, Nc, D, out, (hipStream_t)stream);
}
int sk_linear_adjoint_f32(const double *dYt, int64_t ldy, const float *W, int64_t ldw, const float *scale, int64_t A, int64_t B,
                          int Mc, int Nc, int D, float *out, void *stream) {
    if (!dYt || !W || !out || A < 0 || B < 0 || Mc < 1 || Nc < 1 || D < 1 || ldy < Nc || (ldw != 0 && ldw < Nc)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_linear_adjoint_dyt<float>(dYt, ldy, W, ldw ? ldw : Nc, scale, A, B, Mc, Nc, D, out, (hipStream_t)stream);
}

int sk_static_adjoint2_f64(int kind, double param, const double *X, const double *Y, const double *dXr, int Mrows,
                           const double *W, int64_t ldw, const double *scale, int64_t A, int64_t B, int b0, int M, int N, int D,
                           double *out, void *stream) {
    if (!W || !out || A < 0 || B < 1 || b0 < 0 || b0 > B || M < 2 || N < 2 || D < 1 || (kind != 0 && kind != 1)) return SK_ERR_BAD_ARG;
    if ((ldw != 0 && ldw < N - 1) || (kind == 1 && (!(param > 0) || !X || !Y)) || (kind == 0 && (!dXr || Mrows < M - 1)))
        return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_static_adjoint2<double>(kind, param, X, Y, dXr, Mrows, W, ldw ? ldw : N - 1, scale, A, B, b0, M, N, D, out,
                                          (hipStream_t)stream);
}
int sk_static_adjoint2_f32(int kind, double param, const float *X, const float *Y, const double *dXr, int Mrows, const float *W,
                           int64_t ldw, const float *scale, int64_t A, int64_t B, int b0, int M, int N, int D, float *out,
                           void *stream) {
    if (!W || !out || A < 0 || B < 1 || b0 < 0 || b0 > B || M < 2 || N < 2 || D < 1 || (kind != 0 && kind != 1)) return SK_ERR_BAD_ARG;
    if ((ldw != 0 && ldw < N - 1) || (kind == 1 && (!(param > 0) || !X || !Y)) || (kind == 0 && (!dXr || Mrows < M - 1)))
        return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_static_adjoint2<float>(kind, param, X, Y, dXr, Mrows, W, ldw ? ldw : N - 1, scale, A, B, b0, M, N, D, out,
                                         (hipStream_t)stream);
}

int sk_increments_adjoint_f64(const double *W, int64_t ldw, const double *scale, int64_t P, int M, int N, double *dG,
                              void *stream) {
    if (!W || !dG || P < 0 || M < 2 || N < 2 || (ldw != 0 && ldw < N - 1)) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    return launch_increments_adjoint<double>(W, ldw ? ldw : N - 1, scale, P, M, N, dG, (hipStream_t)stream);
}
int sk_increments_adjoint_f32(const float *W, int64_t ldw, const float *scale, int64_t P, int M, int N, float *dG,
                              void *stream) {
    if (!W || !dG || P < 0 || M < 2 || N < 2 || (ldw != 0 && ldw < N - 1)) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    return launch_increments_adjoint<float>(W, ldw ? ldw : N - 1, scale, P, M, N, dG, (hipStream_t)stream);
}

int sk_solve_fwd_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, double *out_final,
                     double *out_grid, double *out_edges, void *stream) {
    return solve_fwd<double>(inc_c, ld, P, Mc, Nc, dyadic, scheme, flags, out_final, out_grid, out_edges, stream);
}
int sk_solve_fwd_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, float *out_final,
                     float *out_grid, double *out_edges, void *stream) {
    return solve_fwd<float>(inc_c, ld, P, Mc, Nc, dyadic, scheme, flags, out_final, out_grid, out_edges, stream);
}

int sk_solve_fwd_linear_f64(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                            int dyadic, int scheme, double *out_final, void *queue, void *stream) {
    if (D < 1 || !dXr || !dYt || !out_final || A < 0 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    return launch_fwd_fused_linear<double>(dXr, dYt, A, B, Mrows, Ncp, D, g, out_final, nullptr, queue, (hipStream_t)stream);
}
int sk_solve_fwd_linear_f32(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                            int dyadic, int scheme, float *out_final, void *queue, void *stream) {
    if (D < 1 || !dXr || !dYt || !out_final || A < 0 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    return launch_fwd_fused_linear<float>(dXr, dYt, A, B, Mrows, Ncp, D, g, out_final, nullptr, queue, (hipStream_t)stream);
}

int sk_solve_fwd_rbf_f64(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                         int scheme, double inv_sigma, double *out_final, void *queue, void *stream) {
    if (D < 1 || !Xr || !Yt || !out_final || A < 0 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (!(inv_sigma > 0.0) || !(inv_sigma < 1e300)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    return launch_fwd_fused_rbf<double>(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, out_final, nullptr, queue, (hipStream_t)stream);
}
int sk_solve_fwd_rbf_f32(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                         int scheme, double inv_sigma, float *out_final, void *queue, void *stream) {
    if (D < 1 || !Xr || !Yt || !out_final || A < 0 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (!(inv_sigma > 0.0) || !(inv_sigma < 1e300)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    return launch_fwd_fused_rbf<float>(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, out_final, nullptr, queue, (hipStream_t)stream);
}

size_t sk_solve_fwd_static_workspace_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int D) {
    if (P <= 0 || Mc < 1 || Nc < 1) return 0;
    return fused_mb_workspace_bytes(kind, P, Mc, Nc, dyadic, D);
}
int sk_solve_fwd_static_split(int kind, int64_t P, int Mc, int Nc, int dyadic, int D) {
    if (P <= 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 2 || (kind != 0 && kind != 1)) return 0;
    return fused_mb_split(kind, P, Mc, Nc, dyadic, D);
}
int sk_solve_fwd_static_rows(int kind, int Mc, int dyadic) {
    if (Mc < 1 || dyadic < 0 || dyadic > 2) return 0;
    return fused_mb_rows(kind, Mc, dyadic);
}
int sk_route_query(int op, int kind, int D, int M, int N, int dyadic, int scheme, int elem_size, int flags) {
    return route_query(op, kind, D, M, N, dyadic, scheme == SK_SCHEME_NAIVE, elem_size, flags);
}
double sk_cost_query(int which) { return cost_value(which); }
const char *sk_cost_name(int which) { return cost_name(which); }
const char *sk_cost_note(int which) { return cost_note(which); }
int sk_solve_fwd_static_cols(int kind, int Nc) {
    if (Nc < 1 || (kind != 0 && kind != 1)) return 0;
    return fused_mb_cols(kind, Nc);
}
int sk_solve_fwd_static_f64(int kind, double param, const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc,
                            int Ncp, int D, int fd, int dyadic, int scheme, double *out_final, double *edges, void *workspace,
                            size_t workspace_bytes, void *stream) {
    return solve_fwd_static<double>(kind, param, Xr, Yt, 0, A, B, Mrows, Mc, Nc, Ncp, D, fd, dyadic, scheme, out_final, edges, workspace,
                                    workspace_bytes, stream);
}
int sk_solve_fwd_static_f32(int kind, double param, const double *Xr, const void *Yt, int yt_f32, int64_t A, int64_t B, int Mrows,
                            int Mc, int Nc, int Ncp, int D, int fd, int dyadic, int scheme, float *out_final, double *edges,
                            void *workspace, size_t workspace_bytes, void *stream) {
    return solve_fwd_static<float>(kind, param, Xr, Yt, yt_f32, A, B, Mrows, Mc, Nc, Ncp, D, fd, dyadic, scheme, out_final, edges, workspace,
                                   workspace_bytes, stream);
}

int sk_solve_fwd_linear_sym_f64(const double *dXr, const double *dXt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                                int scheme, double *out, void *queue, void *stream) {
    return solve_fwd_sym<double>(0, dXr, dXt, pair_tab, A, Mrows, Mc, Nc, Ncp, D, dyadic, scheme, 0.0, out, queue, stream);
}
int sk_solve_fwd_linear_sym_f32(const double *dXr, const double *dXt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                                int scheme, float *out, void *queue, void *stream) {
    return solve_fwd_sym<float>(0, dXr, dXt, pair_tab, A, Mrows, Mc, Nc, Ncp, D, dyadic, scheme, 0.0, out, queue, stream);
}
int sk_solve_fwd_rbf_sym_f64(const double *Xr, const double *Xt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                                int scheme, double inv_sigma, double *out, void *queue, void *stream) {
    return solve_fwd_sym<double>(1, Xr, Xt, pair_tab, A, Mrows, Mc, Nc, Ncp, D, dyadic, scheme, inv_sigma, out, queue, stream);
}
int sk_solve_fwd_rbf_sym_f32(const double *Xr, const double *Xt, const int *pair_tab, int64_t A, int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic,
                                int scheme, double inv_sigma, float *out, void *queue, void *stream) {
    return solve_fwd_sym<float>(1, Xr, Xt, pair_tab, A, Mrows, Mc, Nc, Ncp, D, dyadic, scheme, inv_sigma, out, queue, stream);
}

int sk_solve_fwd_rbf_edges_f64(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                               int dyadic, int scheme, double inv_sigma, double *out_final, double *edges, void *queue, void *stream) {
    if (D < 1 || !Xr || !Yt || !out_final || !edges || A < 0 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 2) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (!(inv_sigma > 0.0) || !(inv_sigma < 1e300)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    return launch_fwd_fused_rbf<double>(Xr, Yt, A, B, Mrows, Ncp, D, g, inv_sigma, out_final, edges, queue, (hipStream_t)stream);
}

int sk_solve_fwd_linear_edges_f64(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                                  int dyadic, int scheme, double *out_final, double *edges, void *queue, void *stream) {
    if (D < 1 || !dXr || !dYt || !out_final || !edges || A < 0 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 2) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    return launch_fwd_fused_linear<double>(dXr, dYt, A, B, Mrows, Ncp, D, g, out_final, edges, queue, (hipStream_t)stream);
}

int sk_linear_adjoint_fused_f64(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp,
                                int dyadic, int scheme, const double *edges, const double *scale, double *tpart,
                                size_t tpart_doubles, double *err, double *ypart, size_t ypart_doubles, int *ppg_out, int *rows_out,
                                int *ycols_out, const double *kfinal, double screen, double tol, void *rescue_ws,
                                size_t rescue_ws_bytes, void *stream) {
    if (!dXr || !dYt || !edges || A < 0 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (((tpart || ypart) && !err) || (kfinal && !rescue_ws)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    const FusedRescue fr{kfinal, screen, tol, rescue_ws, rescue_ws_bytes};
    return launch_adj_fused_linear(dXr, dYt, A, B, Mrows, Ncp, g, edges, scale, tpart, tpart_doubles, err, ypart, ypart_doubles, ppg_out,
                                   rows_out, ycols_out, rescue_ws ? &fr : nullptr, (hipStream_t)stream);
}

int sk_rbf_adjoint_fused_f64(const double *Xr, const double *Yt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D,
                             int dyadic, int scheme, double sigma, const double *edges, const double *scale, double *gpart,
                             size_t gpart_doubles, double *err, double *ypart, size_t ypart_doubles, int *ppg_out, int *rows_out,
                             int *outw_out, int *ycols_out, const double *kfinal, double screen, double tol, void *rescue_ws,
                             size_t rescue_ws_bytes, void *stream) {
    if (!Xr || !Yt || !edges || A < 0 || B < 0 || Mc < 1 || Nc < 1 || D < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (!(sigma > 0.0) || !(sigma < 1e300) || ((gpart || ypart) && !err) || (kfinal && !rescue_ws)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    const FusedRescue fr{kfinal, screen, tol, rescue_ws, rescue_ws_bytes};
    return launch_adj_fused_rbf(Xr, Yt, A, B, Mrows, Ncp, D, g, 1.0 / sigma, edges, scale, gpart, gpart_doubles, err, ypart, ypart_doubles,
                                ycols_out != nullptr, ppg_out, rows_out, outw_out, ycols_out, rescue_ws ? &fr : nullptr, (hipStream_t)stream);
}

size_t sk_solve_deriv_static_workspace_bytes(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows) {
    if (P < 1 || Mc < 1 || Nc < 1 || D < 1 || dyadic < 0 || dyadic > 16) return 0;
    return deriv_fused_workspace_bytes(P, Mc, Nc, dyadic, D, mrows);
}

int sk_solve_deriv_static_f64(int kind, double param, const double *X0r, const double *X1r, const double *X2r, const double *Yt, int64_t A,
                              int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D, int fd, int dyadic, int scheme, double eps, double *out_k,
                              double *out_kd, double *out_kdd, void *workspace, size_t workspace_bytes, void *stream) {
    if (!X0r || !X1r || !X2r || !Yt || !out_k || !out_kd || !out_kdd || A < 0 || B < 0 || Mc < 1 || Nc < 1 || D < 1 || dyadic < 0 ||
        dyadic > 16 || !(eps > 0))
        return SK_ERR_BAD_ARG;
    if ((kind != 0 && kind != 1) || (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE)) return SK_ERR_BAD_ARG;
    if (kind == 1 && (!(param > 0.0) || !(param < 1e300))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    return launch_deriv_fused(kind, X0r, X1r, X2r, Yt, A, B, Mrows, Ncp, D, fd, g, kind == 1 ? 1.0 / param : 0.0, eps, out_k, out_kd, out_kdd,
                              workspace, workspace_bytes, (hipStream_t)stream);
}

int sk_rbf_adjoint_fused_mb_layout(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows, int *rows, int *outw, int *ncols,
                                   int64_t *edge_doubles, size_t *workspace_bytes) {
    if (P < 1 || Mc < 1 || Nc < 1 || D < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    return adj_fused_mb_layout(P, Mc, Nc, dyadic, D, mrows, rows, outw, ncols, edge_doubles, nullptr, nullptr, workspace_bytes)
               ? SK_OK : SK_ERR_UNSUPPORTED;
}

int sk_rbf_adjoint_fused_mb_f64(const double *Xr, const void *Yt, int yt_f32, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D, int fd,
                                int dyadic, int scheme, double sigma, const double *edges, const double *scale, double *gpart,
                                size_t gpart_doubles, double *n0, size_t n0_doubles, double *err, void *workspace, size_t workspace_bytes,
                                const double *yt64, const double *kfinal, double screen, double tol, void *rescue_ws, size_t rescue_ws_bytes,
                                void *stream) {
    if ((kfinal && !rescue_ws) || (rescue_ws && !yt64)) return SK_ERR_BAD_ARG;
    if (!Xr || !Yt || !edges || !gpart || !n0 || !err || A < 0 || B < 0 || Mc < 1 || Nc < 1 || D < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (!(sigma > 0.0) || !(sigma < 1e300)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    const FusedRescue fr{kfinal, screen, tol, rescue_ws, rescue_ws_bytes};
    return launch_adj_fused_rbf_mb(Xr, Yt, yt_f32, A, B, Mrows, Ncp, D, fd, g, 1.0 / sigma, edges, scale, gpart, gpart_doubles, n0, n0_doubles, err,
                                   workspace, workspace_bytes, rescue_ws ? &fr : nullptr, yt64, (hipStream_t)stream);
}

int sk_linear_adjoint_fused_mb_layout(int64_t P, int Mc, int Nc, int dyadic, int D, int *mrows, int *rows, int *outw, int64_t *edge_doubles,
                                      size_t *workspace_bytes) {
    if (P < 1 || Mc < 1 || Nc < 1 || D < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    return adj_fused_mb_layout(P, Mc, Nc, dyadic, D, mrows, rows, outw, nullptr, edge_doubles, nullptr, nullptr, workspace_bytes, 0)
               ? SK_OK : SK_ERR_UNSUPPORTED;
}

int sk_linear_adjoint_fused_mb_f64(const double *dXr, const double *dYt, int64_t A, int64_t B, int Mrows, int Mc, int Nc, int Ncp, int D, int fd,
                                   int dyadic, int scheme, const double *edges, const double *scale, double *gpart, size_t gpart_doubles,
                                   double *err, void *workspace, size_t workspace_bytes, const double *kfinal, double screen, double tol,
                                   void *rescue_ws, size_t rescue_ws_bytes, void *stream) {
    if (kfinal && !rescue_ws) return SK_ERR_BAD_ARG;
    if (!dXr || !dYt || !edges || !gpart || !err || A < 0 || B < 0 || Mc < 1 || Nc < 1 || D < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    const Geom g = make_geom(B > 0 ? A * B : A, Mc, Nc, dyadic, scheme);
    const FusedRescue fr{kfinal, screen, tol, rescue_ws, rescue_ws_bytes};
    return launch_adj_fused_linear_mb(dXr, dYt, A, B, Mrows, Ncp, D, fd, g, edges, scale, gpart, gpart_doubles, err, workspace, workspace_bytes,
                                      rescue_ws ? &fr : nullptr, (hipStream_t)stream);
}

size_t sk_fused_rescue_workspace_bytes(int kind, int64_t P, int Mc, int Nc, int dyadic, int blocks) {
    if ((kind != 0 && kind != 1) || P < 1 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return 0;
    return fused_rescue_workspace_bytes(kind, P, Mc, Nc, dyadic, blocks);
}

size_t sk_adj_workspace_bytes(int64_t P, int Mc, int Nc, int dyadic, int flags, int elem_size) {
    if (P <= 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16 || (elem_size != 4 && elem_size != 8)) return 0;
    const Geom g = make_geom(P, Mc, Nc, dyadic, SK_SCHEME_DEFAULT);
    const size_t simple = adj_simple_workspace_bytes(g);
    if (flags & (SK_FLAG_EXACT | SK_FLAG_SIMPLE)) return simple;
    const bool fast_shape = dyadic <= (elem_size == 8 ? 2 : 1);   // launch_adj_wave's scope
    const size_t fast = fast_shape ? adj_fast_workspace_bytes(g, elem_size) : 0;
    if (flags & SK_FLAG_FAST_ONLY) return fast;
    return fast > simple ? fast : simple;
}

size_t sk_adj_rescue_slot_bytes(int Mc, int Nc, int dyadic) {
    if (Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return 0;
    if (simple_lds_bytes(make_geom(1, Mc, Nc, dyadic, SK_SCHEME_DEFAULT)) > 160 * 1024) return 0;   // the stored-grid kernel cannot hold it
    return (size_t)2 * (size_t)((Mc << dyadic) + 1) * (size_t)((Nc << dyadic) + 1) * sizeof(double);
}

int sk_adj_rescue_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, const double *err,
                      double tol, double *out_final, double *W, int64_t ldw, void *workspace, size_t workspace_bytes, void *stream) {
    if (bad_common(inc_c, P, Mc, Nc, dyadic, scheme, 0) || !err || !W || (ld != 0 && ld < Nc) || (ldw != 0 && ldw < Nc))
        return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    const Geom g = make_geom(P, Mc, Nc, dyadic, scheme, ld);
    return launch_adj_rescue<double>(inc_c, g, err, tol, out_final, W, ldw ? ldw : Nc, workspace, workspace_bytes,
                                     (hipStream_t)stream);
}
int sk_adj_rescue_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, const double *err,
                      double tol, float *out_final, float *W, int64_t ldw, void *workspace, size_t workspace_bytes, void *stream) {
    if (bad_common(inc_c, P, Mc, Nc, dyadic, scheme, 0) || !err || !W || (ld != 0 && ld < Nc) || (ldw != 0 && ldw < Nc))
        return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    const Geom g = make_geom(P, Mc, Nc, dyadic, scheme, ld);
    return launch_adj_rescue<float>(inc_c, g, err, tol, out_final, W, ldw ? ldw : Nc, workspace, workspace_bytes,
                                    (hipStream_t)stream);
}

int sk_prep_paths_f64(const double *X, int64_t A, int M, int D, int diff, int dim_major, double scale, double *out, int rows, int fd,
                      void *stream) {
    if (!X || !out || A < 0 || M < 1 || D < 1 || fd < D || rows < (diff ? M - 1 : M) || (diff && M < 2)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    if (dim_major == 2) return SK_ERR_UNSUPPORTED;   // the packed fp32 layout is for fp32 paths (exact)
    return launch_prep_paths<double>(X, A, M, D, diff != 0, dim_major != 0, scale, out, rows, fd, (hipStream_t)stream);
}
int sk_prep_paths_f32(const float *X, int64_t A, int M, int D, int diff, int dim_major, double scale, double *out, int rows, int fd,
                      void *stream) {
    if (!X || !out || A < 0 || M < 1 || D < 1 || fd < D || rows < (diff ? M - 1 : M) || (diff && M < 2)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    if (dim_major == 2 && ((fd & 1) || (rows & 1))) return SK_ERR_BAD_ARG;
    return launch_prep_paths<float>(X, A, M, D, diff != 0, dim_major, scale, out, rows, fd, (hipStream_t)stream);
}

int sk_prep_pair_f64(const double *X, int64_t A, int M, const double *Y, int64_t B, int N, int D, int diff, double scale_x, double scale_y,
                     double *out_x, int rows_x, double *out_y, int rows_y, int fd, void *stream) {
    if (!X || !Y || !out_x || !out_y || A < 0 || B < 0 || M < 1 || N < 1 || D < 1 || fd < D) return SK_ERR_BAD_ARG;
    if (rows_x < (diff ? M - 1 : M) || rows_y < (diff ? N - 1 : N) || (diff && (M < 2 || N < 2))) return SK_ERR_BAD_ARG;
    if (A == 0 && B == 0) return SK_OK;
    return launch_prep_pair<double>(X, A, M, Y, B, N, D, diff != 0, scale_x, scale_y, out_x, rows_x, out_y, rows_y, fd, (hipStream_t)stream);
}
int sk_prep_pair_f32(const float *X, int64_t A, int M, const float *Y, int64_t B, int N, int D, int diff, double scale_x, double scale_y,
                     double *out_x, int rows_x, double *out_y, int rows_y, int fd, void *stream) {
    if (!X || !Y || !out_x || !out_y || A < 0 || B < 0 || M < 1 || N < 1 || D < 1 || fd < D) return SK_ERR_BAD_ARG;
    if (rows_x < (diff ? M - 1 : M) || rows_y < (diff ? N - 1 : N) || (diff && (M < 2 || N < 2))) return SK_ERR_BAD_ARG;
    if (A == 0 && B == 0) return SK_OK;
    return launch_prep_pair<float>(X, A, M, Y, B, N, D, diff != 0, scale_x, scale_y, out_x, rows_x, out_y, rows_y, fd, (hipStream_t)stream);
}

size_t sk_strip_edges_bytes(int64_t P, int Mc, int Nc, int dyadic, int elem_size) {
    if (P <= 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > (elem_size == 8 ? 2 : 1) || (elem_size != 4 && elem_size != 8)) return 0;
    const Geom g = make_geom(P, Mc, Nc, dyadic, SK_SCHEME_DEFAULT);
    return (size_t)P * strip_edge_doubles(g, elem_size) * sizeof(double);
}
int sk_solve_fwd_edges_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, double *out_final,
                           double *edges, void *stream) {
    return solve_fwd_edges<double>(inc_c, ld, P, Mc, Nc, dyadic, scheme, out_final, edges, stream);
}
int sk_solve_fwd_edges_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, float *out_final,
                           double *edges, void *stream) {
    return solve_fwd_edges<float>(inc_c, ld, P, Mc, Nc, dyadic, scheme, out_final, edges, stream);
}

int sk_solve_adj_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags,
                     double *out_final, double *W, int64_t ldw, double *out_err, void *workspace, size_t workspace_bytes,
                     void *stream) {
    return solve_adj<double>(inc_c, ld, P, Mc, Nc, dyadic, scheme, flags, out_final, W, ldw, out_err, workspace,
                             workspace_bytes, stream);
}
int sk_solve_adj_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags,
                     float *out_final, float *W, int64_t ldw, double *out_err, void *workspace, size_t workspace_bytes,
                     void *stream) {
    return solve_adj<float>(inc_c, ld, P, Mc, Nc, dyadic, scheme, flags, out_final, W, ldw, out_err, workspace,
                            workspace_bytes, stream);
}

int sk_deriv_increments_f64(const double *G0, const double *G1, const double *G2, double eps, int64_t P, int M, int N,
                            double *inc, double *inc_d, double *inc_dd, int64_t ld, void *stream) {
    if (!G0 || !G1 || !G2 || !inc || !inc_d || !inc_dd || !(eps > 0) || P < 0 || M < 2 || N < 2 || (ld != 0 && ld < N - 1))
        return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    return launch_deriv_increments<double>(G0, G1, G2, eps, P, M, N, inc, inc_d, inc_dd, ld ? ld : N - 1, (hipStream_t)stream);
}
int sk_deriv_increments_f32(const float *G0, const float *G1, const float *G2, double eps, int64_t P, int M, int N,
                            float *inc, float *inc_d, float *inc_dd, int64_t ld, void *stream) {
    if (!G0 || !G1 || !G2 || !inc || !inc_d || !inc_dd || !(eps > 0) || P < 0 || M < 2 || N < 2 || (ld != 0 && ld < N - 1))
        return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    return launch_deriv_increments<float>(G0, G1, G2, eps, P, M, N, inc, inc_d, inc_dd, ld ? ld : N - 1, (hipStream_t)stream);
}
int sk_static_deriv_increments_f64(int kind, double param, const double *X0, const double *X1, const double *X2,
                                   const double *Y, int64_t A, int64_t B, int M, int N, int D, double eps, double *inc,
                                   double *inc_d, double *inc_dd, int64_t ld, void *stream) {
    if (!X0 || !X1 || !X2 || !Y || !inc || !inc_d || !inc_dd || A < 0 || B < 1 || M < 2 || N < 2 || D < 1 || !(eps > 0))
        return SK_ERR_BAD_ARG;
    if ((kind != 0 && kind != 1) || (ld != 0 && ld < N - 1) || (kind == 1 && !(param > 0))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_static_deriv_increments<double>(kind, param, X0, X1, X2, Y, A, B, M, N, D, eps, inc, inc_d, inc_dd,
                                                  ld ? ld : N - 1, (hipStream_t)stream);
}
int sk_static_deriv_increments_f32(int kind, double param, const float *X0, const float *X1, const float *X2, const float *Y,
                                   int64_t A, int64_t B, int M, int N, int D, double eps, float *inc, float *inc_d,
                                   float *inc_dd, int64_t ld, void *stream) {
    if (!X0 || !X1 || !X2 || !Y || !inc || !inc_d || !inc_dd || A < 0 || B < 1 || M < 2 || N < 2 || D < 1 || !(eps > 0))
        return SK_ERR_BAD_ARG;
    if ((kind != 0 && kind != 1) || (ld != 0 && ld < N - 1) || (kind == 1 && !(param > 0))) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_static_deriv_increments<float>(kind, param, X0, X1, X2, Y, A, B, M, N, D, eps, inc, inc_d, inc_dd,
                                                 ld ? ld : N - 1, (hipStream_t)stream);
}
int sk_solve_deriv_f64(const double *inc, const double *inc_d, const double *inc_dd, int64_t ld, int64_t P, int Mc, int Nc,
                       int dyadic, int flags, double *out_k, double *out_kd, double *out_kdd, void *stream) {
    return solve_deriv<double>(inc, inc_d, inc_dd, ld, P, Mc, Nc, dyadic, flags, out_k, out_kd, out_kdd, stream);
}
int sk_solve_deriv_f32(const float *inc, const float *inc_d, const float *inc_dd, int64_t ld, int64_t P, int Mc, int Nc,
                       int dyadic, int flags, float *out_k, float *out_kd, float *out_kdd, void *stream) {
    return solve_deriv<float>(inc, inc_d, inc_dd, ld, P, Mc, Nc, dyadic, flags, out_k, out_kd, out_kdd, stream);
}

int sk_prep_cat_f64(const double *X, int64_t A, const double *Y, int64_t B, int M, int D, int diff, double scale_rows, double scale_rows2,
                    double *out_rows, double *out_rows2, int rows, double *out_cols, int cols, int fd, int *pair_tab, int64_t tri_n, void *stream) {
    if ((A > 0 && !X) || (B > 0 && !Y) || !out_rows || !out_cols || A < 0 || B < 0 || M < 1 || D < 1 || fd < D) return SK_ERR_BAD_ARG;
    if (rows < (diff ? M - 1 : M) || cols < (diff ? M - 1 : M) || (diff && M < 2)) return SK_ERR_BAD_ARG;
    if (pair_tab && (tri_n < -1 || tri_n > B || A + B > 0x7fffffffLL)) return SK_ERR_BAD_ARG;
    if (A + B == 0) return SK_OK;
    return launch_prep_cat<double>(X, A, Y, B, M, D, diff != 0, scale_rows, scale_rows2, out_rows, out_rows2, rows, out_cols, cols, fd, pair_tab, tri_n,
                               (hipStream_t)stream);
}
int sk_prep_cat_f32(const float *X, int64_t A, const float *Y, int64_t B, int M, int D, int diff, double scale_rows, double scale_rows2,
                    double *out_rows, double *out_rows2, int rows, double *out_cols, int cols, int fd, int *pair_tab, int64_t tri_n, void *stream) {
    if ((A > 0 && !X) || (B > 0 && !Y) || !out_rows || !out_cols || A < 0 || B < 0 || M < 1 || D < 1 || fd < D) return SK_ERR_BAD_ARG;
    if (rows < (diff ? M - 1 : M) || cols < (diff ? M - 1 : M) || (diff && M < 2)) return SK_ERR_BAD_ARG;
    if (pair_tab && (tri_n < -1 || tri_n > B || A + B > 0x7fffffffLL)) return SK_ERR_BAD_ARG;
    if (A + B == 0) return SK_OK;
    return launch_prep_cat<float>(X, A, Y, B, M, D, diff != 0, scale_rows, scale_rows2, out_rows, out_rows2, rows, out_cols, cols, fd, pair_tab, tri_n,
                               (hipStream_t)stream);
}

int sk_solve_fwd_loss_f64(int kind, double param, const double *Zr, const double *Zt, const int *pair_tab, int64_t A, int64_t B, int64_t tri_n,
                          int Mrows, int Mc, int Nc, int Ncp, int D, int dyadic, int scheme, double *out, double *edges, void *queue, void *stream) {
    if (D < 1 || !Zr || !Zt || !out || A < 1 || B < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return SK_ERR_BAD_ARG;
    // the pair table of sk_prep_cat_* lies right behind the staged columns (the kernel finds it from Zt: no pointer of its own in the
    // scalar registers of every launch)
    if (reinterpret_cast<const void *>(pair_tab) != reinterpret_cast<const void *>(Zt + (A + B) * (int64_t)8 * Ncp)) return SK_ERR_BAD_ARG;
    if ((kind != 0 && kind != 1) || (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE)) return SK_ERR_BAD_ARG;
    if (kind == 1 && (!(param > 0.0) || !(param < 1e300))) return SK_ERR_BAD_ARG;
    if (tri_n != 0 && tri_n != B) return SK_ERR_BAD_ARG;
    if (dyadic > 2) return SK_ERR_UNSUPPORTED;
    const int64_t Bz = A + B;
    const int64_t loss[2] = {tri_n, A};
    const Geom g = make_geom(A * Bz + (tri_n > 1 ? tri_n * (tri_n - 1) / 2 : 0), Mc, Nc, dyadic, scheme);
    if (kind == 0) return launch_fwd_fused_linear<double>(Zr, Zt, A, Bz, Mrows, Ncp, D, g, out, edges, queue, (hipStream_t)stream, 2, loss);
    return launch_fwd_fused_rbf<double>(Zr, Zt, A, Bz, Mrows, Ncp, D, g, param, out, edges, queue, (hipStream_t)stream, 2, loss);
}

int sk_loss_value_f64(const double *out, int64_t A, int64_t B, int with_yy, double *value, double *wb, void *stream) {
    if (!out || !value || A < 1 || B < 1) return SK_ERR_BAD_ARG;
    return launch_loss_value(out, A, B, with_yy, value, wb, (hipStream_t)stream);
}
int sk_loss_weights_f64(int64_t A, int64_t B, const double *grad_out, double *go, void *stream) {
    if (!go || A < 1 || B < 1) return SK_ERR_BAD_ARG;
    return launch_loss_weights(A, B, grad_out, go, (hipStream_t)stream);
}
int sk_rbf_adjoint_finish_f64(const double *gpart, int64_t A, int64_t chunks, int rows, int outw, const double *X, int M, int D, double sigma,
                              const double *gscale, double *grad, void *stream) {
    if (!gpart || !X || !grad || A < 0 || chunks < 1 || M < 1 || D < 1 || rows < M || outw < 2 + D || !(sigma > 0.0)) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_rbf_adjoint_finish(gpart, A, chunks, rows, outw, X, M, D, sigma, gscale, grad, (hipStream_t)stream);
}
int sk_linear_adjoint_finish_f64(const double *tpart, int64_t A, int64_t chunks, int rows, int M, int D, double scale2, const double *gscale,
                                 double *grad, void *stream) {
    if (!tpart || !grad || A < 0 || chunks < 1 || M < 2 || D < 1 || D > 8 || rows < M - 1) return SK_ERR_BAD_ARG;
    if (A == 0) return SK_OK;
    return launch_linear_adjoint_finish(tpart, A, chunks, rows, M, D, scale2, gscale, grad, (hipStream_t)stream);
}

}  // extern "C"
