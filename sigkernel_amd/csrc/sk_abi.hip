// sk_abi.hip -- the extern "C" surface declared in include/sigkernel_amd.h.
// Argument checking and kernel selection only; no torch types, no allocation, no synchronisation.
#include "sk_internal.h"

using namespace sk;

namespace {

bool bad_common(const void *inc, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags) {
    if (!inc || P < 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return true;
    if (scheme != SK_SCHEME_DEFAULT && scheme != SK_SCHEME_NAIVE) return true;
    if (flags & ~(SK_FLAG_EXACT | SK_FLAG_SIMPLE | SK_FLAG_FAST_ONLY)) return true;
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
    if (!(flags & (SK_FLAG_EXACT | SK_FLAG_SIMPLE)) && out_final && !out_grid && !out_edges) {
        const int rc = launch_fwd_wave<T>(inc_c, g.ld, g, out_final, (hipStream_t)stream);
        if (rc != SK_ERR_UNSUPPORTED || (flags & SK_FLAG_FAST_ONLY)) return rc;
    } else if (flags & SK_FLAG_FAST_ONLY) {
        return SK_ERR_UNSUPPORTED;
    }
    return launch_fwd_simple<T>(inc_c, g, out_final, out_grid, out_edges, (hipStream_t)stream);
}

template <typename T>
int solve_adj(const T *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, T *out_final, T *W,
              void *ws, size_t ws_bytes, void *stream) {
    if (bad_common(inc_c, P, Mc, Nc, dyadic, scheme, flags) || !W || (ld != 0 && ld < Nc)) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    const Geom g = make_geom(P, Mc, Nc, dyadic, scheme, ld);
    return launch_adj_simple<T>(inc_c, g, out_final, W, ws, ws_bytes, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int sk_version(void) { return 100; }

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
int sk_increments_adjoint_f64(const double *W, const double *scale, int64_t P, int M, int N, double *dG, void *stream) {
    if (!W || !dG || P < 0 || M < 2 || N < 2) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    return launch_increments_adjoint<double>(W, scale, P, M, N, dG, (hipStream_t)stream);
}
int sk_increments_adjoint_f32(const float *W, const float *scale, int64_t P, int M, int N, float *dG, void *stream) {
    if (!W || !dG || P < 0 || M < 2 || N < 2) return SK_ERR_BAD_ARG;
    if (P == 0) return SK_OK;
    return launch_increments_adjoint<float>(W, scale, P, M, N, dG, (hipStream_t)stream);
}

int sk_solve_fwd_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, double *out_final,
                     double *out_grid, double *out_edges, void *stream) {
    return solve_fwd<double>(inc_c, ld, P, Mc, Nc, dyadic, scheme, flags, out_final, out_grid, out_edges, stream);
}
int sk_solve_fwd_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, float *out_final,
                     float *out_grid, double *out_edges, void *stream) {
    return solve_fwd<float>(inc_c, ld, P, Mc, Nc, dyadic, scheme, flags, out_final, out_grid, out_edges, stream);
}

size_t sk_adj_workspace_bytes(int64_t P, int Mc, int Nc, int dyadic, int flags, int elem_size) {
    (void)flags; (void)elem_size;
    if (P <= 0 || Mc < 1 || Nc < 1 || dyadic < 0 || dyadic > 16) return 0;
    return adj_simple_workspace_bytes(make_geom(P, Mc, Nc, dyadic, SK_SCHEME_DEFAULT));
}

int sk_solve_adj_f64(const double *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, double *out_final,
                     double *W, void *workspace, size_t workspace_bytes, void *stream) {
    return solve_adj<double>(inc_c, ld, P, Mc, Nc, dyadic, scheme, flags, out_final, W, workspace, workspace_bytes, stream);
}
int sk_solve_adj_f32(const float *inc_c, int64_t ld, int64_t P, int Mc, int Nc, int dyadic, int scheme, int flags, float *out_final,
                     float *W, void *workspace, size_t workspace_bytes, void *stream) {
    return solve_adj<float>(inc_c, ld, P, Mc, Nc, dyadic, scheme, flags, out_final, W, workspace, workspace_bytes, stream);
}

}  // extern "C"
