// sk_loss.hip -- the glue of a training-sized loss step as single launches.
//
// compute_mmd(X, Y).backward() on a few dozen paths is three PDE launches and, in the reference's composition
// (sigkernel.py:180-197 over three compute_Gram calls, :404-416 for their backward), some twenty elementwise / reduction
// kernels of a few microseconds each around them: concatenation, staging of each batch twice, multiply + sum per matrix,
// the weights of d loss / dK, the fold of the adjoint's partial sums into dL/dX.  On MI355X every dispatch costs 4-5 us whatever
// it does, so at 32 x 32 paths a third of the step was glue.  Here each of those stages is ONE launch:
//   k_prep_cat            Z = [X; Y] staged for the fused kernels in both layouts (rows [A+B][rows][8], cols [A+B][8][cols]),
//                         straight from the two batches -- no concatenated copy;
//   k_loss_value          the scalar  sum_{a != b} K_XX / (A (A-1))  -  2 mean(K_XY)  [+ sum_{i != j} K_YY / (B (B-1))]  from the
//                         output of sk_solve_fwd_loss_f64 (rectangle K(X, [X; Y]) + strict triangle of K(Y, Y)), in a fixed order;
//   k_loss_weights        d loss / dK of the rectangle times the upstream gradient (a device scalar: no host read-back), with the
//                         reference's 2x rule for the K_XX block (sigkernel.py:410-412);
//   k_*_adjoint_finish    the fused adjoints' partial sums [A][chunks][rows][w] -> dL/dX (A, M, D), chunks added in ascending order.
// All results are deterministic (no atomics; fixed reduction trees).
#include "sk_internal.h"

namespace sk {
namespace {

template <typename T, bool DIFF>
__global__ __launch_bounds__(256) void k_prep_cat(const T *__restrict__ X, int64_t A, const T *__restrict__ Y, int64_t B, int M, int D,
                                                  double scale_rows, double scale_rows2, double *__restrict__ out_rows,
                                                  double *__restrict__ out_rows2, int rows, double *__restrict__ out_cols, int cols, int FDp,
                                                  int *__restrict__ pair_tab, int64_t tri_n) {
    const int64_t Z = A + B;
    const int64_t nr = Z * (int64_t)rows * FDp, nc = Z * (int64_t)cols * FDp;
    const int nvalid = DIFF ? M - 1 : M;
    // the pairs of a triangular layout, (a, b) per pair.  tri_n >= 0, the LOSS layout: the rectangle A x (A + B), then the strict upper
    // triangle (i < j, row-major) of the tri_n paths from Z[A] on;  tri_n = -1, the SYMMETRIC Gram: the inclusive upper triangle (a <= b,
    // row-major) of all A + B paths
    const bool sym = tri_n < 0;
    const int64_t P_rect = pair_tab && !sym ? A * Z : 0;
    const int64_t P_all = !pair_tab ? 0 : sym ? Z * (Z + 1) / 2 : P_rect + (tri_n > 1 ? tri_n * (tri_n - 1) / 2 : 0);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P_all; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t a, b;
        if (p < P_rect) {
            a = p / Z;
            b = p - a * Z;
        } else {          // pair q of the inclusive triangle of n1 paths; the strict triangle of n = that of n - 1 with b shifted by one
            const int64_t q = p - P_rect, n1 = sym ? Z : tri_n - 1, off = sym ? 0 : A, shift = sym ? 0 : 1;
            const double t = (double)(2 * n1 + 1);
            int64_t r = (int64_t)((t - sqrt(t * t - 8.0 * (double)q)) * 0.5);
            r = r < 0 ? 0 : (r >= n1 ? n1 - 1 : r);
            while (r > 0 && r * n1 - r * (r - 1) / 2 > q) --r;
            while (r + 1 < n1 && (r + 1) * n1 - (r + 1) * r / 2 <= q) ++r;
            a = off + r;
            b = off + r + (q - (r * n1 - r * (r - 1) / 2)) + shift;
        }
        pair_tab[2 * p] = (int)a;
        pair_tab[2 * p + 1] = (int)b;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nr + nc; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t z;
        int p, j;
        const bool is_row = i < nr;
        if (is_row) {          // [z][p][j]
            z = i / ((int64_t)rows * FDp);
            const int rem = (int)(i - z * (int64_t)rows * FDp);
            p = rem / FDp;
            j = rem - p * FDp;
        } else {               // [z][j][q], q < cols
            const int64_t k = i - nr;
            z = k / ((int64_t)cols * FDp);
            const int rem = (int)(k - z * (int64_t)cols * FDp);
            j = rem / cols;
            p = rem - j * cols;
        }
        double v = 0.0;
        if (p < nvalid && j < D) {
            const T *x = (z < A ? X + z * (int64_t)M * D : Y + (z - A) * (int64_t)M * D) + (int64_t)p * D + j;
            v = DIFF ? ((double)x[D] - (double)x[0]) : (double)x[0];
        }
        if (is_row) {
            out_rows[i] = v * scale_rows;
            if (out_rows2) out_rows2[i] = v * scale_rows2;
        } else {
            out_cols[i - nr] = v;
        }
    }
}

constexpr int LV_THREADS = 1024;
constexpr int LV_BATCH = 8;

// one workgroup: thread t adds entries t, t + 1024, ... of each part in ascending order (loads in batches of eight: a launch this
// small is bound by the latency of its dependent load -> add chain), then a fixed tree over the threads.  wb (nullable, [A (A+B)]):
// d value / dK of the rectangle with the K_XX block doubled -- the upstream gradient of the adjoint, written on the way.
__global__ __launch_bounds__(LV_THREADS) void k_loss_value(const double *__restrict__ out, int A, int B, int with_yy,
                                                           double *__restrict__ value, double *__restrict__ wb) {
    __shared__ double red[LV_THREADS];
    const unsigned Bz = (unsigned)(A + B), P_rect = (unsigned)A * Bz, P_tri = with_yy && B > 1 ? (unsigned)B * (unsigned)(B - 1) / 2 : 0;
    const double wxx = A > 1 ? 1.0 / ((double)A * (double)(A - 1)) : 0.0, wxy = -2.0 / ((double)A * (double)B);
    const double wyy = B > 1 ? 2.0 / ((double)B * (double)(B - 1)) : 0.0;
    double sxx = 0.0, sxy = 0.0, syy = 0.0;
    for (unsigned p0 = threadIdx.x; p0 < P_rect; p0 += LV_THREADS * LV_BATCH) {
        double k[LV_BATCH];
#pragma unroll
        for (int i = 0; i < LV_BATCH; ++i) {
            const unsigned p = p0 + (unsigned)i * LV_THREADS;
            k[i] = p < P_rect ? out[p] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < LV_BATCH; ++i) {
            const unsigned p = p0 + (unsigned)i * LV_THREADS;
            if (p < P_rect) {
                const unsigned a = p / Bz, b = p - a * Bz;
                const bool xy = b >= (unsigned)A, diag = b == a;
                if (xy) sxy += k[i];
                else if (!diag) sxx += k[i];
                if (wb) wb[p] = xy ? wxy : (diag ? 0.0 : 2.0 * wxx);
            }
        }
    }
    for (unsigned q0 = threadIdx.x; q0 < P_tri; q0 += LV_THREADS * LV_BATCH) {
        double k[LV_BATCH];
#pragma unroll
        for (int i = 0; i < LV_BATCH; ++i) {
            const unsigned q = q0 + (unsigned)i * LV_THREADS;
            k[i] = q < P_tri ? out[P_rect + q] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < LV_BATCH; ++i) syy += k[i];
    }
    red[threadIdx.x] = sxx * wxx + sxy * wxy + syy * wyy;
    __syncthreads();
    for (int s = LV_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *value = red[0];
}

__global__ __launch_bounds__(256) void k_loss_weights(int64_t A, int64_t B, const double *__restrict__ grad_out, double *__restrict__ go) {
    const int64_t Bz = A + B, P_rect = A * Bz;
    const double g = grad_out ? *grad_out : 1.0;
    const double wxx = A > 1 ? 2.0 / ((double)A * (double)(A - 1)) : 0.0, wxy = -2.0 / ((double)A * (double)B);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P_rect; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = p / Bz, b = p - a * Bz;
        go[p] = g * (b >= A ? wxy : (b != a ? wxx : 0.0));
    }
}

constexpr int FIN_BATCH = 8;
constexpr int FIN_SPLIT = 4;      // lanes per output of the finish kernels (a power of two: the lanes of an output are adjacent)

// v of the FIN_SPLIT adjacent lanes of an output -> (v0 + v1) + (v2 + v3) on the first of them (every lane of the wave takes part: the
// loops below run whole groups, `t < n * FIN_SPLIT` with blockDim a multiple of FIN_SPLIT -- a group is inside the bound or outside it)
__device__ __forceinline__ void fin_combine(double &v) {
    v += __shfl_down(v, 1, FIN_SPLIT);
    v += __shfl_down(v, 2, FIN_SPLIT);
}

// gpart [A][chunks][rows][outw]: node row r < M of x_a: cs = sum_c [a][c][r][0], accd[j] = sum_c [a][c][r][2 + j];
// dL/dx_a[r][j] = gscale (-2 / sigma) (x_a[r][j] cs - accd[j]); gscale: a device scalar (nullable = 1) -- the adjoint is linear in the
// upstream gradient, so a scalar factor of it can be applied here instead of to every pair's weight
__global__ __launch_bounds__(256) void k_rbf_adjoint_finish(const double *__restrict__ gpart, int64_t A, int64_t chunks, int rows, int outw,
                                                            const double *__restrict__ X, int M, int D, double c,
                                                            const double *__restrict__ gscale, double *__restrict__ grad) {
    const int64_t n = A * (int64_t)M * D;
    const double gs = gscale ? *gscale : 1.0;
    const int64_t step = (int64_t)rows * outw;
    // FIN_SPLIT adjacent lanes share an output: lane s adds the s-th quarter of the chunks in ascending order, the quarters are combined
    // as (q0 + q1) + (q2 + q3) -- a fixed tree; a training-sized batch has too few outputs to hide 8 dependent rounds of loads otherwise
    const int64_t quarter = (chunks + FIN_SPLIT - 1) / FIN_SPLIT;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * FIN_SPLIT; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / FIN_SPLIT;
        const int sub = (int)(t - i * FIN_SPLIT);
        const int64_t a = i / ((int64_t)M * D);
        const int rem = (int)(i - a * (int64_t)M * D);
        const int r = rem / D, j = rem - r * D;
        const double *src = gpart + (a * chunks * rows + r) * (int64_t)outw;
        const int64_t c_lo = sub * quarter, c_hi = c_lo + quarter < chunks ? c_lo + quarter : chunks;
        double cs = 0.0, acc = 0.0;
        for (int64_t ch0 = c_lo; ch0 < c_hi; ch0 += FIN_BATCH) {     // chunks in ascending order, eight loads in flight at a time
            double u[FIN_BATCH], v[FIN_BATCH];
#pragma unroll
            for (int k = 0; k < FIN_BATCH; ++k) {
                const bool in = ch0 + k < c_hi;
                u[k] = in ? src[(ch0 + k) * step] : 0.0;
                v[k] = in ? src[(ch0 + k) * step + 2 + j] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < FIN_BATCH; ++k) { cs += u[k]; acc += v[k]; }
        }
        fin_combine(cs);
        fin_combine(acc);
        if (sub == 0) grad[i] = gs * (c * (X[i] * cs - acc));
    }
}

// tpart [A][chunks][rows][8], coarse rows FLIPPED (row rows - 1 - p holds coarse row p): T[a][p][j] = sum_c [a][c][rows - 1 - p][j];
// dL/dx_a[r][j] = gscale scale2 (T[a][r - 1][j] - T[a][r][j])  (d inc[p][q] / d x[p + 1] = + s^2 dy[q], / d x[p] = - s^2 dy[q])
__global__ __launch_bounds__(256) void k_linear_adjoint_finish(const double *__restrict__ tpart, int64_t A, int64_t chunks, int rows, int M, int D,
                                                               double scale2, const double *__restrict__ gscale, double *__restrict__ grad) {
    const int64_t n = A * (int64_t)M * D;
    const int Mc = M - 1;
    const double gs = gscale ? *gscale : 1.0;
    const int64_t quarter = (chunks + FIN_SPLIT - 1) / FIN_SPLIT;      // (as k_rbf_adjoint_finish)
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * FIN_SPLIT; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / FIN_SPLIT;
        const int sub = (int)(t - i * FIN_SPLIT);
        const int64_t a = i / ((int64_t)M * D);
        const int rem = (int)(i - a * (int64_t)M * D);
        const int r = rem / D, j = rem - r * D;
        const double *src = tpart + a * chunks * rows * (int64_t)8 + j;
        const int64_t c_lo = sub * quarter, c_hi = c_lo + quarter < chunks ? c_lo + quarter : chunks;
        double up = 0.0, dn = 0.0;     // T[r - 1], T[r]
        for (int64_t ch0 = c_lo; ch0 < c_hi; ch0 += FIN_BATCH) {
            double u[FIN_BATCH], v[FIN_BATCH];
#pragma unroll
            for (int k = 0; k < FIN_BATCH; ++k) {
                const bool in = ch0 + k < c_hi;
                u[k] = in && r >= 1 ? src[((ch0 + k) * rows + (rows - r)) * (int64_t)8] : 0.0;
                v[k] = in && r < Mc ? src[((ch0 + k) * rows + (rows - 1 - r)) * (int64_t)8] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < FIN_BATCH; ++k) { up += u[k]; dn += v[k]; }
        }
        fin_combine(up);
        fin_combine(dn);
        if (sub == 0) grad[i] = gs * (scale2 * (up - dn));
    }
}

// ONE chunk per a (paired batches: a slot per pair; small Gram rows): one lane per output, nothing to add up -- the FIN_SPLIT form above
// leaves three of four lanes idle there and ran at 0.5 TB/s on the 262 144 pairs of a big paired batch (2.0 ms of an 18 ms step)
__global__ __launch_bounds__(256) void k_rbf_adjoint_finish1(const double *__restrict__ gpart, int64_t A, int rows, int outw, const double *__restrict__ X,
                                                             int M, int D, double c, const double *__restrict__ gscale, double *__restrict__ grad) {
    const int64_t n = A * (int64_t)M * D;
    const double gs = gscale ? *gscale : 1.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = i / ((int64_t)M * D);
        const int rem = (int)(i - a * (int64_t)M * D);
        const int r = rem / D, j = rem - r * D;
        const double *src = gpart + (a * rows + r) * (int64_t)outw;
        grad[i] = gs * (c * (X[i] * src[0] - src[2 + j]));
    }
}
__global__ __launch_bounds__(256) void k_linear_adjoint_finish1(const double *__restrict__ tpart, int64_t A, int rows, int M, int D, double scale2,
                                                                const double *__restrict__ gscale, double *__restrict__ grad) {
    const int64_t n = A * (int64_t)M * D;
    const int Mc = M - 1;
    const double gs = gscale ? *gscale : 1.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = i / ((int64_t)M * D);
        const int rem = (int)(i - a * (int64_t)M * D);
        const int r = rem / D, j = rem - r * D;
        const double *src = tpart + a * rows * (int64_t)8 + j;
        const double up = r >= 1 ? src[(rows - r) * (int64_t)8] : 0.0, dn = r < Mc ? src[(rows - 1 - r) * (int64_t)8] : 0.0;
        grad[i] = gs * (scale2 * (up - dn));
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

template <typename T>
int launch_prep_cat(const T *X, int64_t A, const T *Y, int64_t B, int M, int D, int diff, double scale_rows, double scale_rows2, double *out_rows,
                    double *out_rows2, int rows, double *out_cols, int cols, int FDp, int *pair_tab, int64_t tri_n, hipStream_t s) {
    const int64_t n = (A + B) * (int64_t)(rows + cols) * FDp;
    if (diff) SK_LAUNCH((k_prep_cat<T, true>), dim3(grid_for(n)), dim3(256), 0, s, X, A, Y, B, M, D, scale_rows, scale_rows2, out_rows, out_rows2, rows, out_cols, cols, FDp, pair_tab, tri_n);
    else SK_LAUNCH((k_prep_cat<T, false>), dim3(grid_for(n)), dim3(256), 0, s, X, A, Y, B, M, D, scale_rows, scale_rows2, out_rows, out_rows2, rows, out_cols, cols, FDp, pair_tab, tri_n);
    return check_launch();
}
template int launch_prep_cat<double>(const double *, int64_t, const double *, int64_t, int, int, int, double, double, double *, double *, int, double *,
                                     int, int, int *, int64_t, hipStream_t);
template int launch_prep_cat<float>(const float *, int64_t, const float *, int64_t, int, int, int, double, double, double *, double *, int, double *,
                                    int, int, int *, int64_t, hipStream_t);

int launch_loss_value(const double *out, int64_t A, int64_t B, int with_yy, double *value, double *wb, hipStream_t s) {
    // (32-bit indices; the loss wrappers' merged route ends far below.  The triangle of K(Y, Y) only counts when it is there.)
    if (A < 0 || B < 0 || (A + B) * A + (with_yy ? B * (B - 1) / 2 : 0) >= 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    SK_LAUNCH(k_loss_value, dim3(1), dim3(LV_THREADS), 0, s, out, (int)A, (int)B, with_yy, value, wb);
    return check_launch();
}

int launch_loss_weights(int64_t A, int64_t B, const double *grad_out, double *go, hipStream_t s) {
    SK_LAUNCH(k_loss_weights, dim3(grid_for(A * (A + B))), dim3(256), 0, s, A, B, grad_out, go);
    return check_launch();
}

int launch_rbf_adjoint_finish(const double *gpart, int64_t A, int64_t chunks, int rows, int outw, const double *X, int M, int D, double sigma,
                              const double *gscale, double *grad, hipStream_t s) {
    if (chunks == 1) {
        SK_LAUNCH(k_rbf_adjoint_finish1, dim3(grid_for(A * (int64_t)M * D)), dim3(256), 0, s, gpart, A, rows, outw, X, M, D, -2.0 / sigma, gscale, grad);
        return check_launch();
    }
    SK_LAUNCH(k_rbf_adjoint_finish, dim3(grid_for(A * (int64_t)M * D * FIN_SPLIT)), dim3(256), 0, s, gpart, A, chunks, rows, outw, X, M, D,
                       -2.0 / sigma, gscale, grad);
    return check_launch();
}

int launch_linear_adjoint_finish(const double *tpart, int64_t A, int64_t chunks, int rows, int M, int D, double scale2, const double *gscale,
                                 double *grad, hipStream_t s) {
    if (chunks == 1) {
        SK_LAUNCH(k_linear_adjoint_finish1, dim3(grid_for(A * (int64_t)M * D)), dim3(256), 0, s, tpart, A, rows, M, D, scale2, gscale, grad);
        return check_launch();
    }
    SK_LAUNCH(k_linear_adjoint_finish, dim3(grid_for(A * (int64_t)M * D * FIN_SPLIT)), dim3(256), 0, s, tpart, A, chunks, rows, M, D, scale2, gscale, grad);
    return check_launch();
}

}  // namespace sk
