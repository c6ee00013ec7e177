// sk_prep.hip -- path staging for the fused kernels (sk_wave_fused.hip, sk_wave_adj_fused.hip).
//
// The fused solvers read the two path batches in fp64, zero-padded, the first row-major and the second dimension-major:
//     rows [A][Mrows][FD]   (differences s^2 (x[p+1]-x[p]) or points x[p])
//     cols [B][FD][Ncp]     (differences y[q+1]-y[q] or points y[q])
// One launch builds one of the two from the caller's dense (batch, length, dim) tensor of either precision, padding
// included, so that the host layer issues no elementwise torch ops or zero-fills per call (they were 0.1-0.2 ms of a
// 0.3 ms Gram at the C2 size).  The differences are formed from the UP-CAST points, as the edge-keeping forward and the
// fused adjoint both expect.  Replaces the tensor arithmetic around static_kernels.py:26-33 / :58-73 that the reference
// performs on the device before its solver launch.
#include "sk_internal.h"

namespace sk {
namespace {

template <typename T, bool DIFF, bool DIM_MAJOR>
__global__ __launch_bounds__(256) void k_prep_paths(const T *__restrict__ X, int64_t A, int M, int D, double scale,
                                                    double *__restrict__ out, int rows, int FDp) {
    // one thread per output element; consecutive threads write consecutive addresses
    const int64_t n = A * (int64_t)rows * FDp;
    const int nvalid = DIFF ? M - 1 : M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t a;
        int p, j;
        if (DIM_MAJOR) {   // [a][j][p], p < rows
            a = i / ((int64_t)rows * FDp);
            const int rem = (int)(i - a * (int64_t)rows * FDp);
            j = rem / rows;
            p = rem - j * rows;
        } else {           // [a][p][j]
            a = i / ((int64_t)rows * FDp);
            const int rem = (int)(i - a * (int64_t)rows * FDp);
            p = rem / FDp;
            j = rem - p * FDp;
        }
        double v = 0.0;
        if (p < nvalid && j < D) {
            const T *x = X + (a * M + p) * (int64_t)D + j;
            v = DIFF ? ((double)x[D] - (double)x[0]) * scale : (double)x[0] * scale;
        }
        out[i] = v;
    }
}

// Layout 2 (the fp32 y ring of sk_wave_fused_mb.hip): per path fd / 2 rows of packed fp32 points followed by ONE row of
// fp64 squared norms -- as bytes, a dimension-major fp64 array [A][fd / 2 + 1][rows]:
//   rows 0 .. fd/2 - 1: float [rows / 2][4], the 4 floats of a unit = {dim 2j col 2u, dim 2j col 2u+1, dim 2j+1 col 2u,
//                       dim 2j+1 col 2u+1};    row fd / 2: double |x_p|^2 (of the scaled, up-cast points), zero past the path.
// Points only (an fp32 point is stored exactly; a difference would be rounded).
template <typename T>
__global__ __launch_bounds__(256) void k_prep_paths_packed32(const T *__restrict__ X, int64_t A, int M, int D, double scale,
                                                             float *__restrict__ out, int rows, int FDp) {
    const int64_t per = (int64_t)rows * (FDp + 2);     // floats per path
    const int64_t n = A * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = i / per;
        int rem = (int)(i - a * per);
        if (rem < rows * FDp) {
            const int jp = rem / (2 * rows);
            rem -= jp * 2 * rows;
            const int uu = rem >> 2, w = rem & 3;
            const int j = 2 * jp + (w >> 1), p = 2 * uu + (w & 1);
            float v = 0.f;
            if (p < M && j < D) v = (float)((double)X[(a * M + p) * (int64_t)D + j] * scale);
            out[i] = v;
        } else {
            rem -= rows * FDp;
            if (rem & 1) continue;       // one thread per double
            const int p = rem >> 1;
            double q2 = 0.0;
            if (p < M)
                for (int j = 0; j < D; ++j) {
                    const double v = (double)(float)((double)X[(a * M + p) * (int64_t)D + j] * scale);   // the value the ring holds
                    q2 = fma(v, v, q2);
                }
            *reinterpret_cast<double *>(out + i) = q2;
        }
    }
}

// both staged arrays of a call in ONE launch: rows [A][rows_x][FDp] from X and cols [B][FDp][rows_y] from Y (small calls are
// launch-bound: a C1-sized compute_Gram was 2 staging launches + 1 solve)
template <typename T, bool DIFF>
__global__ __launch_bounds__(256) void k_prep_pair(const T *__restrict__ X, int64_t A, int M, const T *__restrict__ Y, int64_t B, int N, int D,
                                                   double scale_x, double scale_y, double *__restrict__ out_x, int rows_x,
                                                   double *__restrict__ out_y, int rows_y, int FDp) {
    const int64_t nx = A * (int64_t)rows_x * FDp, ny = B * (int64_t)rows_y * FDp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nx + ny; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nx) {          // [a][p][j]
            const int64_t a = i / ((int64_t)rows_x * FDp);
            const int rem = (int)(i - a * (int64_t)rows_x * FDp);
            const int p = rem / FDp, j = rem - p * FDp;
            double v = 0.0;
            if (p < (DIFF ? M - 1 : M) && j < D) {
                const T *x = X + (a * M + p) * (int64_t)D + j;
                v = DIFF ? ((double)x[D] - (double)x[0]) * scale_x : (double)x[0] * scale_x;
            }
            out_x[i] = v;
        } else {               // [b][j][q], q < rows_y
            const int64_t k = i - nx;
            const int64_t b = k / ((int64_t)rows_y * FDp);
            const int rem = (int)(k - b * (int64_t)rows_y * FDp);
            const int j = rem / rows_y, q = rem - j * rows_y;
            double v = 0.0;
            if (q < (DIFF ? N - 1 : N) && j < D) {
                const T *y = Y + (b * N + q) * (int64_t)D + j;
                v = DIFF ? ((double)y[D] - (double)y[0]) * scale_y : (double)y[0] * scale_y;
            }
            out_y[k] = v;
        }
    }
}

}  // namespace

template <typename T>
int launch_prep_pair(const T *X, int64_t A, int M, const T *Y, int64_t B, int N, int D, int diff, double scale_x, double scale_y, double *out_x,
                     int rows_x, double *out_y, int rows_y, int FDp, hipStream_t s) {
    const int64_t n = (A * (int64_t)rows_x + B * (int64_t)rows_y) * FDp;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    if (diff) SK_LAUNCH((k_prep_pair<T, true>), dim3((unsigned)blocks), dim3(256), 0, s, X, A, M, Y, B, N, D, scale_x, scale_y, out_x, rows_x, out_y, rows_y, FDp);
    else SK_LAUNCH((k_prep_pair<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, X, A, M, Y, B, N, D, scale_x, scale_y, out_x, rows_x, out_y, rows_y, FDp);
    return check_launch();
}
template int launch_prep_pair<double>(const double *, int64_t, int, const double *, int64_t, int, int, int, double, double, double *, int, double *, int,
                                      int, hipStream_t);
template int launch_prep_pair<float>(const float *, int64_t, int, const float *, int64_t, int, int, int, double, double, double *, int, double *, int,
                                     int, hipStream_t);

template <typename T>
int launch_prep_paths(const T *X, int64_t A, int M, int D, int diff, int dim_major, double scale, double *out, int rows, int FDp,
                      hipStream_t s) {
    const int64_t n = A * (int64_t)rows * FDp;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks), b(256);
    if (dim_major == 2) {
        if (diff) return SK_ERR_UNSUPPORTED;
        blocks = (A * (int64_t)rows * (FDp + 2) + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        if constexpr (sizeof(T) != 4) return SK_ERR_UNSUPPORTED;      // (the packed layout is fp32 paths as they are: no fp64 source)
        else {
            SK_LAUNCH((k_prep_paths_packed32<T>), dim3((unsigned)blocks), b, 0, s, X, A, M, D, scale, reinterpret_cast<float *>(out), rows, FDp);
            return check_launch();
        }
    }
    if (diff) {
        if (dim_major) SK_LAUNCH((k_prep_paths<T, true, true>), g, b, 0, s, X, A, M, D, scale, out, rows, FDp);
        else SK_LAUNCH((k_prep_paths<T, true, false>), g, b, 0, s, X, A, M, D, scale, out, rows, FDp);
    } else {
        if (dim_major) SK_LAUNCH((k_prep_paths<T, false, true>), g, b, 0, s, X, A, M, D, scale, out, rows, FDp);
        else SK_LAUNCH((k_prep_paths<T, false, false>), g, b, 0, s, X, A, M, D, scale, out, rows, FDp);
    }
    return check_launch();
}

template int launch_prep_paths<double>(const double *, int64_t, int, int, int, int, double, double *, int, int, hipStream_t);
template int launch_prep_paths<float>(const float *, int64_t, int, int, int, int, double, double *, int, int, hipStream_t);

}  // namespace sk
