// sk_increments.hip -- static Gram -> coarse increments, and its transpose.
//
// inc_c = G[1:,1:] + G[:-1,:-1] - G[1:,:-1] - G[:-1,1:]  (sigkernel.py:217, :363), evaluated left
// to right like the reference's chain of torch ops, in ONE pass: G is read once (row i+1 of a
// strip is carried in registers to become row i of the next output row) and inc_c written once.
// The reference then materialises the 4^d-times larger refined tensor with tile() (:218, :364);
// here refinement is index arithmetic inside the solver kernels.
#include "sk_internal.h"

namespace sk {
namespace {

constexpr int TPB = 256;
constexpr int ROWS = 16;  // output rows per block strip

template <typename T>
__global__ __launch_bounds__(TPB) void k_increments(const T *__restrict__ G, int M, int N, int strips,
                                                    T *__restrict__ inc, int64_t ld) {
    const int Mc = M - 1, Nc = N - 1;
    const int64_t p = blockIdx.x / strips;
    const int i0 = (int)(blockIdx.x % strips) * ROWS;
    const int i1 = min(i0 + ROWS, Mc);
    const T *g = G + p * (int64_t)M * N;
    T *o = inc + p * (int64_t)Mc * ld;
    for (int j = Nc + threadIdx.x; j < ld; j += TPB)   // zero the row padding
        for (int i = i0; i < i1; ++i) o[(int64_t)i * ld + j] = (T)0;
    for (int j = threadIdx.x; j < Nc; j += TPB) {
        T a0 = g[(int64_t)i0 * N + j], a1 = g[(int64_t)i0 * N + j + 1];  // row i:   G[i][j], G[i][j+1]
        for (int i = i0; i < i1; ++i) {
            const T b0 = g[(int64_t)(i + 1) * N + j], b1 = g[(int64_t)(i + 1) * N + j + 1];
            o[(int64_t)i * ld + j] = ((b1 + a0) - b0) - a1;
            a0 = b0; a1 = b1;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(TPB) void k_increments_adjoint(const T *__restrict__ W, int64_t ldw,
                                                            const T *__restrict__ scale, int M, int N, int strips,
                                                            T *__restrict__ dG) {
    const int Mc = M - 1, Nc = N - 1;
    const int64_t p = blockIdx.x / strips;
    const int m0 = (int)(blockIdx.x % strips) * ROWS;
    const int m1 = min(m0 + ROWS, M);
    const T *w = W + p * (int64_t)Mc * ldw;
    T *o = dG + p * (int64_t)M * N;
    const T s = scale ? scale[p] : (T)1;
    for (int n = threadIdx.x; n < N; n += TPB) {
        const bool hl = n >= 1, hr = n < Nc;  // W columns n-1 / n exist
        // row m-1 of W (zero above the first row)
        T u0 = (m0 >= 1 && hl) ? w[(int64_t)(m0 - 1) * ldw + n - 1] : (T)0;
        T u1 = (m0 >= 1 && hr) ? w[(int64_t)(m0 - 1) * ldw + n] : (T)0;
        for (int m = m0; m < m1; ++m) {
            const T v0 = (m < Mc && hl) ? w[(int64_t)m * ldw + n - 1] : (T)0;
            const T v1 = (m < Mc && hr) ? w[(int64_t)m * ldw + n] : (T)0;
            o[(int64_t)m * N + n] = s * (((u0 + v1) - u1) - v0);
            u0 = v0; u1 = v1;
        }
    }
}

}  // namespace

template <typename T>
int launch_increments(const T *G, int64_t P, int M, int N, T *inc_c, int64_t ld, hipStream_t s) {
    const int strips = (M - 1 + ROWS - 1) / ROWS;
    const int64_t blocks = P * strips;
    if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_increments<T>, dim3((unsigned)blocks), dim3(TPB), 0, s, G, M, N, strips, inc_c, ld);
    return check_launch();
}

template <typename T>
int launch_increments_adjoint(const T *W, int64_t ldw, const T *scale, int64_t P, int M, int N, T *dG, hipStream_t s) {
    const int strips = (M + ROWS - 1) / ROWS;
    const int64_t blocks = P * strips;
    if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_increments_adjoint<T>, dim3((unsigned)blocks), dim3(TPB), 0, s, W, ldw, scale, M, N, strips, dG);
    return check_launch();
}

template int launch_increments<double>(const double *, int64_t, int, int, double *, int64_t, hipStream_t);
template int launch_increments<float>(const float *, int64_t, int, int, float *, int64_t, hipStream_t);
template int launch_increments_adjoint<double>(const double *, int64_t, const double *, int64_t, int, int, double *,
                                               hipStream_t);
template int launch_increments_adjoint<float>(const float *, int64_t, const float *, int64_t, int, int, float *, hipStream_t);

}  // namespace sk
