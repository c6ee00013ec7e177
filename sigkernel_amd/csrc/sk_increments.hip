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

// Increments of the static kernel and of its first / second finite-difference derivative along gamma, from the
// static Gram matrices of X, X + eps*gamma, X + 2*eps*gamma (sigkernel.py:526-541), in the reference's operand
// order: the Gram matrices are scaled first (-(1/eps)*G0, (1/eps)*G1, -(1/eps)*(-(1/eps)*G0), -(2/eps)*((1/eps)*G1),
// (1/eps^2)*G2), each is 4-corner differenced, and the differences are added left to right.  The sums cancel
// ~1/eps^2 = 1e8 of magnitude, so the order matters at the 1e-8 level: no FMA contraction here.
template <typename T>
__global__ __launch_bounds__(TPB) void k_deriv_increments(const T *__restrict__ G0, const T *__restrict__ G1,
                                                          const T *__restrict__ G2, T c1, T c2, T c3, int M, int N,
                                                          int strips, T *__restrict__ inc, T *__restrict__ inc_d,
                                                          T *__restrict__ inc_dd, int64_t ld) {
#pragma clang fp contract(off)
    const int Mc = M - 1, Nc = N - 1;
    const int64_t p = blockIdx.x / strips;
    const int i0 = (int)(blockIdx.x % strips) * ROWS;
    const int i1 = min(i0 + ROWS, Mc);
    const int64_t gi = p * (int64_t)M * N, oi = p * (int64_t)Mc * ld;
    for (int j = Nc + threadIdx.x; j < ld; j += TPB)
        for (int i = i0; i < i1; ++i) {
            inc[oi + (int64_t)i * ld + j] = (T)0;
            inc_d[oi + (int64_t)i * ld + j] = (T)0;
            inc_dd[oi + (int64_t)i * ld + j] = (T)0;
        }
    // v[0..4]: G0, d1 = -c1*G0, d2 = c1*G1, dd1 = -c1*d1, dd2 = -c2*d2, dd3 = c3*G2   (c1 = 1/eps, c2 = 2/eps, c3 = 1/eps^2)
    auto load = [&](int64_t o, T (&v)[6]) {
        const T g0 = G0[gi + o], g1 = G1[gi + o], g2 = G2[gi + o];
        v[0] = g0;
        v[1] = -c1 * g0;
        v[2] = c1 * g1;
        v[3] = -c1 * v[1];
        v[4] = -c2 * v[2];
        v[5] = c3 * g2;
    };
    for (int j = threadIdx.x; j < Nc; j += TPB) {
        T a0[6], a1[6], b0[6], b1[6], c[6];
        load((int64_t)i0 * N + j, a0);
        load((int64_t)i0 * N + j + 1, a1);
        for (int i = i0; i < i1; ++i) {
            load((int64_t)(i + 1) * N + j, b0);
            load((int64_t)(i + 1) * N + j + 1, b1);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                c[k] = ((b1[k] + a0[k]) - b0[k]) - a1[k];
                a0[k] = b0[k];
                a1[k] = b1[k];
            }
            inc[oi + (int64_t)i * ld + j] = c[0];
            inc_d[oi + (int64_t)i * ld + j] = c[1] + c[2];
            inc_dd[oi + (int64_t)i * ld + j] = (c[3] + c[4]) + c[5];
        }
    }
}

}  // namespace

template <typename T>
int launch_increments(const T *G, int64_t P, int M, int N, T *inc_c, int64_t ld, hipStream_t s) {
    const int strips = (M - 1 + ROWS - 1) / ROWS;
    const int64_t blocks = P * strips;
    if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    SK_LAUNCH(k_increments<T>, dim3((unsigned)blocks), dim3(TPB), 0, s, G, M, N, strips, inc_c, ld);
    return check_launch();
}

template <typename T>
int launch_increments_adjoint(const T *W, int64_t ldw, const T *scale, int64_t P, int M, int N, T *dG, hipStream_t s) {
    const int strips = (M + ROWS - 1) / ROWS;
    const int64_t blocks = P * strips;
    if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    SK_LAUNCH(k_increments_adjoint<T>, dim3((unsigned)blocks), dim3(TPB), 0, s, W, ldw, scale, M, N, strips, dG);
    return check_launch();
}

template int launch_increments<double>(const double *, int64_t, int, int, double *, int64_t, hipStream_t);
template int launch_increments<float>(const float *, int64_t, int, int, float *, int64_t, hipStream_t);
template int launch_increments_adjoint<double>(const double *, int64_t, const double *, int64_t, int, int, double *,
                                               hipStream_t);
template int launch_increments_adjoint<float>(const float *, int64_t, const float *, int64_t, int, int, float *, hipStream_t);

template <typename T>
int launch_deriv_increments(const T *G0, const T *G1, const T *G2, double eps, int64_t P, int M, int N, T *inc, T *inc_d,
                            T *inc_dd, int64_t ld, hipStream_t s) {
    const int strips = (M - 1 + ROWS - 1) / ROWS;
    const int64_t blocks = P * strips;
    if (blocks > 0x7fffffffLL) return SK_ERR_UNSUPPORTED;
    // the python scalars of sigkernel.py:529-539, rounded to T when they meet the tensor like torch does
    const T c1 = (T)(1. / eps), c2 = (T)(2. / eps), c3 = (T)(1. / (eps * eps));
    SK_LAUNCH(k_deriv_increments<T>, dim3((unsigned)blocks), dim3(TPB), 0, s, G0, G1, G2, c1, c2, c3, M, N, strips,
                       inc, inc_d, inc_dd, ld);
    return check_launch();
}

template int launch_deriv_increments<double>(const double *, const double *, const double *, double, int64_t, int, int,
                                             double *, double *, double *, int64_t, hipStream_t);
template int launch_deriv_increments<float>(const float *, const float *, const float *, double, int64_t, int, int, float *,
                                            float *, float *, int64_t, hipStream_t);

}  // namespace sk
