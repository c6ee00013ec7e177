// Microbenchmark: the solver's block recurrence without any memory traffic, for different block shapes R x S per
// macro-step (same DPP neighbour exchange, same 3-FMA cell), to see how much ILP / per-step overhead is worth.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double dpp_shr1(double v, double fill) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int R, int S, int CR, int CS>   // block R x S fine cells; coefficient granularity CR x CS cells (r x r)
__global__ __launch_bounds__(64) void k(double *out, int steps, double g0) {
    double left[R], bot[S], corner = 1.0;
#pragma unroll
    for (int i = 0; i < R; ++i) left[i] = 1.0;
#pragma unroll
    for (int i = 0; i < S; ++i) bot[i] = 1.0;
    double g = g0 * (threadIdx.x + 1);
    for (int t = 0; t < steps; ++t) {
        double top[S];
#pragma unroll
        for (int i = 0; i < S; ++i) top[i] = dpp_shr1(bot[i], 1.0);
        double ca[R / CR][S / CS], cb[R / CR][S / CS];
#pragma unroll
        for (int k2 = 0; k2 < R / CR; ++k2)
#pragma unroll
            for (int q = 0; q < S / CS; ++q) {
                g = g * 0.999 + 1e-7;   // stand-in for the increment
                const double g2 = g * g;
                ca[k2][q] = fma(g2, 1. / 12, fma(g, 0.5, 1.0));
                cb[k2][q] = fma(g2, -1. / 12, 1.0);
            }
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
            double above = top[cc], diag = cc == 0 ? corner : top[cc - 1];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const double a = ca[rr / CR][cc / CS], b = cb[rr / CR][cc / CS], k10 = left[rr];
                const double v = fma(above, a, fma(k10, a, -(diag * b)));
                diag = k10; above = v; left[rr] = v;
            }
            bot[cc] = above;
        }
        corner = top[S - 1];
    }
    double s = corner;
#pragma unroll
    for (int i = 0; i < R; ++i) s += left[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int R, int S, int CR, int CS> void run(int w) {
    const int blocks = 256 * 4 * w;
    const int steps = 200000 / (R * S) * 16;
    double *d; hipMalloc(&d, sizeof(double) * blocks * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<R, S, CR, CS><<<blocks, 64>>>(d, 10, 1e-6);
    hipEventRecord(e0); k<R, S, CR, CS><<<blocks, 64>>>(d, steps, 1e-6); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cells = (double)steps * R * S;   // per lane
    printf("block %dx%d (coef %dx%d) waves/SIMD=%d : %.2f cycles per cell per wave-slot at 2.1 GHz, %.2f Tcell/s chip\n", R, S, CR, CS, w,
           ms * 1e-3 * 2.1e9 / (cells * w), cells * blocks * 64 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<4, 4, 2, 2>(w); run<4, 8, 2, 2>(w); run<8, 4, 2, 2>(w); run<8, 8, 2, 2>(w); run<4, 16, 2, 2>(w);
    }
    run<4, 8, 4, 4>(2); run<4, 16, 4, 4>(2); run<8, 8, 4, 4>(2);
    return 0;
}
