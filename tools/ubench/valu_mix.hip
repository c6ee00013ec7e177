// Microbenchmark: issue cost of the non-fp64 VALU instructions the solver kernels use, alone and mixed with fp64 FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(64) void k(double *out, int iters, double a, double b, int sel) {
    double x[4]; int y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[c] = threadIdx.x * 1e-3 + c; y[c] = threadIdx.x + c; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (OP == 0) { y[c] = y[c] * 3 + sel; }                               // v_mad / v_mul_lo + add
                if (OP == 1) { y[c] = (y[c] > sel) ? y[c] + 1 : y[c] - 3; }            // cmp + cndmask-ish
                if (OP == 2) { y[c] = __builtin_amdgcn_update_dpp(0, y[c], 0x138, 0xf, 0xf, false) + 1; }  // dpp mov + add
                if (OP == 3) { x[c] = fma(x[c], a, b); y[c] = y[c] + sel; }            // fp64 fma + int add interleaved
                if (OP == 4) { y[c] = y[c] + sel; }                                    // v_add_u32
                if (OP == 5) { x[c] = (y[c] & 1) ? x[c] : a; y[c] += 1; }              // 64-bit select (2 cndmask) + add
            }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += x[c] + y[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int n_inst_per_rep) {
    const int w = 2, blocks = 256 * 4 * w, iters = 4000;
    double *d; hipMalloc(&d, sizeof(double) * blocks * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 64>>>(d, 10, 1.0000001, 1e-9, 3);
    hipEventRecord(e0); k<OP><<<blocks, 64>>>(d, iters, 1.0000001, 1e-9, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 8 * 4;   // per wave
    printf("%-28s : %.3f ms, %.2f ns per op-group per SIMD (2 waves/SIMD) = %.1f cycles at 2.1 GHz (%d instr/group nominal)\n", name, ms,
           ms * 1e6 / (groups * w), ms * 1e6 / (groups * w) * 2.1, n_inst_per_rep);
    hipFree(d);
}
int main() {
    run<4>("v_add_u32", 1); run<0>("mul_lo+add", 2); run<1>("cmp+2 add+cndmask", 4); run<2>("dpp mov + add", 2);
    run<3>("fma_f64 + v_add_u32", 2); run<5>("and+cmp+2 cndmask+add", 5);
    return 0;
}
