// Microbenchmark: issue rate and dependent latency of v_fma_f64 (and a few neighbours) on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/fp64_rate.hip -o /tmp/fp64_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS, int OP>
__global__ __launch_bounds__(64) void k(double *out, int iters, double a, double b) {
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) x[c] = fma(x[c], a, b);              // v_fma_f64
                if (OP == 1) x[c] = x[c] * a;                     // v_mul_f64
                if (OP == 2) x[c] = x[c] + a;                     // v_add_f64
                if (OP == 3) { float f = (float)x[c]; f = fmaf(f, (float)a, (float)b); x[c] = f; }
            }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS, int OP>
void run(const char *name, int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd, iters = 4000;
    double *d;
    hipMalloc(&d, sizeof(double) * blocks * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS, OP><<<blocks, 64>>>(d, 10, 1.0000001, 1e-9);
    hipEventRecord(e0);
    k<CHAINS, OP><<<blocks, 64>>>(d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = (double)iters * 8 * CHAINS;
    const double ns_per_inst_per_simd = ms * 1e6 / (insts_per_wave * waves_per_simd);
    printf("%-10s chains=%2d waves/SIMD=%d : %.3f ms, %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.1 GHz), %.1f Tinstr-lanes/s\n",
           name, CHAINS, waves_per_simd, ms, ns_per_inst_per_simd, ns_per_inst_per_simd * 2.1,
           insts_per_wave * blocks * 64 / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<1, 0>("fma_f64", w);
        run<2, 0>("fma_f64", w);
        run<4, 0>("fma_f64", w);
        run<8, 0>("fma_f64", w);
    }
    run<8, 1>("mul_f64", 2);
    run<8, 2>("add_f64", 2);
    run<1, 2>("add_f64", 1);
    return 0;
}
