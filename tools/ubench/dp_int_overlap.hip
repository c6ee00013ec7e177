// Do integer VALU instructions cost fp64 issue time?  Per iteration a wave issues 96 v_fma_f64 (16 chains) and NI integer
// instructions (v_mad_u32_u24 chains the compiler cannot fold), at 1 / 2 / 3 waves per SIMD.
// (build: hipcc -O3 --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NF, int NI, int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b, unsigned m) {
    double v[16];
    unsigned y[8];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    for (int i = 0; i < 8; ++i) y[i] = threadIdx.x * 7u + i;
    for (int it = 0; it < iters; ++it) {
        constexpr int N = NF > NI ? NF : NI;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i < NF) v[i & 15] = __builtin_fma(v[i & 15], a, b);
            if (NI > 0 && (i * NI) / N != ((i + 1) * NI) / N) {       // NI of them, spread evenly
                unsigned t = y[i & 7];
                if (OP == 0) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(t) : "v"(m));
                if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(m));
                if (OP == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t) : "v"(m) : "vcc");
                if (OP == 3) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(y[(i + 1) & 7]));
                if (OP == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(y[(i + 1) & 7]));
                y[i & 7] = t;
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += y[i];
    if (s == 12345.678) out[0] = s;
}
template <int NF, int NI, int OP>
void run(int wpc, double *d, const char *name) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * wpc / 4;
    for (int w = 0; w < 3; ++w) k<NF, NI, OP><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9, 3u);
    (void)hipEventRecord(e0);
    k<NF, NI, OP><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9, 3u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%2d waves/CU  fma_f64 %3d  %-14s %3d : %.3f ms, %.1f nominal (2.4 GHz) cycles per wave-iteration per SIMD\n", wpc, NF, name, NI, ms,
           ms * 1e-3 * 2.4e9 / ((double)blocks * 4 * iters / 1024));
}
int main() {
    double *d; (void)hipMalloc(&d, 8);
    for (int wpc : {4, 12}) {
        run<96, 0, 0>(wpc, d, "-");
        run<96, 32, 0>(wpc, d, "v_mad_u32_u24"); run<0, 96, 0>(wpc, d, "v_mad_u32_u24");
        run<96, 32, 1>(wpc, d, "v_add_u32"); run<0, 96, 1>(wpc, d, "v_add_u32");
        run<96, 32, 2>(wpc, d, "v_cndmask_b32"); run<0, 96, 2>(wpc, d, "v_cndmask_b32");
        run<96, 32, 3>(wpc, d, "v_mov_b32_dpp"); run<0, 96, 3>(wpc, d, "v_mov_b32_dpp");
        run<96, 32, 4>(wpc, d, "v_mov_b32"); run<0, 96, 4>(wpc, d, "v_mov_b32");
    }
    return 0;
}
