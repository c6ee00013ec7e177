// Does the fp64 matrix pipe run beside the fp64 vector pipe?  Per loop iteration a wave issues NF independent v_fma_f64
// (16 accumulators) and NM v_mfma_f64 (4x4x4 4-block, or 16x16x4), separately and interleaved; the report is SIMD cycles
// per wave-iteration at the nominal 2.4 GHz.  (build: hipcc -O3 --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NF, int NM4, int NM16>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b) {
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    double m4[8];
    for (int i = 0; i < 8; ++i) m4[i] = 0.0;
    v4d m16[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double xa = threadIdx.x * 1e-3, xb = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
        constexpr int NMAX = NF > 0 ? NF : (NM4 > NM16 ? NM4 : NM16);
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            if (i < NF) v[i & 15] = __builtin_fma(v[i & 15], a, b);
            if (NF > 0 ? (NM4 > 0 && (i % (NF / (NM4 > 0 ? NM4 : 1))) == 0 && i / (NF / (NM4 > 0 ? NM4 : 1)) < NM4) : i < NM4) {
                const int j = NF > 0 ? i / (NF / (NM4 > 0 ? NM4 : 1)) : i;
                m4[j & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(xa, xb, m4[j & 7], 0, 0, 0);
            }
            if (NF > 0 ? (NM16 > 0 && (i % (NF / (NM16 > 0 ? NM16 : 1))) == 0 && i / (NF / (NM16 > 0 ? NM16 : 1)) < NM16) : i < NM16) {
                const int j = NF > 0 ? i / (NF / (NM16 > 0 ? NM16 : 1)) : i;
                m16[j & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xb, m16[j & 1], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += m4[i];
    s += m16[0][0] + m16[0][1] + m16[0][2] + m16[0][3] + m16[1][0] + m16[1][1] + m16[1][2] + m16[1][3];
    if (s == 12345.678) out[0] = s;
}
template <int NF, int NM4, int NM16>
void run(int wpc, double *d) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wpc / 4;
    k<NF, NM4, NM16><<<blocks, 256>>>(d, 16, 1.0000001, 1e-9);
    hipEventRecord(e0);
    k<NF, NM4, NM16><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_iters_per_simd = (double)blocks * 4 * iters / 1024;
    printf("%2d waves/CU  fma %2d  mfma4x4x4 %2d  mfma16x16x4 %2d : %.3f ms, %.1f cycles per wave-iteration per SIMD (2.4 GHz)\n", wpc, NF, NM4, NM16,
           ms, ms * 1e-3 * 2.4e9 / wave_iters_per_simd);
}
int main() {
    double *d; hipMalloc(&d, 8);
    for (int wpc : {4, 8, 12}) {
        run<32, 0, 0>(wpc, d);    // the 32 increment FMAs of a macro-step
        run<0, 8, 0>(wpc, d);     // the same products on the matrix pipe, 4x4 tiles
        run<0, 0, 2>(wpc, d);     // ... 16x16 tiles
        run<96, 0, 0>(wpc, d);    // a whole macro-step's fp64 work without the increments (64) and with (96)
        run<64, 0, 0>(wpc, d);
        run<64, 8, 0>(wpc, d);    // 64 vector FMAs + the increments on the matrix pipe
        run<64, 0, 2>(wpc, d);
        run<32, 8, 0>(wpc, d);
        run<32, 0, 2>(wpc, d);
    }
    return 0;
}
