// Does the fp64 matrix pipe of gfx950 run beside the fp64 vector pipe?  (round 5: the headline kernel spends 32 of its 92 fp64
// instructions per macro-step on the increments <dx, dy>, a GEMM; if v_mfma_f64 issues in the shadow of v_fma_f64 the increments
// could move there.)  Each wave runs ITER iterations of NV dependent-chain v_fma_f64 (8 chains) and NM v_mfma_f64 (KIND: 0 none,
// 1 = 16x16x4, 2 = 4x4x4 4 blocks); 4 waves per SIMD.  Prints ns per iteration per wave-slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV, int NM, int KIND>
__global__ __launch_bounds__(256) void k(double *out, int iter, double s) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-9 + i;
    double m = s * 0.5, n = s * 0.25;
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double acc1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iter; ++it) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            a[v & 7] = __builtin_fma(a[v & 7], s, n);
            if (NM > 0 && (v % (NV / (NM > 0 ? NM : 1) > 0 ? NV / (NM > 0 ? NM : 1) : 1)) == 0 && v / (NV / NM > 0 ? NV / NM : 1) < NM) {
                const int j = v / (NV / NM > 0 ? NV / NM : 1);
                if (KIND == 1) acc[j & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(m, n, acc[j & 3], 0, 0, 0);
                if (KIND == 2) acc1[j & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(m, n, acc1[j & 7], 0, 0, 0);
            }
        }
        if (NV == 0) {
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                if (KIND == 1) acc[j & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(m, n, acc[j & 3], 0, 0, 0);
                if (KIND == 2) acc1[j & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(m, n, acc1[j & 7], 0, 0, 0);
            }
        }
    }
    double r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + acc1[i];
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NV, int NM, int KIND>
static void run(const char *name, double *out, int blocks) {
    const int iter = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NV, NM, KIND><<<blocks, 256>>>(out, 100, 0.999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, NM, KIND><<<blocks, 256>>>(out, iter, 0.999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s blocks %5d: %8.1f ns per iteration  (%.1f us total)\n", name, blocks, ms * 1e6 / iter, ms * 1e3);
}

int main() {
    double *out;
    hipMalloc(&out, 8192 * 256 * sizeof(double));
    for (int blocks : {256, 1024}) {      // 1 / 4 waves per SIMD
        run<125, 0, 0>("125 v_fma_f64", out, blocks);
        run<93, 0, 0>("93 v_fma_f64", out, blocks);
        run<93, 2, 1>("93 v_fma_f64 + 2 mfma_f64_16x16x4", out, blocks);
        run<93, 4, 1>("93 v_fma_f64 + 4 mfma_f64_16x16x4", out, blocks);
        run<93, 8, 2>("93 v_fma_f64 + 8 mfma_f64_4x4x4", out, blocks);
        run<0, 2, 1>("2 mfma_f64_16x16x4", out, blocks);
        run<0, 8, 1>("8 mfma_f64_16x16x4", out, blocks);
        run<0, 8, 2>("8 mfma_f64_4x4x4", out, blocks);
    }
    return 0;
}
