// fp64 FMA dependent-issue latency: NCH independent chains of v_fma_f64 per wave, 1 or 3 waves per SIMD.
// cycles per instruction per SIMD at ILP 1 = the latency; the ILP at which it stops falling = latency / issue time.
// (build: hipcc -O3 --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b) {
    double v[NCH];
    for (int i = 0; i < NCH; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / NCH; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) v[i] = __builtin_fma(v[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
}
template <int NCH>
void run(int wpc, double *d) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * wpc / 4;
    k<NCH><<<blocks, 256>>>(d, 16, 1.0000001, 1e-9);
    (void)hipEventRecord(e0);
    k<NCH><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_wave = (double)iters * 48;
    printf("%2d waves/CU  %d chains: %.3f ms, %.2f nominal (2.4 GHz) cycles per instruction of ONE wave, %.2f per SIMD\n", wpc, NCH, ms,
           ms * 1e-3 * 2.4e9 / inst_per_wave, ms * 1e-3 * 2.4e9 / (inst_per_wave * wpc / 4));
}
int main() {
    double *d; (void)hipMalloc(&d, 8);
    for (int wpc : {4, 8, 12}) { run<1>(wpc, d); run<2>(wpc, d); run<3>(wpc, d); run<4>(wpc, d); run<6>(wpc, d); run<8>(wpc, d); run<16>(wpc, d); }
    return 0;
}
