// fp64 VALU issue-rate probe: waves of independent v_fma_f64 / v_mul_f64 / v_add_f64 chains (build: hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b) {
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) v[i] = __builtin_fma(v[i], a, b);
                else if (OP == 1) v[i] = v[i] * a;
                else if (OP == 2) v[i] = v[i] + b;
                else if (OP == 3) v[i] = __builtin_rint(v[i]);
                else if (OP == 4) v[i] = __builtin_ldexp(v[i], (int)threadIdx.x & 1);
                else if (OP == 5) v[i] = (double)((int)v[i]);          // cvt_i32_f64 + cvt_f64_i32
                else if (OP == 7) v[i] = (double)((float)v[i] * 1.0000001f);   // cvt_f32_f64 + mul_f32 + cvt_f64_f32
                else if (OP == 8) { float f = __builtin_bit_cast(float, (int)(__builtin_bit_cast(long long, v[i]) >> 32)); v[i] = (double)f + b; }   // cvt_f64_f32 + add
                else if (OP == 9) { int lo = __builtin_amdgcn_update_dpp(0, (int)__builtin_bit_cast(long long, v[i]), 0x138, 0xf, 0xf, false); v[i] = __builtin_bit_cast(double, ((long long)lo << 32) | (unsigned)lo); }   // 1 dpp mov (+ pack)
                else v[i] = v[i] - b;
            }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
}
template <int OP>
void run(const char *name, int wpc, double *d) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wpc / 4;
    k<OP><<<blocks, 256>>>(d, 16, 1.0000001, 1e-9);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters * 64;   // wave instructions
    printf("%-8s %2d waves/CU: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz, %.1f T lane-ops/s\n", name, wpc, ms,
           ms * 1e-3 * 2.4e9 / (inst / 1024), inst * 64 / ms / 1e9);
}
int main() {
    double *d; hipMalloc(&d, 8);
    for (int wpc : {4, 8, 12}) {
        run<0>("fma_f64", wpc, d); run<1>("mul_f64", wpc, d); run<2>("add_f64", wpc, d); run<3>("rndne_f64", wpc, d);
        run<4>("ldexp_f64", wpc, d); run<5>("cvt_pair", wpc, d); run<7>("f64->f32->f64", wpc, d); run<8>("cvt_f64_f32+add", wpc, d);
        run<9>("dpp_mov", wpc, d);
    }
    return 0;
}
