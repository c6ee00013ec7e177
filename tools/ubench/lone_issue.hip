// What does ONE wave per SIMD pay per instruction?  Per iteration a wave issues NF v_fma_f64 (16 chains), NI 32-bit VALU
// (v_add_u32, 8 chains), NS scalar ALU instructions (s_add_u32, 4 chains) and NB taken scalar branches, at 1 / 2 / 3 waves per SIMD.
// If the scalar stream were hidden behind the vector stream of the SAME wave, adding NS would cost nothing at one wave per SIMD.
// (build: hipcc -O3 --offload-arch=gfx950 lone_issue.hip -o lone_issue)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NF, int NI, int NS, int NB>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b, unsigned m) {
    double v[16];
    unsigned y[8];
    unsigned s0 = m, s1 = m + 1, s2 = m + 2, s3 = m + 3;
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    for (int i = 0; i < 8; ++i) y[i] = threadIdx.x * 7u + i;
    for (int it = 0; it < iters; ++it) {
        constexpr int N = NF > NI ? (NF > NS ? NF : NS) : (NI > NS ? NI : NS);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i < NF) v[i & 15] = __builtin_fma(v[i & 15], a, b);
            if (NI > 0 && (i * NI) / N != ((i + 1) * NI) / N) {
                unsigned t = y[i & 7];
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(m));
                y[i & 7] = t;
            }
            if (NS > 0 && (i * NS) / N != ((i + 1) * NS) / N) {
                if ((i & 3) == 0) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
                if ((i & 3) == 1) asm volatile("s_add_u32 %0, %0, 5" : "+s"(s1) : : "scc");
                if ((i & 3) == 2) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s2) : : "scc");
                if ((i & 3) == 3) asm volatile("s_add_u32 %0, %0, 9" : "+s"(s3) : : "scc");
            }
            if (NB > 0 && (i * NB) / N != ((i + 1) * NB) / N) {
                // a taken branch over one instruction (s_cmp sets scc = 1: s0 == s0)
                asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n1:" : "+s"(s0) : : "scc");
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += y[i];
    s += s0 + s1 + s2 + s3;
    if (s == 12345.678) out[0] = s;
}
template <int NF, int NI, int NS, int NB>
void run(int wpc, double *d) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * wpc / 4;
    for (int w = 0; w < 3; ++w) k<NF, NI, NS, NB><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9, 3u);
    (void)hipEventRecord(e0);
    k<NF, NI, NS, NB><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9, 3u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.1e9 / iters;   // cycles (at the ~2.1 GHz the chip sustains) per iteration of the whole SIMD
    printf("%2d waves/SIMD  fma_f64 %3d  v_add_u32 %3d  s_add_u32 %3d  taken branches %2d : %.3f ms, %7.1f cycles per iteration and SIMD, %5.2f per instruction of one wave\n",
           wpc / 4, NF, NI, NS, NB, ms, cyc, cyc / (wpc / 4) / (NF + NI + NS + 2 * NB));
}
int main() {
    double *d; (void)hipMalloc(&d, 8);
    for (int wpc : {4, 8, 12}) {
        run<96, 0, 0, 0>(wpc, d);
        run<0, 96, 0, 0>(wpc, d);
        run<0, 0, 96, 0>(wpc, d);
        run<96, 48, 0, 0>(wpc, d);
        run<96, 0, 48, 0>(wpc, d);
        run<96, 0, 96, 0>(wpc, d);
        run<96, 48, 48, 0>(wpc, d);
        run<96, 48, 48, 8>(wpc, d);
        run<96, 48, 48, 16>(wpc, d);
    }
    return 0;
}
