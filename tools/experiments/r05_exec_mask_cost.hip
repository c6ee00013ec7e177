// What does a VALU instruction cost when only some lanes are active?  (round 5: the edge-keeping forward runs ~60 VALU instructions per
// macro-step in branches that one or two lanes of 64 take; handing the pair's edge block down the lanes by DPP instead removed 6 % of
// the kernel's VALU instructions -- PMC -- and made it 2.4 % SLOWER.)  NV dependent-chain v_fma_f64 / v_add_u32 per iteration with
// lanes [0, n) active, four waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, int iter, double s, int nact) {
    const int lane = threadIdx.x & 63;
    double a[8];
    int b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-9 + i; b[i] = threadIdx.x + i; }
    if (KIND != 2 && (nact >= 0 ? lane < nact : lane >= 64 + nact)) {      // nact < 0: the LAST -nact lanes
        for (int it = 0; it < iter; ++it) {
#pragma unroll
            for (int v = 0; v < 96; ++v) {
                if (KIND == 0) a[v & 7] = __builtin_fma(a[v & 7], s, 0.25);
                else { b[v & 7] = b[v & 7] * 3 + it; asm volatile("" : "+v"(b[v & 7])); }
            }
        }
    }
    if (KIND == 2) {      // 96 instructions on all lanes + 96 on lanes [0, nact) per iteration: a busy kernel with a lane-sparse branch
        double c[8];
        for (int i = 0; i < 8; ++i) c[i] = a[i] + 1.0;
        for (int it = 0; it < iter; ++it) {
#pragma unroll
            for (int v = 0; v < 96; ++v) a[v & 7] = __builtin_fma(a[v & 7], s, 0.25);
            if (lane < nact) {
                asm volatile("");
#pragma unroll
                for (int v = 0; v < 96; ++v) c[v & 7] = __builtin_fma(c[v & 7], s, 0.5);
            }
        }
        for (int i = 0; i < 8; ++i) a[i] += c[i];
    }
    double r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + b[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, double *out, int nact) {
    const int iter = 20000, blocks = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 100, 0.999, nact);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iter, 0.999, nact);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s lanes [0, %2d) active: %8.1f ns per iteration of 96 instructions\n", name, nact, ms * 1e6 / iter);
}

int main() {
    double *out;
    hipMalloc(&out, 1024 * 256 * sizeof(double));
    for (int n : {64, 32, 16, 12, 8, 6, 5, 4, 3, 2, 1, -1, -2, -8, -16}) run<0>("v_fma_f64", out, n);
    for (int n : {64, 16, 8, 4, 3, 2, 1}) run<1>("v_mad_u32 (int)", out, n);
    for (int n : {0, 1, 2, 8, 16, 32, 64}) run<2>("96 dense + 96 v_fma_f64", out, n);
    return 0;
}
