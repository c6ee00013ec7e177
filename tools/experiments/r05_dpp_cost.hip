// What does a DPP move cost on gfx950, by control?  (round 5: handing two more registers down the lanes with `wave_shr:1` made the
// edge-keeping forward slower although it removed 6 % of its VALU instructions.)  96 dependent-chain moves per iteration, 8 chains,
// four waves per SIMD; KIND 0: v_mov_b32, 1: wave_shr:1, 2: row_shr:1, 3: row_bcast:15 (row_mask 0xe), 4: v_fma_f64,
// 5: row_shr:1 + row_bcast:15 pair (a wave shift built from row-level controls), 6: quad_perm, 7: 48 v_fma_f64 + 48 wave_shr:1 interleaved,
// 8: 48 v_fma_f64 + 48 row_shr:1 interleaved
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, int iter, double s) {
    int a[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 3 + i; d[i] = threadIdx.x * 1e-9 + i; }
    for (int it = 0; it < iter; ++it) {
#pragma unroll
        for (int v = 0; v < 96; ++v) {
            int &x = a[v & 7];
            if (KIND == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(x));
            if (KIND == 1) x = __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
            if (KIND == 2) x = __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false);
            if (KIND == 3) x = __builtin_amdgcn_update_dpp(x, x, 0x142, 0xe, 0xf, false);
            if (KIND == 4) d[v & 7] = __builtin_fma(d[v & 7], s, 0.25);
            if (KIND == 5) { if (v & 1) x = __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false); else x = __builtin_amdgcn_update_dpp(x, x, 0x142, 0xe, 0x1, false); }
            if (KIND == 6) x = __builtin_amdgcn_update_dpp(x, x, 0x4e, 0xf, 0xf, false);
            if (KIND == 7) { if (v & 1) x = __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); else d[v & 7] = __builtin_fma(d[v & 7], s, 0.25); }
            if (KIND == 8) { if (v & 1) x = __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false); else d[v & 7] = __builtin_fma(d[v & 7], s, 0.25); }
        }
    }
    double r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, double *out) {
    const int iter = 20000, blocks = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 100, 0.999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iter, 0.999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.1f ns per iteration of 96 instructions\n", name, ms * 1e6 / iter);
}

int main() {
    double *out;
    hipMalloc(&out, 1024 * 256 * sizeof(double));
    run<4>("v_fma_f64", out);
    run<0>("v_mov_b32", out);
    run<1>("v_mov_b32_dpp wave_shr:1", out);
    run<2>("v_mov_b32_dpp row_shr:1", out);
    run<3>("v_mov_b32_dpp row_bcast:15 row_mask:0xe", out);
    run<5>("row_bcast:15 bank_mask:0x1 + row_shr:1 (48 pairs)", out);
    run<6>("v_mov_b32_dpp quad_perm", out);
    run<7>("48 v_fma_f64 + 48 wave_shr:1", out);
    run<8>("48 v_fma_f64 + 48 row_shr:1", out);
    return 0;
}
