// Achievable HBM read bandwidth probe: every workgroup streams a contiguous chunk with 16-byte loads, nothing is written
// (build: hipcc -O3 --offload-arch=gfx950 hbm_read.hip -o hbm_read; run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ src, size_t n_vec, float *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
}
int main() {
    const size_t bytes = 16ull << 30;
    f4 *d; float *sink;
    hipMalloc(&d, bytes); hipMalloc(&sink, 4);
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        k_read<8><<<blocks, 256>>>(d, bytes / 16, sink);
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) k_read<8><<<blocks, 256>>>(d, bytes / 16, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read 16 GiB, %5d workgroups of 256: %.3f ms per pass, %.0f GB/s\n", blocks, ms / 3, bytes / (ms / 3) / 1e6);
    }
    return 0;
}
