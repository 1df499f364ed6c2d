// ubench_hbm.hip — empirical HBM read-stream ceiling: a trivial sum over a buffer far larger than the
// 256 MiB Infinity Cache, 16-byte loads per lane, grid-stride.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o /tmp/ubench_hbm && /tmp/ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void sum_kernel(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n; i += stride) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;  // keeps the loads alive
}

int main() {
    const size_t bytes = 8ull << 30;  // 8 GiB
    uint4 *d;
    uint32_t *out;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(d, 1, bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int per_cu : {4, 8, 16, 32}) {
        const int grid = prop.multiProcessorCount * per_cu;
        hipLaunchKernelGGL(sum_kernel<8>, dim3(grid), dim3(256), 0, 0, d, bytes / 16, out);
        CHECK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(sum_kernel<8>, dim3(grid), dim3(256), 0, 0, d, bytes / 16, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("read 8 GiB, %2d workgroups/CU: best %.3f ms -> %.0f GB/s (%.1f %% of 8000)\n", per_cu, best,
               bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9 / 80.0);
    }
    return 0;
}
