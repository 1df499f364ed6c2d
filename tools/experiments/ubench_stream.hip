// ubench_stream.hip — how fast can ONE pass over a 400 MB plane (the 50M store's filter plane; larger than the 256 MB
// Infinity Cache) be read, back to back, by load form?  Decides whether the one-query streaming pass (scan_lazy_kernel) can
// go past the 0.70-0.74 of peak it shares with the trivial grid-stride read.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_stream tools/experiments/ubench_stream.hip && /tmp/ubench_stream
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                               \
        }                                                                           \
    } while (0)

template <bool NT>
__device__ __forceinline__ uint4 ld(const uint4 *p) {
    if (NT) {  // global_load_dwordx4 ... nt
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

// grid-stride, U loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stride(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NT>(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n; i += stride) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}

// every workgroup owns ONE contiguous span (what a tile-per-wave scan does): U x 4 KB in flight per workgroup
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_span(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t acc = 0;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NT>(p + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < hi; i += 256) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}

// LDS-DMA: global -> LDS without a register hop, a ring of R x 4 KB per workgroup, nothing consumed but one word per slot
template <int R, bool NT>
__global__ __launch_bounds__(256) void k_dma(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
    __shared__ uint4 ring[R][256];
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (; i + (R - 1) * stride + 256 <= n; i += R * stride) {
#pragma unroll
        for (int r = 0; r < R; r++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + i + r * stride + threadIdx.x),
                                             (__attribute__((address_space(3))) void *)(&ring[r][wave * 64]), 16, 0, NT ? 2 : 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < R; r++) acc += ring[r][threadIdx.x].x;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the scan's own pattern: a wave reads the 2 KB of filter words inside each of its 4 tiles' 10 KB records (tile-major store)
template <bool NT>
__global__ __launch_bounds__(256) void k_tiles(const uint4 *__restrict__ p, size_t n_tiles, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t tile0 = ((size_t)blockIdx.x * 4 + wave) * 4;
    uint32_t acc = 0;
    uint4 v[8];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const size_t tile = tile0 + t < n_tiles ? tile0 + t : 0;
        const uint4 *src = p + tile * 640 + 4 * 2 * 64 + lane;  // 10 planes-words x 64 lanes per tile; the filter plane's two words
        v[2 * t] = ld<NT>(src);
        v[2 * t + 1] = ld<NT>(src + 64);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    if (acc == 0x12345678u) out[0] = acc;
}

// the same read with pieces of a scan kernel's structure added one by one: MODE 1 = 12 KB of LDS per workgroup (touched),
// 2 = + three barriers, 3 = + a returning ticket atomic per workgroup (64 counters, as finish_rows), 4 = + 60 VGPRs more per lane
template <int MODE>
__global__ __launch_bounds__(256) void k_tiles_plus(const uint4 *__restrict__ p, size_t n_tiles, uint32_t *out, uint32_t *tickets) {
    __shared__ uint4 lds[MODE >= 1 ? 768 : 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t tile0 = ((size_t)blockIdx.x * 4 + wave) * 4;
    uint32_t acc = 0;
    uint4 v[8];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const size_t tile = tile0 + t < n_tiles ? tile0 + t : 0;
        const uint4 *src = p + tile * 640 + 4 * 2 * 64 + lane;
        v[2 * t] = ld<true>(src);
        v[2 * t + 1] = ld<true>(src + 64);
    }
    if (MODE >= 1) lds[threadIdx.x] = make_uint4(lane, wave, 0u, 0u);
    if (MODE >= 2) __syncthreads();
    if (MODE >= 1) acc += lds[(threadIdx.x * 7u) & 255u].x;
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    if (MODE >= 4) {  // register pressure of the real kernel: values that must stay live across the loads
        uint32_t keep[60];
#pragma unroll
        for (int k = 0; k < 60; k++) keep[k] = acc * (k + 3u) + lane;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 60; k++) acc ^= keep[k] >> (k & 7);
    }
    if (MODE >= 2) __syncthreads();
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3 && threadIdx.x == 0) {
        uint32_t *mine = tickets + 32 * (1 + blockIdx.x % 64);
        const uint32_t members = (gridDim.x - blockIdx.x % 64 + 63) / 64;
        if (atomicAdd(mine, 1u) == members - 1) {
            atomicExch(mine, 0u);
            if (atomicAdd(tickets, 1u) == 63u) atomicExch(tickets, 0u);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const size_t bytes = 400001024, n = bytes / 16;
    uint4 *d = nullptr;
    uint32_t *out = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(d, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int cus = prop.multiProcessorCount;
    auto run = [&](const char *name, auto launch) -> int {
        for (int per_cu : {4, 8, 16, 32}) {
            const int grid = cus * per_cu;
            for (int i = 0; i < 5; i++) launch(grid);  // warm-up
            CK(hipEventRecord(e0, nullptr));
            const int passes = 100;
            for (int i = 0; i < passes; i++) launch(grid);  // back to back, like smafa_scan_each
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double gbs = (double)bytes * passes / (ms * 1e-3) / 1e9;
            printf("%-28s %2d WG/CU  %7.2f us/pass  %6.0f GB/s  %.3f of 8 TB/s\n", name, per_cu, ms * 1e3 / passes, gbs, gbs / 8000.0);
        }
        return 0;
    };
#define RUN(name, kern) \
    if (run(name, [&](int grid) { hipLaunchKernelGGL((kern), dim3(grid), dim3(256), 0, nullptr, d, n, out); })) return 1;
    RUN("stride U=8", (k_stride<8, false>))
    RUN("stride U=8 nt", (k_stride<8, true>))
    RUN("stride U=4 nt", (k_stride<4, true>))
    RUN("stride U=16 nt", (k_stride<16, true>))
    RUN("span U=4", (k_span<4, false>))
    RUN("span U=4 nt", (k_span<4, true>))
    RUN("span U=8 nt", (k_span<8, true>))
    RUN("lds-dma ring 4", (k_dma<4, false>))
    RUN("lds-dma ring 4 nt", (k_dma<4, true>))
    RUN("lds-dma ring 8 nt", (k_dma<8, true>))
    {  // tile-major store of 50M subjects: 195 313 tiles x 10 KB = 2 GB, of which 400 MB are read
        const size_t n_tiles = (50000000 + 255) / 256;
        uint4 *big = nullptr;
        CK(hipMalloc(&big, n_tiles * 10240));
        CK(hipMemset(big, 1, n_tiles * 10240));
        const int grid = (int)((n_tiles + 15) / 16);
        for (int nt = 0; nt < 2; nt++) {
            for (int rep = 0; rep < 2; rep++) {
                for (int i = 0; i < 5; i++) {
                    if (nt) hipLaunchKernelGGL((k_tiles<true>), dim3(grid), dim3(256), 0, nullptr, big, n_tiles, out);
                    else hipLaunchKernelGGL((k_tiles<false>), dim3(grid), dim3(256), 0, nullptr, big, n_tiles, out);
                }
                CK(hipEventRecord(e0, nullptr));
                for (int i = 0; i < 100; i++) {
                    if (nt) hipLaunchKernelGGL((k_tiles<true>), dim3(grid), dim3(256), 0, nullptr, big, n_tiles, out);
                    else hipLaunchKernelGGL((k_tiles<false>), dim3(grid), dim3(256), 0, nullptr, big, n_tiles, out);
                }
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double gbs = (double)n_tiles * 2048 * 100 / (ms * 1e-3) / 1e9;
                printf("%-28s one wave per 4 tiles  %7.2f us/pass  %6.0f GB/s  %.3f of 8 TB/s\n", nt ? "2 KB of every 10 KB, nt" : "2 KB of every 10 KB", ms * 1e3 / 100, gbs, gbs / 8000.0);
            }
        }
    }
    {
        const size_t n_tiles = (50000000 + 255) / 256;
        uint4 *big = nullptr;
        uint32_t *tickets = nullptr;
        CK(hipMalloc(&big, n_tiles * 10240));
        CK(hipMemset(big, 1, n_tiles * 10240));
        CK(hipMalloc(&tickets, 65 * 128));
        CK(hipMemset(tickets, 0, 65 * 128));
        const int grid = (int)((n_tiles + 15) / 16);
        auto timeit = [&](const char *name, auto launch) -> int {
            for (int i = 0; i < 5; i++) launch();
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < 100; i++) launch();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double gbs = (double)n_tiles * 2048 * 100 / (ms * 1e-3) / 1e9;
            printf("%-44s %7.2f us/pass  %6.0f GB/s  %.3f of 8 TB/s\n", name, ms * 1e3 / 100, gbs, gbs / 8000.0);
            return 0;
        };
#define PLUS(name, M) \
    if (timeit(name, [&] { hipLaunchKernelGGL((k_tiles_plus<M>), dim3(grid), dim3(256), 0, nullptr, big, n_tiles, out, tickets); })) return 1;
        PLUS("tiles nt, bare", 0)
        PLUS("tiles nt + 12 KB LDS", 1)
        PLUS("tiles nt + LDS + 3 barriers", 2)
        PLUS("tiles nt + LDS + barriers + ticket", 3)
        PLUS("tiles nt + LDS + barriers + ticket + regs", 4)
    }
    return 0;
}
