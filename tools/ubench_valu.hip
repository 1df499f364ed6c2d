// ubench_valu.hip — issue-rate microbenchmark for the VALU instructions of the scan kernel on gfx950.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
// Prints, per instruction mix and occupancy, lane-ops/s for the whole chip and cycles per wave-instruction
// per SIMD at the clock measured inside the kernel (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4000;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *clk, uint32_t q_in) {
    uint32_t a[8], s[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i; s[i] = threadIdx.x * 13 + i * 5; }
    uint32_t q = __builtin_amdgcn_readfirstlane(q_in);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) {  // v_xor_b32 VOP2 vgpr,vgpr
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 1) {  // v_bitop3 v,v,s
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a[i]) : "v"(s[i]), "s"(q));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 2) {  // v_bitop3 v,v,v
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 3) {  // v_bcnt
#define X(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(s[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 4) {  // v_xor VOP2 with sgpr src0
#define X(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(q));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 5) {  // the scan mix: 5 planes: xor + 4 bitop3, then bcnt, per 8 chains (=32 instr: 24 + 8 padded)
#define X(i) asm volatile("v_xor_b32 %0, %2, %1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6\n v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s[i]), "s"(q));
            REP8(X)
#undef X
        } else if (KIND == 6) {  // v_or3 v,v,v (VOP3, 3 vgpr)
#define X(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 8) {  // v_bcnt v, 0
#define X(i) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(a[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 9) {  // v_min_u32 VOP2
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 10) {  // v_min3_u32
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 11) {  // v_cmp_le_u32 -> vcc
#define X(i) asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(a[i]), "v"(s[i]) : "vcc");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 12) {  // v_add_u32 VOP2
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 13) {  // v_mov_b32 v, s
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(q));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 14) {  // filter mix per subject: xor, bitop3, bcnt(v,0)  x8 + 8 min
#define X(i) asm volatile("v_xor_b32 %0, %0, %1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6\n v_bcnt_u32_b32 %0, %0, 0\n v_min_u32 %0, %0, %2" : "+v"(a[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
            REP8(X)
#undef X
        } else if (KIND == 15) {  // v_xor VOP2 v,v interleaved with v_bcnt 3:1
#define X(i) asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_bcnt_u32_b32 %0, %0, 0" : "+v"(a[i]) : "v"(s[i]));
            REP8(X)
#undef X
        } else if (KIND == 7) {  // v_and_or VOP3 v,v,s
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "s"(q));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        clk[w * 2] = t1 - t0;
        clk[w * 2 + 1] = r1 - r0;
    }
}

template <int KIND>
int run(const char *name, int per_cu, int n_cu, uint32_t *d_out, unsigned long long *d_clk) {
    const int grid = n_cu * per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> clk((size_t)grid * 8);
    CHECK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int w = 0; w < grid * 4; w++) { cyc += clk[w * 2]; real += clk[w * 2 + 1]; }
    const double ghz = cyc / real * 0.1;  // s_memrealtime ticks at 100 MHz
    const double instr_per_wave = (double)ITERS * 32;
    const double waves_per_simd = per_cu;  // 4 waves per block, 4 SIMDs per CU
    const double cyc_per_wave = cyc / (grid * 4);
    const double cyc_per_instr_simd = cyc_per_wave / (instr_per_wave * waves_per_simd);
    const double lane_ops = (double)grid * 256 * instr_per_wave / (ms * 1e-3);
    printf("%-28s waves/SIMD=%d  %8.3f ms  %6.2f T lane-ops/s  clock %.2f GHz  %.2f cyc/instr/SIMD\n", name, per_cu, ms,
           lane_ops / 1e12, ghz, cyc_per_instr_simd);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("%s  CUs=%d  clockRate=%d kHz\n", p.name, n_cu, p.clockRate);
    uint32_t *d_out;
    unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
    CHECK(hipMalloc(&d_clk, (size_t)n_cu * 8 * 4 * 16));
    for (int per_cu : {2, 4, 8}) {
        run<0>("v_xor_b32 v,v (VOP2)", per_cu, n_cu, d_out, d_clk);
        run<4>("v_xor_b32 s,v (VOP2)", per_cu, n_cu, d_out, d_clk);
        run<1>("v_bitop3_b32 v,v,s", per_cu, n_cu, d_out, d_clk);
        run<2>("v_bitop3_b32 v,v,v", per_cu, n_cu, d_out, d_clk);
        run<6>("v_or3_b32 v,v,v", per_cu, n_cu, d_out, d_clk);
        run<7>("v_and_or_b32 v,v,s", per_cu, n_cu, d_out, d_clk);
        run<3>("v_bcnt_u32_b32", per_cu, n_cu, d_out, d_clk);
        run<5>("scan mix xor+2bitop3+bcnt", per_cu, n_cu, d_out, d_clk);
        run<8>("v_bcnt_u32_b32 v,0", per_cu, n_cu, d_out, d_clk);
        run<9>("v_min_u32 (VOP2)", per_cu, n_cu, d_out, d_clk);
        run<10>("v_min3_u32", per_cu, n_cu, d_out, d_clk);
        run<11>("v_cmp_le_u32 vcc", per_cu, n_cu, d_out, d_clk);
        run<12>("v_add_u32 (VOP2)", per_cu, n_cu, d_out, d_clk);
        run<13>("v_mov_b32 v,s", per_cu, n_cu, d_out, d_clk);
        run<14>("filter mix xor,bitop3,bcnt,min", per_cu, n_cu, d_out, d_clk);
        run<15>("3 xor : 1 bcnt", per_cu, n_cu, d_out, d_clk);
    }
    return 0;
}
