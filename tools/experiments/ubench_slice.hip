// ubench_slice.hip — speed test of the "per-letter bit-slices in LDS" form of the full comparison:
// a tile of subjects is held in LDS as one bitset per (column, letter) — bit j = "subject j has this letter here" —, a query
// reads ONE row per column (the row of its own letter: match bits), and a carry-save adder network counts the matches per
// subject in bit-sliced form; matches >= L - bound is one more carry chain.  Random data; the kernel's answer is checked
// against a direct count on the host for a few (query, subject) pairs.
//   SPLIT = false: a lane owns 32 subjects (one word per row), tile = 2048 subjects, 60 rows read per query  (nucleotides: 5 letters)
//   SPLIT = true : lanes 0-31 take columns 0..29, lanes 32-63 columns 30..59 of the SAME 32 words (tile = 1024 subjects): the
//                  form that fits 20 letters x 60 columns into 160 KB; the halves' counts are added after a lane swap
// Build + run on the GPU box (tools/experiments/run_ubench_slice.sh).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "csa30.inc"
#include "csa60.inc"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int L = 60;
constexpr int NW = 16;  // waves per workgroup (one workgroup per CU: the tile fills the LDS)

// matches >= need  <=>  carry out of (count + (2^digits - need)); k[i] = bit i of (2^digits - need) as a lane mask
__device__ __forceinline__ uint32_t at_least(const uint32_t *b, int digits, uint32_t kbits) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
        if (i < digits) c = __builtin_amdgcn_bitop3_b32(b[i], c, ((kbits >> i) & 1u) ? 0xffffffffu : 0u, 0xE8);
    return c;
}

template <bool SPLIT, int A>
__global__ __launch_bounds__(NW * 64) void slice_kernel(const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ qrows, uint32_t nq,
                                                        uint32_t need, unsigned long long *hits, uint32_t *sample_out) {
    extern __shared__ uint32_t lds[];
    constexpr int ROW_WORDS = SPLIT ? 32 : 64;
    constexpr int ROWS = L * A;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t *src = tiles + (size_t)blockIdx.x * ROWS * ROW_WORDS;
    for (uint32_t i = tid; i < ROWS * ROW_WORDS / 4; i += NW * 64)
        reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(src)[i];
    __syncthreads();
    unsigned long long found = 0;
    const uint32_t lane_off = SPLIT ? (lane & 31u) * 4u : lane * 4u;
    const uint32_t half_shift = SPLIT ? (lane >= 32u ? 16u : 0u) : 0u;
    for (uint32_t q = wave; q < nq; q += NW) {
        // the query's row numbers: 30 dwords, two 16-bit rows each (SPLIT: column c and column c + 30; else columns 2i, 2i + 1)
        const uint32_t *qr = qrows + (size_t)__builtin_amdgcn_readfirstlane((int)q) * 32;
        uint32_t b[6] = {0, 0, 0, 0, 0, 0};
        if (SPLIT) {
#define X(i) (*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + (__builtin_amdgcn_ubfe(qr[i], half_shift, 16u) << 7) + lane_off))
            CSA30(X, b[0], b[1], b[2], b[3], b[4]);
#undef X
            // the other half's count of the same 32 subjects: swap halves, add (5-bit + 5-bit, bit-sliced)
            uint32_t o[5], c = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) o[i] = (uint32_t)__shfl_xor((int)b[i], 32, 64);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const uint32_t s = __builtin_amdgcn_bitop3_b32(b[i], o[i], c, 0x96);
                c = __builtin_amdgcn_bitop3_b32(b[i], o[i], c, 0xE8);
                b[i] = s;
            }
            b[5] = c;
        } else {
#define X(i) (*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + ((((i) & 1) ? (qr[(i) >> 1] >> 16) : (qr[(i) >> 1] & 0xffffu)) << 8) + lane_off))
            CSA60(X, b[0], b[1], b[2], b[3], b[4], b[5]);
#undef X
        }
        const uint32_t ok = at_least(b, 6, 64u - need);
        if (__ballot(ok != 0u) != 0ull) found += __builtin_popcount(ok);  // (the product kernel would emit rows here)
        if (sample_out && blockIdx.x == 0 && q < 4u && (!SPLIT || lane < 32u)) {  // counts of subject (lane, bit 0..) for the host check
#pragma unroll
            for (int i = 0; i < 6; i++) sample_out[(q * 64 + lane) * 6 + i] = b[i];
        }
    }
    if (SPLIT && lane >= 32u) found = 0;  // both halves hold the same answer
    for (int off = 32; off > 0; off >>= 1) found += __shfl_xor((unsigned long long)found, off, 64);
    if (lane == 0 && found) atomicAdd(hits, found);
}

template <bool SPLIT, int A>
int run(const char *label, uint32_t n_subjects, uint32_t nq, uint32_t need) {
    constexpr int ROW_WORDS = SPLIT ? 32 : 64;
    constexpr int ROWS = L * A;
    const uint32_t tile_subjects = ROW_WORDS * 32 / (SPLIT ? 1 : 1);
    const uint32_t per_tile = SPLIT ? 1024u : 2048u;
    (void)tile_subjects;
    const uint32_t n_tiles = (n_subjects + per_tile - 1) / per_tile;
    const size_t tile_words = (size_t)ROWS * ROW_WORDS;
    // one random tile repeated (the kernel's speed does not depend on the data; the check uses tile 0)
    std::vector<uint8_t> letters((size_t)per_tile * L);
    srand(7);
    for (auto &x : letters) x = (uint8_t)(rand() % A);
    std::vector<uint32_t> tile(tile_words, 0);
    for (uint32_t j = 0; j < per_tile; j++)
        for (int c = 0; c < L; c++) {
            const uint32_t row = c * A + letters[(size_t)j * L + c];
            tile[(size_t)row * ROW_WORDS + (j >> 5)] |= 1u << (j & 31);
        }
    std::vector<uint8_t> qletters((size_t)nq * L);
    for (auto &x : qletters) x = (uint8_t)(rand() % A);
    for (int c = 0; c < L; c++) qletters[c] = letters[c];  // query 0 = subject 0 but for three columns
    qletters[3] ^= 1, qletters[40] ^= 1, qletters[59] ^= 1;
    std::vector<uint32_t> qrows((size_t)nq * 32, 0);
    for (uint32_t q = 0; q < nq; q++)
        for (int c = 0; c < L; c++) {
            const uint32_t row = c * A + (qletters[(size_t)q * L + c] % A);
            if (SPLIT) qrows[(size_t)q * 32 + (c % 30)] |= row << (c >= 30 ? 16 : 0);
            else qrows[(size_t)q * 32 + (c >> 1)] |= row << ((c & 1) ? 16 : 0);
        }
    uint32_t *d_tiles, *d_q, *d_sample;
    unsigned long long *d_hits;
    CHECK(hipMalloc(&d_tiles, tile_words * 4 * (size_t)n_tiles));
    for (uint32_t t = 0; t < n_tiles; t++) CHECK(hipMemcpy(d_tiles + tile_words * t, tile.data(), tile_words * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_q, qrows.size() * 4));
    CHECK(hipMemcpy(d_q, qrows.data(), qrows.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_hits, 8));
    CHECK(hipMalloc(&d_sample, 4 * 64 * 6 * 4));
    CHECK(hipMemset(d_sample, 0, 4 * 64 * 6 * 4));
    const size_t lds_bytes = tile_words * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(slice_kernel<SPLIT, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    unsigned long long hits = 0;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipMemset(d_hits, 0, 8));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((slice_kernel<SPLIT, A>), dim3(n_tiles), dim3(NW * 64), lds_bytes, 0, d_tiles, d_q, nq, need, d_hits, rep == 0 ? d_sample : nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
        CHECK(hipMemcpy(&hits, d_hits, 8, hipMemcpyDeviceToHost));
    }
    // check: the bit-sliced counts of queries 0..3 against subjects 0..(64 lanes x 32 bits) of tile 0
    std::vector<uint32_t> sample(4 * 64 * 6);
    CHECK(hipMemcpy(sample.data(), d_sample, sample.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    const uint32_t lanes = SPLIT ? 32u : 64u;
    for (uint32_t q = 0; q < 4; q++)
        for (uint32_t lane = 0; lane < lanes; lane++)
            for (uint32_t bit = 0; bit < 32; bit++) {
                const uint32_t j = lane * 32 + bit;
                uint32_t want = 0, got = 0;
                for (int c = 0; c < L; c++) want += letters[(size_t)j * L + c] == qletters[(size_t)q * L + c] % A;
                for (int i = 0; i < 6; i++) got |= ((sample[(q * 64 + lane) * 6 + i] >> bit) & 1u) << i;
                bad += want != got;
            }
    const double pairs = (double)n_tiles * per_tile * nq;
    printf("%-44s %u tiles x %u queries: %.3f ms  = %.3f ms per 1e11 pairs  (hits %llu, count check: %s, LDS %zu KB)\n", label, n_tiles, nq, best,
           best * 1e11 / pairs, hits, bad ? "MISMATCH" : "ok", lds_bytes >> 10);
    hipFree(d_tiles), hipFree(d_q), hipFree(d_hits), hipFree(d_sample);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000u, nq = argc > 2 ? (uint32_t)atol(argv[2]) : 10000u;
    int rc = 0;
    rc |= run<false, 5>("unsplit, 5 letters (nt), tile 2048", n, nq, 55);
    rc |= run<false, 10>("unsplit, 10 classes, tile 2048", n, nq, 55);
    rc |= run<true, 20>("split halves, 20 letters (aa), tile 1024", n, nq, 55);
    return rc;
}
