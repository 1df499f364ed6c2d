// Block index of a resident store: the exact answer to a fixed tight bound without visiting every subject.
//
// Pigeonhole: cut the L packed columns into B disjoint blocks.  A subject within distance d of a query has at most d
// mismatching columns, so of ANY d + 1 blocks at least one holds none — the subject and the query agree on every column of
// that block.  The index keeps, per block, the store's positions sorted by a 32-bit key of the block's column codes
// (kp[b][i] = {key, position}) and a directory over the key's top bits.  A scan with bound d <= B - 1 probes d + 1 blocks per
// query: the subjects with the query's key sit in one directory slot, every subject of the run is compared IN
// FULL (all planes, the same mismatch mask as every scan kernel), and a pair within the bound is reported by the FIRST
// probed block on which it agrees exactly — once, whatever the hash does: a key collision only adds a candidate that the
// full comparison then rejects or that some other probe owns.  Nothing is approximated; the rows are those of the scan
// kernels (tests/test_gpu_index.py holds the two against each other and against the oracle).
//
// This is the reference's get_distances loop (src/lib.rs:71-89) restricted to the subjects that can still be within the
// bound — for uniform 60-column amino-acid rows at d = 5 that is ~1 candidate per probe instead of 10M subjects per query.
// Whether it pays is a property of the STORE (engine.hip index_usable): blocks whose largest run or mean run per subject is
// long — gap-only or low-complexity columns, dense families — are not probed, and without d + 1 good blocks the scan kernels run.
#pragma once

namespace smafa {

constexpr int kIndexMaxBlocks = 32;  // blocks per store (bounds up to 31)
constexpr int kIndexMaxWords = 4;    // words per plane the index handles (L <= 128, the per-length kernels' range)
#ifndef SMAFA_INDEX_GROUP
#define SMAFA_INDEX_GROUP 4  // 2: 17, 4: 14, 8: 20 us per 10 000 queries x 6 probes (profiles/r04_index.txt)
#endif
constexpr int kIndexGroup = SMAFA_INDEX_GROUP;  // lanes that share one (query, block) probe: they stride over the run of candidates

struct IndexArgs {
    const uint2 *kp;       // [B][n] sorted per block by key: {key, position in the packed store}
    const uint4 *rows;     // [n][index_row_vectors(PS, W)]: every subject's plane words side by side (p * W + w), zero-padded
    const uint32_t *dir;   // [B][2^dir_bits + 1]: dir[b][j] = first slot whose key >> (32 - dir_bits) >= j
    uint32_t n;            // subjects indexed (= the store's)
    uint32_t dir_bits;
    uint32_t n_probes;     // bound + 1
    uint32_t bound;
    uint32_t L, QS;
    uint32_t q_begin, q_end;
    uint8_t probe_block[kIndexMaxBlocks];   // block probed by probe j
    uint32_t probe_cols[kIndexMaxBlocks];   // ... = packed columns [lo 16 bits, hi 16 bits)
};

// 32-bit key of a block of columns: word(p, w) = word w of plane p of the row.  The same function for subjects and queries.
template <class WordFn>
__device__ __forceinline__ uint32_t index_block_key(uint32_t planes, uint32_t c0, uint32_t c1, WordFn word) {
    unsigned long long h = 0x9e3779b97f4a7c15ull;
    for (uint32_t p = 0; p < planes; p++)
        for (uint32_t w = c0 >> 5; w <= (c1 - 1u) >> 5; w++) {
            const uint32_t lo = max(c0, w * 32u) - w * 32u, hi = min(c1, w * 32u + 32u) - w * 32u;
            const uint32_t bits = (word(p, w) >> lo) & (hi - lo == 32u ? ~0u : (1u << (hi - lo)) - 1u);
            h = (h ^ bits) * 0xff51afd7ed558ccdull;
            h ^= h >> 32;
        }
    h *= 0xc4ceb9fe1a85ec53ull;
    return (uint32_t)(h >> 32);
}

// is the mismatch mask clear on columns [c0, c1)?
__device__ __forceinline__ bool index_block_clear(const uint32_t *m, uint32_t c0, uint32_t c1) {
    uint32_t any = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)kIndexMaxWords; w++) {
        const uint32_t b0 = w * 32u, b1 = b0 + 32u;
        if (c0 < b1 && c1 > b0) {
            const uint32_t lo = max(c0, b0) - b0, hi = min(c1, b1) - b0;
            any |= (m[w] >> lo) & (hi - lo == 32u ? ~0u : (1u << (hi - lo)) - 1u);
        }
    }
    return any == 0u;
}

// keys of one block for every position of the store (+ the identity payload of the sort)
__global__ __launch_bounds__(256) void index_keys_kernel(const uint32_t *__restrict__ planes, uint32_t PS, uint32_t W, uint32_t n,
                                                         uint32_t c0, uint32_t c1, uint32_t *__restrict__ keys,
                                                         uint32_t *__restrict__ iota) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *t = planes + (size_t)(i >> 8) * ((size_t)PS * W * 256) + (i & 255u);
    keys[i] = index_block_key(PS, c0, c1, [&](uint32_t p, uint32_t w) { return t[(size_t)(p * W + w) * 256]; });
    iota[i] = i;
}

__device__ __forceinline__ uint32_t index_lower_bound(const uint32_t *__restrict__ keys, uint32_t lo, uint32_t hi, uint32_t key) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void index_dir_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t dir_bits,
                                                        uint32_t *__restrict__ dir) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t entries = 1u << dir_bits;
    if (j > entries) return;
    dir[j] = j == entries ? n : index_lower_bound(keys, 0u, n, j << (32u - dir_bits));
}

// What probing this block costs: stats[0] = longest run of equal keys, stats[1..2] = sum over runs of len^2 (u64) — a
// query drawn like the store's own rows meets sum(len^2) / n candidates in this block.
__global__ __launch_bounds__(256) void index_stats_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                          unsigned long long *__restrict__ stats) {
    // a fixed grid striding over the keys, one pair of atomics per wave at the end (one per wave and 64 keys was 3.6 of the
    // 4.3 ms a block of a 10M-row store took to build: ~300 000 atomics on two addresses)
    unsigned long long sq = 0, mx = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (i != 0 && keys[i - 1] == keys[i]) continue;
        const uint32_t k = keys[i];
        uint32_t step = 1;  // gallop to the end of the run (runs are short where the index is of any use)
        while (i + step < n && keys[i + step] == k) step <<= 1;
        uint32_t lo = i + (step >> 1), hi = min(i + step, n);  // keys[lo] == k, keys[hi] != k or hi == n
        while (lo + 1 < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (keys[mid] == k) lo = mid;
            else hi = mid;
        }
        const unsigned long long len = hi - i;
        sq += len * len;
        mx = len > mx ? len : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sq += shfl_u64(sq, (int)((__lane_id() + off) & 63u));
        const unsigned long long o = shfl_u64(mx, (int)((__lane_id() + off) & 63u));
        mx = o > mx ? o : mx;
    }
    if (__lane_id() == 0 && sq) {
        atomicMax(stats, mx);
        atomicAdd(stats + 1, sq);
    }
}

// A candidate is compared from a ROW-MAJOR copy of the store the index keeps for itself: the bit-plane layout spreads a subject
// over PS * W arrays (ten 4-byte gathers from ten cache lines for 60 amino-acid columns); side by side they are 1-2 lines and
// three 16-byte loads (bound 14 on 10M aa rows, 950 candidates per query: 1.86 -> see profiles/r04_index.txt).
__host__ __device__ constexpr int index_row_vectors(int ps, int w) { return (ps * w + 3) / 4; }

__global__ __launch_bounds__(256) void index_rows_kernel(const uint32_t *__restrict__ planes, uint32_t PS, uint32_t W, uint32_t n,
                                                         uint32_t *__restrict__ rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *t = planes + (size_t)(i >> 8) * ((size_t)PS * W * 256) + (i & 255u);
    const uint32_t words = PS * W, stride = ((words + 3u) / 4u) * 4u;
    uint32_t *r = rows + (size_t)i * stride;
    for (uint32_t j = 0; j < stride; j++) r[j] = j < words ? t[(size_t)j * 256] : 0u;
}

// One group of kIndexGroup lanes per (query, probe).  Rows are parked in LDS and written out once per workgroup: every row
// taking its own slot from the list's counter costs ~12 ns of serialised atomics on one address — 5 555 rows of the metric's
// launch were 55 of the kernel's 65 us (profiles/r04_index.txt).
constexpr int kIndexWg = 1024;        // threads per workgroup: 256 probes at 4 lanes each
constexpr int kIndexStageRows = 512;  // rows a workgroup parks before the rest goes straight to the list
template <int PS, int PQ, int W>
__global__ __launch_bounds__(kIndexWg) void index_probe_kernel(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ qrec,
                                                               const IndexArgs x, const ScanArgs a) {
    static_assert(W <= kIndexMaxWords, "index_block_clear spells out kIndexMaxWords words");
    __shared__ smafa_hit stage[kIndexStageRows];
    __shared__ uint32_t n_staged;
    __shared__ unsigned long long base;
    if (threadIdx.x == 0) n_staged = 0;
    __syncthreads();
    const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / (uint32_t)kIndexGroup;
    const uint32_t sub = threadIdx.x % (uint32_t)kIndexGroup;
    const uint32_t nq = x.q_end - x.q_begin;
    [&]() {
        if (g >= nq * x.n_probes) return;
        const uint32_t q = x.q_begin + g / x.n_probes, j = g % x.n_probes;
        const uint32_t b = x.probe_block[j];
        const uint32_t cols = x.probe_cols[j], c0 = cols & 0xffffu, c1 = cols >> 16;
        const uint32_t *rec = qrec + (size_t)q * x.QS;
        uint32_t qw[PQ * W];
#pragma unroll
        for (int p = 0; p < PQ; p++)
#pragma unroll
            for (int w = 0; w < W; w++) qw[p * W + w] = rec[qslot(PQ, W, p, w)];
        // a query letter the store has never seen (a bit in a plane past PS) can match no subject in that column: the probe is empty
        uint32_t extra[kIndexMaxWords] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int w = 0; w < W; w++)
#pragma unroll
            for (int p = PS; p < PQ; p++) extra[w] |= qw[p * W + w];
        if (!index_block_clear(extra, c0, c1)) return;
        const uint32_t key = index_block_key((uint32_t)PS, c0, c1, [&](uint32_t p, uint32_t w) {
            uint32_t v = 0;  // (qw is indexed with compile-time subscripts only: it stays in registers)
#pragma unroll
            for (int pp = 0; pp < PS; pp++)
#pragma unroll
                for (int ww = 0; ww < W; ww++) v = (p == (uint32_t)pp && w == (uint32_t)ww) ? qw[pp * W + ww] : v;
            return v;
        });
        // Four dependent memory round trips per probe: [probe columns, query record] -> directory -> (key, position) slots ->
        // the candidate's row.  The directory slot holds every key with the probe key's top dir_bits bits (~2-3 entries):
        // the group's lanes read them side by side, no search.
        const uint2 *kp = x.kp + (size_t)b * x.n;
        const uint32_t *dir = x.dir + (size_t)b * ((1u << x.dir_bits) + 1u);
        const uint32_t slot = key >> (32u - x.dir_bits);
        const uint32_t lo = dir[slot], hi = dir[slot + 1];
        for (uint32_t i = lo + sub; i < hi; i += (uint32_t)kIndexGroup) {
            const uint2 e = kp[i];
            if (e.x > key) break;  // sorted: nothing further on in this lane's stride can match
            if (e.x != key) continue;
            const uint32_t at = e.y;
            constexpr int RV = index_row_vectors(PS, W);
            uint32_t sw[RV * 4];
#pragma unroll
            for (int v = 0; v < RV; v++) {
                const uint4 x4 = x.rows[(size_t)at * RV + v];
                sw[v * 4 + 0] = x4.x, sw[v * 4 + 1] = x4.y, sw[v * 4 + 2] = x4.z, sw[v * 4 + 3] = x4.w;
            }
            uint32_t m[kIndexMaxWords] = {0u, 0u, 0u, 0u};
            uint32_t dist = 0;
#pragma unroll
            for (int w = 0; w < W; w++) {
                uint32_t mw = extra[w];
#pragma unroll
                for (int p = 0; p < PS; p++) mw = or_xor(mw, sw[p * W + w], qw[p * W + w]);
                m[w] = mw;
                dist += (uint32_t)__builtin_popcount(mw);
            }
            if (dist > x.bound || !index_block_clear(m, c0, c1)) continue;
            bool mine = true;  // the first probed block the pair agrees on reports it
            for (uint32_t e2 = 0; e2 < j; e2++) {
                const uint32_t ec = x.probe_cols[e2];
                if (index_block_clear(m, ec & 0xffffu, ec >> 16)) mine = false;
            }
            if (!mine) continue;
            const uint32_t parked = atomicAdd(&n_staged, 1u);
            if (parked < (uint32_t)kIndexStageRows) {
                smafa_hit h;
                h.query = q;
                h.subject = a.order[at];
                h.dist = dist;
                stage[parked] = h;
            } else {
                emit_direct(a, q, at, dist);  // a dense neighbourhood: straight to the list, one atomic per wave
            }
        }
    }();
    __syncthreads();
    const uint32_t n = min(n_staged, (uint32_t)kIndexStageRows);
    if (n == 0) return;  // (uniform: n_staged is the workgroup's)
    if (threadIdx.x == 0) base = atomicAdd(a.count, (unsigned long long)n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        if (base + i < a.cap) a.hits[base + i] = stage[i];
}

// (key, position) pairs of one block, side by side: a probe reads both with one load
__global__ __launch_bounds__(256) void index_interleave_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ pos,
                                                               uint32_t n, uint2 *__restrict__ kp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) kp[i] = make_uint2(keys[i], pos[i]);
}

}  // namespace smafa
