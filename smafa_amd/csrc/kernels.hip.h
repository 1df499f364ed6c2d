// kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the smafa scan engine.
//
// What the reference does per (query, subject) pair is XOR + popcount over a 5-bit one-hot
// code, halved (WindowSet::get_distances, /root/reference/src/lib.rs:71-89).  Here a symbol is
// a b-bit CODE (b = 3 planes for ACGTN, 5 for the amino-acid extension) stored as BIT-PLANES:
// plane p of a subject is the bitset over columns of bit p of each column's code.  Then
//
//     mismatch_mask = OR_p ( S_p XOR Q_p )        distance = popcount(mismatch_mask)
//
// which is exactly "number of columns whose symbols differ" for any injective code — the same
// integer the reference computes.  Per 32 columns that is P v_bitop3_b32 (acc | (s ^ q), one
// instruction on gfx950) + one v_bcnt_u32_b32 (popcount-accumulate).
//
// HBM layout of the subject block ("wave tile" = 256 subjects = 64 lanes x 4):
//     u32 planes[n_wave_tiles][P][W][256]      W = ceil(seq_len / 32)
// element [t][p][w][i] = word w of plane p of subject t*256 + i.  A lane owns subjects
// 4*lane .. 4*lane+3 of its wave's tile, so every load is one 16-byte global_load_dwordx4 per
// lane, 1 KiB contiguous per wave instruction, and the whole tile (P*W KiB) is read exactly once
// per query block and then lives in VGPRs while the wave walks the query block.
//
// Queries are wave-uniform: a query record (P*W words, padded to QS) is fetched with scalar
// loads into SGPRs and used as the scalar operand of v_bitop3 — no LDS or VGPR traffic at all
// for the query side.  Hits are rare (thresholded), so the append is a wave-aggregated atomic
// (v_mbcnt + s_bcnt1 + one global_atomic_add per wave, emitted by hipcc for atomicAdd(p, 1)).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smafa_amd.h"

namespace smafa {

constexpr int kWaveTile = 256;  // subjects per wave tile
constexpr int kWgWaves = 4;     // waves per workgroup
constexpr int kWgTile = kWaveTile * kWgWaves;

__host__ __device__ constexpr int round_up4(int x) { return (x + 3) & ~3; }
// query record stride in u32 words
__host__ __device__ constexpr int qrec_stride(int planes, int words) { return round_up4(planes * words); }

// planes / qrec are passed as separate `const __restrict__` kernel parameters (not in this struct) so
// that hipcc can prove the query records are never clobbered and fetch them with scalar loads.
struct ScanArgs {
    uint32_t tile_begin, tile_end;  // wave-tile range of this launch
    uint32_t n_wg_tiles;            // ceil((tile_end - tile_begin) / 4)
    uint32_t n_subjects;
    uint32_t q_begin, q_end;  // query range of this launch
    uint32_t qb_size;         // queries per workgroup pass
    uint32_t *thr;            // per-query emission threshold (only ever lowered)
    uint32_t *cnt;            // per-query histogram of emitted distances (k_tight >= 2), stride cnt_stride
    uint32_t cnt_stride;
    uint32_t k_tight;         // 0: thresholds fixed; 1: lower to running minimum; k>=2: lower to running k-th
    smafa_hit *hits;          // NULL: seed pass — tighten thresholds, append nothing
    unsigned long long cap;
    unsigned long long *count;
};

// acc | (s ^ q) in one VALU op.  Truth table over (a=0xF0, b=0xCC, c=0xAA): 0xF0 | (0xCC ^ 0xAA) = 0xF6.
__device__ __forceinline__ uint32_t or_xor(uint32_t acc, uint32_t s, uint32_t q) {
    return __builtin_amdgcn_bitop3_b32(acc, s, q, 0xF6);
}

__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t *p) {
    // agent-scope relaxed load: bypasses the non-coherent caches, so a threshold lowered by a
    // workgroup on another XCD is seen (a stale, higher value would still be correct — see emit()).
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Rare path: append one qualifying pair and, when asked, tighten the query's threshold.
// Correctness of tightening: thr[q] is only lowered to a distance d once at least k subjects with
// distance <= d have been counted, so thr[q] >= (k-th smallest distance of q) at all times and every
// subject within the true bound passes `dist <= thr` whenever it is visited.
__device__ __forceinline__ void emit(const ScanArgs &a, uint32_t q, uint32_t subject, uint32_t dist) {
    if (a.hits) {
        unsigned long long slot = atomicAdd(a.count, 1ull);
        if (slot < a.cap) {
            smafa_hit h;
            h.query = q;
            h.subject = subject;
            h.dist = dist;
            a.hits[slot] = h;
        }
    }
    if (a.k_tight == 1) {
        atomicMin(a.thr + q, dist);
    } else if (a.k_tight >= 2) {
        uint32_t *c = a.cnt + (size_t)q * a.cnt_stride;
        atomicAdd(c + dist, 1u);
        uint32_t seen = 0;
        for (uint32_t t = 0; t <= dist; t++) seen += ld_relaxed(c + t);
        if (seen >= a.k_tight) atomicMin(a.thr + q, dist);
    }
}

// ---------------------------------------------------------------------------------------------
// The scan: one workgroup = 4 waves = 1024 subjects x one block of queries.
// ---------------------------------------------------------------------------------------------
template <int P, int W>
__global__ __launch_bounds__(256) void scan_kernel(const uint4 *__restrict__ planes,
                                                   const uint32_t *__restrict__ qrec, ScanArgs a) {
    constexpr int QS = qrec_stride(P, W);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t wg_tile = blockIdx.x % a.n_wg_tiles;  // tiles fastest: all CUs share one query block
    const uint32_t qblock = blockIdx.x / a.n_wg_tiles;
    const uint32_t tile = a.tile_begin + wg_tile * kWgWaves + wave;
    if (tile >= a.tile_end) return;  // whole wave exits; the kernel has no barrier

    uint4 s[P * W];
    {
        const uint4 *t = planes + (size_t)tile * (P * W * 64) + lane;
#pragma unroll
        for (int i = 0; i < P * W; i++) s[i] = t[i * 64];
    }
    const uint32_t subj0 = tile * kWaveTile + lane * 4u;

    const uint32_t q0 = a.q_begin + qblock * a.qb_size;
    const uint32_t q1 = min(q0 + a.qb_size, a.q_end);

    for (uint32_t qc = q0; qc < q1; qc += 64) {
        const uint32_t nqc = min(64u, q1 - qc);
        // thresholds of the next 64 queries, one per lane, read with one coalesced load
        const uint32_t thr_v = lane < nqc ? ld_relaxed(a.thr + qc + lane) : 0u;
        const uint32_t *qr = qrec + (size_t)qc * QS;  // wave-uniform -> scalar loads
        for (uint32_t i = 0; i < nqc; i++, qr += QS) {
            const uint32_t U = __builtin_amdgcn_readlane(thr_v, i);
            uint32_t d[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t acc = 0;
#pragma unroll
                for (int w = 0; w < W; w++) {
                    uint32_t m = 0;
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const uint4 v = s[p * W + w];
                        const uint32_t sv = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
                        const uint32_t qv = qr[p * W + w];
                        m = (p == 0) ? (sv ^ qv) : or_xor(m, sv, qv);
                    }
                    acc += __builtin_popcount(m);
                }
                d[k] = acc;
            }
            const uint32_t dmin = min(min(d[0], d[1]), min(d[2], d[3]));
            if (dmin <= U) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (d[k] <= U && subj0 + k < a.n_subjects) emit(a, qc + i, subj0 + k, d[k]);
            }
        }
    }
}

// Any (planes, words) shape: subject words are re-read from the tile (L1/L2) per query instead of
// being held in registers.  Correct for every seq_len; used when no specialisation exists.
__global__ __launch_bounds__(256) void scan_generic_kernel(const uint4 *__restrict__ planes,
                                                           const uint32_t *__restrict__ qrec, ScanArgs a, uint32_t P,
                                                           uint32_t W, uint32_t QS) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t wg_tile = blockIdx.x % a.n_wg_tiles;
    const uint32_t qblock = blockIdx.x / a.n_wg_tiles;
    const uint32_t tile = a.tile_begin + wg_tile * kWgWaves + wave;
    if (tile >= a.tile_end) return;
    const uint4 *t = planes + (size_t)tile * ((size_t)P * W * 64) + lane;
    const uint32_t subj0 = tile * kWaveTile + lane * 4u;
    const uint32_t q0 = a.q_begin + qblock * a.qb_size;
    const uint32_t q1 = min(q0 + a.qb_size, a.q_end);
    for (uint32_t q = q0; q < q1; q++) {
        const uint32_t U = ld_relaxed(a.thr + q);
        const uint32_t *qr = qrec + (size_t)q * QS;
        uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        for (uint32_t w = 0; w < W; w++) {
            uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
            for (uint32_t p = 0; p < P; p++) {
                const uint4 v = t[(p * W + w) * 64];
                const uint32_t qv = qr[p * W + w];
                m0 = or_xor(m0, v.x, qv);
                m1 = or_xor(m1, v.y, qv);
                m2 = or_xor(m2, v.z, qv);
                m3 = or_xor(m3, v.w, qv);
            }
            d0 += __builtin_popcount(m0);
            d1 += __builtin_popcount(m1);
            d2 += __builtin_popcount(m2);
            d3 += __builtin_popcount(m3);
        }
        if (d0 <= U && subj0 + 0 < a.n_subjects) emit(a, q, subj0 + 0, d0);
        if (d1 <= U && subj0 + 1 < a.n_subjects) emit(a, q, subj0 + 1, d1);
        if (d2 <= U && subj0 + 2 < a.n_subjects) emit(a, q, subj0 + 2, d2);
        if (d3 <= U && subj0 + 3 < a.n_subjects) emit(a, q, subj0 + 3, d3);
    }
}

// The literal get_distances seam (src/lib.rs:71-89): every subject's distance to ONE query.
// out has n_wave_tiles * 256 entries (padded), one coalesced 16-byte store per lane.
__global__ __launch_bounds__(256) void distances_kernel(const uint4 *__restrict__ planes, uint32_t n_wave_tiles,
                                                        uint32_t P, uint32_t W, const uint32_t *__restrict__ qrec,
                                                        uint4 *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * kWgWaves + (threadIdx.x >> 6);
    if (tile >= n_wave_tiles) return;
    const uint4 *t = planes + (size_t)tile * ((size_t)P * W * 64) + lane;
    uint4 d = make_uint4(0, 0, 0, 0);
    for (uint32_t w = 0; w < W; w++) {
        uint4 m = make_uint4(0, 0, 0, 0);
        for (uint32_t p = 0; p < P; p++) {
            const uint4 v = t[(p * W + w) * 64];
            const uint32_t qv = qrec[p * W + w];
            m.x = or_xor(m.x, v.x, qv);
            m.y = or_xor(m.y, v.y, qv);
            m.z = or_xor(m.z, v.z, qv);
            m.w = or_xor(m.w, v.w, qv);
        }
        d.x += __builtin_popcount(m.x);
        d.y += __builtin_popcount(m.y);
        d.z += __builtin_popcount(m.z);
        d.w += __builtin_popcount(m.w);
    }
    out[(size_t)tile * 64 + lane] = d;
}

// ---------------------------------------------------------------------------------------------
// Packing: code bytes -> bit-planes by wave64 ballot.
// Lane c of a wave reads column 64h + c of one row; __ballot(bit p of the code) IS the 64-column
// slice of plane p (low half = word 2h, high half = word 2h+1).  A wave packs 64 consecutive rows,
// parks row i's words in lane i, then stores each (plane, word) as one coalesced 256-byte row.
//   mode 0: subject tile layout   out[((tile*P + p)*W + w)*256 + (row & 255)]
//   mode 1: query record layout   out[row*QS + p*W + w]
// `first` = absolute index of codes row 0 (appends start mid-tile); rows are absolute indices.
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint8_t *codes, uint64_t first, uint64_t n, uint32_t L,
                                                        uint32_t W, uint32_t *out, int mode, uint32_t QS) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t group = (uint64_t)blockIdx.x * kWgWaves + (threadIdx.x >> 6);
    const uint64_t base = (first / 64 + group) * 64;  // absolute row of lane 0's slot
    const uint64_t end = first + n;
    if (base >= end) return;
    const uint64_t my_row = base + lane;
    const bool mine_valid = my_row >= first && my_row < end;
    for (uint32_t h = 0; h * 2 < W; h++) {
        const uint32_t col = 64 * h + lane;
        uint32_t lo[P], hi[P];
#pragma unroll
        for (int p = 0; p < P; p++) lo[p] = hi[p] = 0;
#pragma unroll 8
        for (uint32_t i = 0; i < 64; i++) {
            const uint64_t row = base + i;  // wave-uniform
            uint32_t code = 0;
            if (row >= first && row < end && col < L) code = codes[(row - first) * L + col];
#pragma unroll
            for (int p = 0; p < P; p++) {
                const unsigned long long b = __ballot((code >> p) & 1u);
                if (lane == i) {
                    lo[p] = (uint32_t)b;
                    hi[p] = (uint32_t)(b >> 32);
                }
            }
        }
        if (mine_valid) {
#pragma unroll
            for (int p = 0; p < P; p++) {
#pragma unroll
                for (int hw = 0; hw < 2; hw++) {
                    const uint32_t w = 2 * h + hw;
                    if (w >= W) continue;
                    const uint32_t v = hw ? hi[p] : lo[p];
                    if (mode == 0) {
                        const uint64_t tile = my_row >> 8;
                        out[((tile * P + p) * W + w) * 256 + (my_row & 255)] = v;
                    } else {
                        out[my_row * QS + p * W + w] = v;
                    }
                }
            }
        }
    }
}

__global__ void fill_u32_kernel(uint32_t *p, uint32_t v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace smafa
