// kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the smafa scan engine.
//
// What the reference does per (query, subject) pair is XOR + popcount over a 5-bit one-hot
// code, halved (WindowSet::get_distances, /root/reference/src/lib.rs:71-89).  Here a symbol is
// a b-bit CODE (3 bits for ACGTN — 2 stored planes while no subject holds an N — and 5 for the amino-acid
// extension) stored as BIT-PLANES:
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
// element [t][p][w][i] = word w of plane p of the subject at POSITION t*256 + i.  A lane owns positions
// 4*lane .. 4*lane+3 of its wave's tile, so every load is one 16-byte global_load_dwordx4 per
// lane, 1 KiB contiguous per wave instruction.
// Three free parameters of that layout are chosen for the prefilter and change no result (a distance counts columns
// whose codes differ): the COLUMN ORDER (most informative columns first), a PER-COLUMN RE-CODING of the symbols (plane 0
// = the best-balanced split of the column's letters: it is the plane every lower bound looks at) — both applied by
// pack_rows_kernel to subjects and queries alike — and the SUBJECT ORDER: big appends are sorted by their filter words,
// order[position] maps back to the subject index, and zone[tile] records the filter bits the 256 subjects of a tile
// share (scan_zone_kernel's first level).  See DESIGN.md §2.
//
// Query records (u32 words, stride qrec_stride(P, W)):
//     [ filter word 0 | filter word 1 | bound slot | filter words 2..W-1 | the other planes' words ]
// (one-word records: [ filter word 0 | bound slot | ... ]) — see qslot() / bound_slot().
// The 4 waves of a workgroup walk the same query block, so they stage 64 records at a time in LDS
// (double-buffered, one barrier per 64 queries).  Measured on MI355X (profiles/r01_ubench_valu*.txt) v_xor / v_add /
// v_bitop3 with all-VGPR sources issue at ~60 T lane-ops/s chip-wide, while the same ops with an SGPR source — and
// v_bcnt, v_min, v_cmp, v_or3, v_readlane regardless of sources — issue at ~37 T.
//
// Hits are rare (thresholded): rows are parked in an LDS stage per workgroup and written out with one global
// reservation per 64-query chunk (RowStage, emit, flush_rows, finish_rows).
#pragma once

#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smafa_amd.h"

namespace smafa {

constexpr int kWaveTile = 256;  // subjects per wave tile
constexpr int kWgWaves = 4;     // waves per workgroup
constexpr int kChunk = 64;      // queries staged in LDS at a time
constexpr int kStageRows = 256;  // rows a workgroup parks in LDS (per chunk parity) before ONE reservation in the row list
// __launch_bounds__ second argument (waves per SIMD the register budget must allow) for the scan kernel:
// the subject words held per lane plus ~40 working registers, mapped through the gfx950 allocation steps.
__host__ __device__ constexpr int scan_min_waves(int ps, int w, int t) {
    const int regs = t * ps * w * 4 + 40;
    return regs <= 64 ? 8 : regs <= 80 ? 6 : regs <= 96 ? 5 : regs <= 128 ? 4 : regs <= 168 ? 3 : 2;
}
#ifndef SMAFA_CASCADE
#define SMAFA_CASCADE 1  // 1: a one-word first-level bound in front of the folded bound (+7 % aa, +4 % nt measured)
#endif
#ifndef SMAFA_WIDE_AND_PAIR
#define SMAFA_WIDE_AND_PAIR 1  // scan_wide_kernel keeps the two-subjects-per-popcount form of level 2: it only runs at bounds
                               // level 1 prunes at, where the form is the cheaper one (one-word stores, 10M x 20 aa, bound 3:
                               // 7.1 ms vs 8.6 ms per subject — profiles/r02_short_check.txt)
#endif
#ifndef SMAFA_SUM_FOLD
#define SMAFA_SUM_FOLD 1  // scan_kernel<.., FOLD = 1 | 2 | 3> for two-word launches with a bound of 13..17 | 18..32 | above (engine.hip launch_scan_t)
#endif
#ifndef SMAFA_SIGN_COMPARE
#define SMAFA_SIGN_COMPARE 1  // scan_kernel's full comparison: one sign test per query instead of a compare per subject
#endif
#ifndef SMAFA_AND_PAIR
#define SMAFA_AND_PAIR 0  // 1: the folded bound takes two subjects per popcount (popcount(a & b) <= both).  +5 % in round 1 at
                          // bound 5; nothing today at tight bounds (the zone level runs in front), and at bounds 9-13 it
                          // lets every query through where the per-subject fold still rejects them: scan_kernel, 10 000
                          // queries x 10M aa, bound 10: 28.7 -> 8.4 ms (profiles/r02_bound_probe.txt)
#endif

__host__ __device__ constexpr int round_up4(int x) { return (x + 3) & ~3; }
// query record stride in u32 words: the plane words plus the bound slot, rounded up to whole uint4s
__host__ __device__ constexpr int qrec_stride(int planes, int words) { return round_up4(planes * words + 1); }
// The plane the prefilter looks at: always plane 0.  Codes are re-coded PER COLUMN when a store is laid out (the
// distance only depends on equality of codes within a column, so any per-column injective map keeps every result):
// bit 0 of the re-coded symbol splits the column's letters into two sets of nearly equal total frequency, which makes
// a mismatch flip the filter bit as often as the column allows (host: choose_layout() in engine.hip).  Without
// statistics the nucleotide map is A=0 C=2 G=1 T=3 N=4: bit 0 separates {A,C} from {G,T}, so it sees both transitions.
__host__ __device__ constexpr int filter_plane(int) { return 0; }
// slot of word w of plane p inside a record
// The bound sits right after the first two filter words, so the first uint4 of a record holds everything
// levels 1 and 2 of scan_wide_kernel need: [f0 f1 bound f2 f3 ... | other planes] ([f0 bound] for one word).
__host__ __device__ constexpr int bound_slot(int words) { return words < 2 ? words : 2; }
__host__ __device__ constexpr int qslot(int planes, int words, int p, int w) {
    return p == filter_plane(planes) ? (w < bound_slot(words) ? w : w + 1)
                                     : words + 1 + (p < filter_plane(planes) ? p : p - 1) * words + w;
}

struct ScanArgs {
    uint32_t tile_begin, tile_end;  // wave-tile range of this launch
    uint32_t n_wg_tiles;            // ceil((tile_end - tile_begin) / (4 * tiles per wave))
    uint32_t n_subjects;
    uint32_t q_begin, q_end;  // query range of this launch
    uint32_t qb_size;         // queries per workgroup pass
    uint32_t *thr;            // per-query emission bound (only ever lowered); NULL: every query uses thr0
    uint32_t thr0;            // the fixed bound of a launch without per-query bounds
    uint32_t *cnt;            // per-query histogram of emitted distances (k_tight >= 2), stride cnt_stride
    uint32_t cnt_stride;
    uint32_t k_tight;         // 0: bounds fixed; 1: lower to running minimum; k>=2: lower to running k-th
    uint32_t use_filter;      // 0: always the full comparison; 1: prefilter with per-wave fallback
    // Row append: ONE list, ONE counter.  A workgroup parks its rows in LDS (RowStage) and reserves room in the list
    // once per 64-query chunk, so a dense scan costs the counter one atomic per workgroup and chunk instead of one per
    // wave and row group (one counter saturates near 80 M atomics/s; a dense scan appends > 1 G rows/s).
    smafa_hit *hits;            // NULL: seed / counting pass — tighten bounds, append nothing
    unsigned long long cap;     // rows `hits` can hold; rows past it are dropped, the counter keeps counting
    unsigned long long *count;  // rows appended so far
    // done != NULL: the last workgroup to finish publishes the total in *publish and zeroes count and done again —
    // a fixed-bound scan is then ONE launch: no counter reset before it, no gather after it
    uint32_t *done;
    unsigned long long *publish;
    const uint32_t *order;      // position in the packed store -> subject index (the order subjects were appended in)
    const uint4 *zone;          // per wave tile: {bits all its subjects share in filter word 0, which bits those are,
                                //                 the same for word 1} — see zone_kernel
    uint32_t zone_on;           // scan_wide_kernel: apply the zone level (the store is sorted well enough for it to pay)
    uint32_t stream_once;       // the launch has ONE query block (every store byte it touches is read once), or what a query
                                // block reads is too big for the Infinity Cache to keep until the next one: scan_lazy_kernel
                                // then loads its filter words (scan_kernel: its tiles) with the non-temporal hint (no cache line is kept for a
                                // re-read that never comes): 66.7 -> 60.6 us per pass over the 50M store's 400 MB plane
                                // (0.74 -> 0.82 of 8 TB/s), 14.0 -> 13.1 us on the 10M store (profiles/r03_stream_nt.txt)
};

// Where a workgroup parks qualifying rows between two flushes: one buffer per chunk parity, so that the rows of
// chunk k are written out while chunk k+1 already appends to the other buffer (no extra barrier on the common path).
template <int ROWS>
struct RowStageT {
    static constexpr int kRows = ROWS;
    smafa_hit rows[2][ROWS];
    uint32_t n[2];            // rows parked (may run past ROWS: the excess went straight to the list)
    unsigned long long base;  // the flushing thread's reservation, read by the whole workgroup
};
using RowStage = RowStageT<kStageRows>;

// acc | (s ^ q) in one VALU op.  Truth table over (a=0xF0, b=0xCC, c=0xAA): 0xF0 | (0xCC ^ 0xAA) = 0xF6.
__device__ __forceinline__ uint32_t or_xor(uint32_t acc, uint32_t s, uint32_t q) {
    return __builtin_amdgcn_bitop3_b32(acc, s, q, 0xF6);
}
// a 16-byte load with the non-temporal hint (global_load_dwordx4 ... nt): for bytes a launch reads exactly once
__device__ __forceinline__ uint4 ld_nt(const uint4 *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// a | b | c through the fast bitop3 path (v_or3_b32 itself issues at the slow rate)
__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE);
}

// number of set bits of a wave mask in the lanes below the calling lane (v_mbcnt_lo/hi): a lane's rank among the
// lanes that append together
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t *p) {
    // agent-scope relaxed load: bypasses the non-coherent caches, so a bound lowered by a workgroup on
    // another XCD is seen (a stale, higher value would still be correct — see emit()).
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    return ((unsigned long long)(uint32_t)__shfl((int)(v >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(v & 0xffffffffull), src, 64);
}

// Rare path: append one qualifying pair and, when asked, tighten the query's bound.  `pos` is the subject's position
// in the packed store; the row carries its subject index (a.order).
// Correctness of tightening: thr[q] is only lowered to a distance d once at least k subjects with
// distance <= d have been counted, so thr[q] >= (k-th smallest distance of q) at all times and every
// subject within the true bound passes `dist <= thr` whenever it is visited.
template <int ROWS>
__device__ __forceinline__ void emit(const ScanArgs &a, RowStageT<ROWS> &rs, int parity, uint32_t q, uint32_t pos, uint32_t dist) {
    if (a.hits) {
        // The lanes that got here together (a wave works on one query at a time, so they all append for the same
        // query) take consecutive slots of the workgroup's LDS stage: one LDS atomic per wave (v_mbcnt ranks the lanes).
        smafa_hit h;
        h.query = q;
        h.subject = a.order[pos];
        h.dist = dist;
        const unsigned long long together = __ballot(1);  // the active lanes
        const uint32_t lane = __lane_id();
        const int leader = __builtin_ctzll(together);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&rs.n[parity], (uint32_t)__builtin_popcountll(together));
        base = (uint32_t)__shfl((int)base, leader, 64);
        const uint32_t slot = base + lanes_below(together);
        if (slot < (uint32_t)ROWS) {
            rs.rows[parity][slot] = h;
        } else {
            // the stage is full until the next flush (a dense neighbourhood): straight to the list, one atomic per wave
            const unsigned long long spill = __ballot(1);
            const int first = __builtin_ctzll(spill);
            unsigned long long g = 0;
            if ((int)lane == first) g = atomicAdd(a.count, (unsigned long long)__builtin_popcountll(spill));
            g = shfl_u64(g, first) + lanes_below(spill);
            if (g < a.cap) a.hits[g] = h;
        }
    }
    if (a.k_tight == 1) {
        // ties with the current minimum are the common case once the bound has settled: they need no atomic
        if (dist < ld_relaxed(a.thr + q)) atomicMin(a.thr + q, dist);
    } else if (a.k_tight >= 2) {
        uint32_t *c = a.cnt + (size_t)q * a.cnt_stride;
        atomicAdd(c + dist, 1u);
        // Can this pair lower the bound?  Only when it lies strictly below it — and most counted pairs sit ON the bound (the
        // distance distribution rises steeply towards it): those are done (k = 50 without a bound, 10 000 queries: 43.7 ->
        // 32.4 ms of kernels).  The walk itself stays the plain loop: with eight loads in flight it is faster by itself
        // (29.9 ms) but changes the register allocation of every kernel that inlines this function — the prefilter-off launch
        // went from 25.4 to 29.2 ms without ever executing it (this form: 26.4); as a real function call: 27.8 ms
        // (profiles/r04_kth.txt).
        if (dist < ld_relaxed(a.thr + q)) {
            uint32_t seen = 0;
            for (uint32_t t = 0; t <= dist; t++) seen += ld_relaxed(c + t);
            if (seen >= a.k_tight) atomicMin(a.thr + q, dist);
        }
    }
}

// Write the rows parked under `parity` out to the list.  Called by EVERY thread of the workgroup right after the
// barrier that ends a chunk: nobody appends to this parity again before the next chunk's barrier, so n is stable and
// the branch is uniform.
template <int ROWS>
__device__ __forceinline__ void flush_rows(const ScanArgs &a, RowStageT<ROWS> &rs, int parity) {
    const uint32_t parked = rs.n[parity];
    if (parked == 0) return;
    const uint32_t n = min(parked, (uint32_t)ROWS);
    if (threadIdx.x == 0) rs.base = atomicAdd(a.count, (unsigned long long)n);
    __syncthreads();
    const unsigned long long base = rs.base;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        if (base + i < a.cap) a.hits[base + i] = rs.rows[parity][i];
    if (threadIdx.x == 0) rs.n[parity] = 0;  // every thread read it before the barrier above
}

// End of a scan kernel, every thread: (optionally) publish the total.  The workgroup whose ticket is the last one
// reads the counter after every other workgroup's reservations: a reservation is an atomic that has RETURNED (the
// thread used its value) before that workgroup's ticket is taken — by the same thread, or by another thread of the
// workgroup ahead of the barrier below — and device-scope atomics are performed at the device's coherence point, so
// no fence is needed (a __threadfence() here writes the XCD's L2 back once per workgroup: it cost 6x on a one-query
// pass).  The last workgroup also resets both counters, so the next launch on the stream starts from zero without a
// memset.  Row data are plain stores, visible to the stream's next operation like any kernel output.
// Tickets are taken in two levels:
constexpr uint32_t kTicketGroups = 64;   // first-level ticket counters (one atomic address takes ~80 M tickets/s:
constexpr uint32_t kTicketStride = 32;   // 2 442 workgroups of a one-query pass would queue for 30 us on a single one)
// `done` = u32 array: [0] second-level tickets, [kTicketStride * (1 + g)] tickets of group g — every counter on its own
// 128-byte line.  Workgroup b belongs to group b % kTicketGroups.
__device__ __forceinline__ void finish_rows(const ScanArgs &a) {
    if (a.done == nullptr) return;
    __syncthreads();  // this workgroup's reservations (flushes and spills) have all returned
    if (threadIdx.x == 0) {
        const uint32_t groups = min(kTicketGroups, gridDim.x);
        const uint32_t g = blockIdx.x % kTicketGroups;
        const uint32_t members = (gridDim.x - g + kTicketGroups - 1) / kTicketGroups;  // workgroups with this remainder
        uint32_t *mine = a.done + kTicketStride * (1 + g);
        if (atomicAdd(mine, 1u) == members - 1) {  // last of its group
            atomicExch(mine, 0u);
            if (atomicAdd(a.done, 1u) == groups - 1) {  // last group: every reservation of the launch has returned
                *a.publish = atomicExch(a.count, 0ull);
                atomicExch(a.done, 0u);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The scan: one workgroup = 4 waves = 1024 subjects x one block of queries.
//
// Per (query, 4 subjects of a lane) the wave first evaluates an exact LOWER BOUND of the distance:
//     lb = popcount( OR_w ( S_f[w] XOR Q_f[w] ) )           f = filter plane
// Every set bit of that OR is a column (in at least one 32-column word) whose filter-plane bits
// differ, i.e. a mismatching column, and distinct bits are distinct columns within a word, so
// lb <= distance.  The record carries ~bound, and v_bcnt's accumulator adds it for free:
//     popcount(m) + ~bound  <  0   <=>   popcount(m) <= bound
// so "can any of my 4 subjects still qualify" is the sign bit of an OR of four popcounts and costs
// one compare per wave step.  Only when some lane says yes does the wave compute the full distance
// for that query (all planes, all words) and test it exactly.  Nothing is approximated: a pair is
// skipped only when its lower bound already exceeds the query's bound.
// A cheaper first level runs in front of it: popcount(S_f[0] XOR Q_f[0]) over the first 32 columns only (one
// xor + one popcount per subject); the folded bound is evaluated only for queries that pass level 1.
// If the prefilter stops paying (dense neighbourhoods: it passes for more than a quarter of a chunk's
// queries) the wave switches to the plain full comparison and re-probes every 16th chunk; level 1 alone is
// dropped the same way when it passes for more than half of a chunk's queries.
// ---------------------------------------------------------------------------------------------
// PS = planes stored per subject, PQ = planes per query record (PS < PQ only for the N-free nucleotide
// store: subjects carry code bits 0-1, queries still carry the N bit, which mismatches every subject).
// T = wave tiles per wave: a lane owns 4*T subjects, so the per-query OR / compare / branch / LDS read are
// shared by 4*T pairs.  T = 2 where the registers allow it without losing occupancy.
template <int PS, int PQ, int W, int T, bool SEED, int FOLD>
__global__ __launch_bounds__(256, scan_min_waves(PS, W, T)) void scan_kernel(const uint4 *__restrict__ planes,
                                                                             const uint32_t *__restrict__ qrec,
                                                                             ScanArgs a) {
    constexpr int RS = qrec_stride(PQ, W);  // words per record
    constexpr int RV = RS / 4;              // uint4 per record
    constexpr int NV = (kChunk * RV + 255) / 256;
    constexpr int FP = filter_plane(PQ);
    static_assert(FP < PS && PS <= PQ, "the filter plane must be one the subjects store");
    constexpr int BS = bound_slot(W);
    constexpr int HV = (W + 1 + 3) / 4;  // uint4s holding the filter words + bound slot
    // two subjects per popcount only where the folded mask is dense enough to survive an AND (W >= 2);
    // with a single word the bound is taken per subject
    constexpr bool kPair = SMAFA_AND_PAIR && W > 1;
    __shared__ uint4 stage[2][kChunk * RV];
    __shared__ RowStage rs;
    int buf = 0;  // LDS buffer of the chunk being computed = parity of the row stage it appends to

    const uint32_t tid = threadIdx.x;
    if (tid == 0) rs.n[0] = rs.n[1] = 0;  // published by the barrier in front of the chunk loop
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t wg_tile = blockIdx.x % a.n_wg_tiles;  // tiles fastest: all CUs share one query block
    const uint32_t qblock = blockIdx.x / a.n_wg_tiles;
    const uint32_t tile0 = a.tile_begin + (wg_tile * kWgWaves + wave) * T;  // this wave's first tile
    // waves (and trailing tiles of a wave) past the range still take part in staging and barriers

    uint4 s[T][PS * W];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const bool live = tile0 + t < a.tile_end;
        const uint4 *src = planes + (size_t)(live ? tile0 + t : a.tile_begin) * (PS * W * 64) + lane;
        if (a.stream_once) {  // bytes this launch reads once (or that no cache can hold until the next query block)
#pragma unroll
            for (int i = 0; i < PS * W; i++) s[t][i] = ld_nt(src + i * 64);
        } else {
#pragma unroll
            for (int i = 0; i < PS * W; i++) s[t][i] = src[i * 64];
        }
    }
    const bool active = tile0 < a.tile_end;
    const uint32_t q0 = a.q_begin + qblock * a.qb_size;
    const uint32_t q1 = min(q0 + a.qb_size, a.q_end);

    uint4 pre[NV];
    auto fetch = [&](uint32_t qc) {  // global -> registers: this thread's share of chunk [qc, qc+64)
        const uint32_t nqc = min((uint32_t)kChunk, q1 - qc);
        const uint4 *src = reinterpret_cast<const uint4 *>(qrec + (size_t)qc * RS);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const uint32_t idx = tid + v * 256;
            if (idx < nqc * RV) {
                uint4 x = src[idx];
                if (idx % RV == BS / 4) {  // the uint4 holding the bound slot: merge ~bound
                    const uint32_t nu = ~(a.thr ? ld_relaxed(a.thr + qc + idx / RV) : a.thr0);
                    if ((BS & 3) == 0) x.x = nu;
                    else if ((BS & 3) == 1) x.y = nu;
                    else if ((BS & 3) == 2) x.z = nu;
                    else x.w = nu;
                }
                pre[v] = x;
            }
        }
    };
    auto commit = [&](int buf, uint32_t qc) {  // registers -> LDS
        const uint32_t nqc = min((uint32_t)kChunk, q1 - qc);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const uint32_t idx = tid + v * 256;
            if (idx < nqc * RV) stage[buf][idx] = pre[v];
        }
    };

    // full comparison of one query against this lane's 4*T subjects; qw = the whole record
    auto full_compare = [&](const uint32_t(&qw)[RS], uint32_t q) {
        const uint32_t U = ~qw[BS];
        uint32_t lo = 0xffffffffu;  // SEED only
#if SMAFA_SIGN_COMPARE
        if (!SEED) {
            // The distances are accumulated on top of ~bound (v_bcnt's accumulator adds it for free), so "within the bound"
            // is the sign bit, the 4*T sign bits are OR-ed, and ONE compare + ballot per query replaces a compare and a
            // branch per subject (the launch with the prefilter off ran 90 scalar instructions per 1024 pairs, most of them
            // these branches: profiles/r03_pmc.json).  A wave that has a subject in range (rare) looks at them one by one.
            const uint32_t nuU = qw[BS];
            uint32_t dd[T][4];
            uint32_t any = 0;
#pragma unroll
            for (int t = 0; t < T; t++) {
#pragma unroll
                for (int w = 0; w < W; w++) {
                    uint32_t extra = 0;
#pragma unroll
                    for (int p = PS; p < PQ; p++) extra |= qw[qslot(PQ, W, p, w)];
                    uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
#pragma unroll
                    for (int p = 0; p < PS; p++) {
                        const uint4 v = s[t][p * W + w];
                        const uint32_t qv = qw[qslot(PQ, W, p, w)];
                        const bool first = p == 0 && PS == PQ;
                        m0 = first ? (v.x ^ qv) : or_xor(m0, v.x, qv);
                        m1 = first ? (v.y ^ qv) : or_xor(m1, v.y, qv);
                        m2 = first ? (v.z ^ qv) : or_xor(m2, v.z, qv);
                        m3 = first ? (v.w ^ qv) : or_xor(m3, v.w, qv);
                    }
                    dd[t][0] = (w ? dd[t][0] : nuU) + __builtin_popcount(m0);
                    dd[t][1] = (w ? dd[t][1] : nuU) + __builtin_popcount(m1);
                    dd[t][2] = (w ? dd[t][2] : nuU) + __builtin_popcount(m2);
                    dd[t][3] = (w ? dd[t][3] : nuU) + __builtin_popcount(m3);
                }
                any = t ? or3(or3(dd[t][0], dd[t][1], dd[t][2]), dd[t][3], any) : (or3(dd[t][0], dd[t][1], dd[t][2]) | dd[t][3]);
            }
            if (__ballot((int32_t)any < 0) == 0ull) return;  // nobody within the bound (a bound of L and more: every pair is)
#pragma unroll
            for (int t = 0; t < T; t++) {
                const uint32_t subj0 = (tile0 + t) * kWaveTile + lane * 4u;
                const bool live = tile0 + t < a.tile_end;  // a trailing tile slot holds a copy of another tile: ignore it
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (live && (int32_t)dd[t][k] < 0 && subj0 + k < a.n_subjects) emit(a, rs, buf, q, subj0 + k, dd[t][k] - nuU);
            }
            return;
        }
#endif
#pragma unroll
        for (int t = 0; t < T; t++) {
            uint32_t d[4];
#pragma unroll
            for (int w = 0; w < W; w++) {
                uint32_t extra = 0;  // query bits in planes no subject has: a mismatch against every subject
#pragma unroll
                for (int p = PS; p < PQ; p++) extra |= qw[qslot(PQ, W, p, w)];
                uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
#pragma unroll
                for (int p = 0; p < PS; p++) {
                    const uint4 v = s[t][p * W + w];
                    const uint32_t qv = qw[qslot(PQ, W, p, w)];
                    const bool first = p == 0 && PS == PQ;
                    m0 = first ? (v.x ^ qv) : or_xor(m0, v.x, qv);
                    m1 = first ? (v.y ^ qv) : or_xor(m1, v.y, qv);
                    m2 = first ? (v.z ^ qv) : or_xor(m2, v.z, qv);
                    m3 = first ? (v.w ^ qv) : or_xor(m3, v.w, qv);
                }
                d[0] = (w ? d[0] : 0u) + __builtin_popcount(m0);
                d[1] = (w ? d[1] : 0u) + __builtin_popcount(m1);
                d[2] = (w ? d[2] : 0u) + __builtin_popcount(m2);
                d[3] = (w ? d[3] : 0u) + __builtin_popcount(m3);
            }
            const uint32_t subj0 = (tile0 + t) * kWaveTile + lane * 4u;
            const bool live = tile0 + t < a.tile_end;  // a trailing tile slot holds a copy of another tile: ignore it
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool real = live && subj0 + k < a.n_subjects;
                if (SEED) {
                    if (real) lo = min(lo, d[k]);
                } else if (real && d[k] <= U) {
                    emit(a, rs, buf, q, subj0 + k, d[k]);
                }
            }
        }
        if (SEED) {
            // seed pass for the running minimum (its own instantiation: this code in the hot kernel cost 3.5 %
            // through register allocation alone): one atomicMin per wave instead of one per qualifying pair
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
            if (lane == 0 && lo < U) atomicMin(a.thr + q, lo);
        }
    };
    auto read_record = [&](const uint4 *rec, uint32_t(&qw)[RS], int from, int to) {
#pragma unroll
        for (int v = from; v < to; v++) {
            const uint4 x = rec[v];
            qw[4 * v + 0] = x.x;
            qw[4 * v + 1] = x.y;
            qw[4 * v + 2] = x.z;
            qw[4 * v + 3] = x.w;
        }
    };

    if (q0 < q1) {
        fetch(q0);
        commit(0, q0);
    }
    __syncthreads();
    bool filter_on = a.use_filter != 0;
    bool level1_on = true;
    uint32_t chunk_no = 0;
    for (uint32_t qc = q0; qc < q1; qc += kChunk, buf ^= 1, chunk_no++) {
        const uint32_t nqc = min((uint32_t)kChunk, q1 - qc);
        const bool more = qc + kChunk < q1;
        if (more) fetch(qc + kChunk);  // in flight while this chunk is computed
        if (active) {
            const uint4 *rec = &stage[buf][0];
            const bool probe = a.use_filter && (filter_on || (chunk_no & 15u) == 0);
            if (probe) {
                uint32_t passes = 0;  // wave-uniform: queries of this chunk that needed the full comparison
                uint32_t level1_passes = 0;
                // pinned to a scalar register: hipcc cannot see through __ballot that these flags are wave-uniform
                // and would otherwise carry them as lane masks / VGPR counters through the hot loop
                const bool l1 = __builtin_amdgcn_readfirstlane((int)level1_on) != 0;
                for (uint32_t i = 0; i < nqc; i++, rec += RV) {
                    uint32_t qw[RS];
                    // fast path: filter-plane words + bound slot only (FOLD 2: the next plane's words as well)
                    constexpr int HV2 = FOLD == 3 ? RV : FOLD == 2 ? (qslot(PQ, W, filter_plane(PQ) == 0 ? 1 : 0, W - 1) + 4) / 4 : HV;
                    read_record(rec, qw, 0, HV2 > HV ? HV2 : HV);
                    const uint32_t nu = qw[BS];
                    uint32_t any = 0;
                    bool go = true;  // wave-uniform: does this query reach the folded bound?
#if SMAFA_CASCADE
                    // level 1: word 0 of the filter plane only — popcount(s ^ q) over 32 columns, one xor + one
                    // popcount per subject; weaker than the folded bound below but cheaper, and still exact
                    if (W > 1 && l1) {
                        uint32_t any1 = 0;
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            const uint32_t u0 = __builtin_popcount(s[t][FP * W].x ^ qw[0]) + nu;
                            const uint32_t u1 = __builtin_popcount(s[t][FP * W].y ^ qw[0]) + nu;
                            const uint32_t u2 = __builtin_popcount(s[t][FP * W].z ^ qw[0]) + nu;
                            const uint32_t u3 = __builtin_popcount(s[t][FP * W].w ^ qw[0]) + nu;
                            any1 = t ? or3(or3(u0, u1, u2), u3, any1) : (or3(u0, u1, u2) | u3);
                        }
                        go = __ballot((int32_t)any1 < 0) != 0ull;  // nobody can qualify: next query
                        level1_passes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(level1_passes + (go ? 1u : 0u)));
                    }
#endif
                    if (go) {
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            // lower bound on the filter plane, folded over the words
                            uint32_t m0 = s[t][FP * W].x ^ qw[0], m1 = s[t][FP * W].y ^ qw[0];
                            uint32_t m2 = s[t][FP * W].z ^ qw[0], m3 = s[t][FP * W].w ^ qw[0];
#pragma unroll
                            for (int w = 1; w < W; w++) {
                                m0 = or_xor(m0, s[t][FP * W + w].x, qw[qslot(PQ, W, FP, w)]);
                                m1 = or_xor(m1, s[t][FP * W + w].y, qw[qslot(PQ, W, FP, w)]);
                                m2 = or_xor(m2, s[t][FP * W + w].z, qw[qslot(PQ, W, FP, w)]);
                                m3 = or_xor(m3, s[t][FP * W + w].w, qw[qslot(PQ, W, FP, w)]);
                            }
                            if (kPair) {
                                // popcount(a & b) <= min(popcount a, popcount b): one popcount bounds two subjects
                                const uint32_t t0 = __builtin_popcount(m0 & m1) + nu, t2 = __builtin_popcount(m2 & m3) + nu;
                                any = t ? or3(any, t0, t2) : (t0 | t2);
                            } else if (FOLD == 2) {
                                // FOLD 2 (launches whose bound is 18..32 at two words, 3 planes and more): the distance
                                // over TWO planes, word by word — a column whose letters differ shows in the filter
                                // plane or in the next one three times out of four (~45 of 60 columns for unrelated
                                // sequences), for half the instructions of the full comparison
                                constexpr int FP2 = FP == 0 ? 1 : 0;
                                uint32_t t0 = nu, t1 = nu, t2 = nu, t3 = nu;
#pragma unroll
                                for (int w = 0; w < W; w++) {
                                    const uint32_t qa = qw[qslot(PQ, W, FP, w)], qb = qw[qslot(PQ, W, FP2, w)];
                                    t0 += __builtin_popcount(or_xor(s[t][FP * W + w].x ^ qa, s[t][FP2 * W + w].x, qb));
                                    t1 += __builtin_popcount(or_xor(s[t][FP * W + w].y ^ qa, s[t][FP2 * W + w].y, qb));
                                    t2 += __builtin_popcount(or_xor(s[t][FP * W + w].z ^ qa, s[t][FP2 * W + w].z, qb));
                                    t3 += __builtin_popcount(or_xor(s[t][FP * W + w].w ^ qa, s[t][FP2 * W + w].w, qb));
                                }
                                any = t ? or3(or3(t0, t1, t2), t3, any) : (or3(t0, t1, t2) | t3);
                            } else if (FOLD == 3) {
                                // FOLD 3 (launches whose bound starts above 32 — the k-th modes without a bound — on stores
                                // of 4 planes and more): the distance over ALL PLANES BUT THE LAST.  Two different letters
                                // share their first PS - 1 code bits only when they are one of the few pairs the last plane
                                // tells apart (20 letters in 5 bits: 4 such pairs of 190), so this bound sits ~1 below the
                                // distance and still rejects where the running bound of a query without any relative in the
                                // store settles (~47 of 60 columns) — for 2 * (PS - 1) bit-ops per subject instead of 2 * PS.
                                // While bounds are still loose it passes for everybody and the wave falls back to the plain
                                // comparison (filter_on), as with every other level.
                                uint32_t t0 = nu, t1 = nu, t2 = nu, t3 = nu;
#pragma unroll
                                for (int w = 0; w < W; w++) {
                                    uint32_t m0 = s[t][w].x ^ qw[qslot(PQ, W, 0, w)], m1 = s[t][w].y ^ qw[qslot(PQ, W, 0, w)];
                                    uint32_t m2 = s[t][w].z ^ qw[qslot(PQ, W, 0, w)], m3 = s[t][w].w ^ qw[qslot(PQ, W, 0, w)];
#pragma unroll
                                    for (int p = 1; p < PS - 1; p++) {
                                        const uint32_t qv = qw[qslot(PQ, W, p, w)];
                                        m0 = or_xor(m0, s[t][p * W + w].x, qv);
                                        m1 = or_xor(m1, s[t][p * W + w].y, qv);
                                        m2 = or_xor(m2, s[t][p * W + w].z, qv);
                                        m3 = or_xor(m3, s[t][p * W + w].w, qv);
                                    }
                                    t0 += __builtin_popcount(m0);
                                    t1 += __builtin_popcount(m1);
                                    t2 += __builtin_popcount(m2);
                                    t3 += __builtin_popcount(m3);
                                }
                                any = t ? or3(or3(t0, t1, t2), t3, any) : (or3(t0, t1, t2) | t3);
                            } else if (FOLD == 1) {
                                // FOLD 1 (launches whose bound is 13..17 at two words): the filter plane's own distance,
                                // one popcount per word chained through v_bcnt's accumulator.  The OR-fold above lays
                                // column j on column j + 32: its popcount is ~24 of 32 for unrelated sequences at W = 2 —
                                // it stops rejecting at 13 (bound 14: 25.3 ms vs 11.7 this way, 10 000 queries x 10M aa)
                                // — and ~30 from W = 3 on, where it rejects up to a bound of ~24 and is the cheaper test.
                                // Chosen per launch: a per-query branch in this loop cost 30 % at bounds 8-12.
                                uint32_t t0 = nu, t1 = nu, t2 = nu, t3 = nu;
#pragma unroll
                                for (int w = 0; w < W; w++) {
                                    const uint32_t qv = qw[qslot(PQ, W, FP, w)];
                                    t0 += __builtin_popcount(s[t][FP * W + w].x ^ qv);
                                    t1 += __builtin_popcount(s[t][FP * W + w].y ^ qv);
                                    t2 += __builtin_popcount(s[t][FP * W + w].z ^ qv);
                                    t3 += __builtin_popcount(s[t][FP * W + w].w ^ qv);
                                }
                                any = t ? or3(or3(t0, t1, t2), t3, any) : (or3(t0, t1, t2) | t3);
                            } else {
                                const uint32_t t0 = __builtin_popcount(m0) + nu, t1 = __builtin_popcount(m1) + nu;
                                const uint32_t t2 = __builtin_popcount(m2) + nu, t3 = __builtin_popcount(m3) + nu;
                                any = t ? or3(or3(t0, t1, t2), t3, any) : (or3(t0, t1, t2) | t3);
                            }
                        }
                        // sign bit set <=> some lower bound <= bound
                        if (__ballot((int32_t)any < 0) != 0ull) {  // wave-uniform branch, rare
                            passes++;
                            read_record(rec, qw, 0, RV);
                            full_compare(qw, qc + i);
                        }
                    }
                }
                filter_on = passes * 4u <= nqc;
                // level 1 earns its keep only while it rejects most queries; like the prefilter as a whole it is
                // re-tried on every 16th chunk
                level1_on = l1 ? level1_passes * 2u <= nqc : (chunk_no & 15u) == 15u;
            } else {
                for (uint32_t i = 0; i < nqc; i++, rec += RV) {
                    uint32_t qw[RS];
                    read_record(rec, qw, 0, RV);
                    full_compare(qw, qc + i);
                }
            }
        }
        if (more) commit(buf ^ 1, qc + kChunk);
        __syncthreads();
        if (!SEED && a.hits) flush_rows(a, rs, buf);
    }
    if (!SEED) finish_rows(a);
}

// ---------------------------------------------------------------------------------------------
// Filter-plane-resident form of the scan.
//
// With the first-level bound a query usually costs one xor + one popcount per subject, so what is left to
// amortise is the per-query work (LDS read, OR tree, compare, branches) — over more subjects per lane — and
// the registers that hold planes the fast path never touches.  Here a lane keeps ONLY the filter plane's
// W words of its 4*T subjects (T wave tiles per wave) and
//   * levels 1 and 2 of the prefilter run from those registers, exactly as in scan_kernel;
//   * when a query survives them (rare), the other planes of the tiles that survived are fetched from
//     L2/HBM for that one comparison;
//   * when the prefilter stops paying for a wave (dense neighbourhoods), the wave walks the 64-query chunk
//     once per tile with that tile's planes loaded into registers: the plain full comparison.
// Sparse-hit scans then stream only W/(PS*W) of the store from HBM (1/5 for amino acids).
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int lazy_min_waves(int ps, int w, int t) {
    const int regs = t * w * 4 + ps * w * 4 + 44;  // filter words of T tiles + one tile's planes (transient) + working set
    return regs <= 64 ? 8 : regs <= 80 ? 6 : regs <= 96 ? 5 : regs <= 128 ? 4 : regs <= 168 ? 3 : 2;
}

// SUMFOLD (two-word stores, launches whose bound is 13..17): level 2 is the filter plane's own distance — one popcount per
// word chained through v_bcnt's accumulator — instead of the popcount of the words OR-ed together, which stops rejecting at 13
// (see scan_kernel's FOLD 1; here with 16 subjects per lane and only the filter plane resident).
template <int PS, int PQ, int W, int T, bool SEED, bool SUMFOLD = false>
__global__ __launch_bounds__(256, lazy_min_waves(PS, W, T)) void scan_lazy_kernel(const uint4 *__restrict__ planes,
                                                                                  const uint32_t *__restrict__ qrec,
                                                                                  ScanArgs a) {
    constexpr int RS = qrec_stride(PQ, W);
    constexpr int RV = RS / 4;
    constexpr int NV = (kChunk * RV + 255) / 256;
    constexpr int FP = filter_plane(PQ);
    static_assert(FP < PS && PS <= PQ, "the filter plane must be one the subjects store");
    constexpr int BS = bound_slot(W);
    constexpr int HV = (W + 1 + 3) / 4;
    constexpr bool kPair = SMAFA_AND_PAIR && W > 1;  // see scan_kernel
    __shared__ uint4 stage[2][kChunk * RV];
    __shared__ RowStage rs;
    int buf = 0;  // LDS buffer of the chunk being computed = parity of the row stage it appends to

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    if (tid == 0) rs.n[0] = rs.n[1] = 0;  // published by the barrier in front of the chunk loop
    const uint32_t wg_tile = blockIdx.x % a.n_wg_tiles;
    const uint32_t qblock = blockIdx.x / a.n_wg_tiles;
    const uint32_t tile0 = a.tile_begin + (wg_tile * kWgWaves + wave) * T;
    const bool active = tile0 < a.tile_end;
    const uint32_t q0 = a.q_begin + qblock * a.qb_size;
    const uint32_t q1 = min(q0 + a.qb_size, a.q_end);

    // filter-plane words of this lane's 4*T subjects; tile slots past the range hold a copy of tile_begin
    // (valid memory) and are ignored wherever rows could come out of them
    uint4 f[T][W];
    const bool stream_once = a.stream_once != 0;  // (wave-uniform)
    auto load_filter = [&](int t) {
        const bool live = tile0 + t < a.tile_end;
        const uint4 *src = planes + (size_t)(live ? tile0 + t : a.tile_begin) * (PS * W * 64) + lane;
#pragma unroll
        for (int w = 0; w < W; w++) f[t][w] = stream_once ? ld_nt(src + (FP * W + w) * 64) : src[(FP * W + w) * 64];
    };
#pragma unroll
    for (int t = 0; t < T; t++) load_filter(t);
    uint4 pre[NV];
    auto fetch = [&](uint32_t qc) {
        const uint32_t nqc = min((uint32_t)kChunk, q1 - qc);
        const uint4 *src = reinterpret_cast<const uint4 *>(qrec + (size_t)qc * RS);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const uint32_t idx = tid + v * 256;
            if (idx < nqc * RV) {
                uint4 x = src[idx];
                if (idx % RV == BS / 4) {
                    const uint32_t nu = ~(a.thr ? ld_relaxed(a.thr + qc + idx / RV) : a.thr0);
                    if ((BS & 3) == 0) x.x = nu;
                    else if ((BS & 3) == 1) x.y = nu;
                    else if ((BS & 3) == 2) x.z = nu;
                    else x.w = nu;
                }
                pre[v] = x;
            }
        }
    };
    auto commit = [&](int b, uint32_t qc) {
        const uint32_t nqc = min((uint32_t)kChunk, q1 - qc);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const uint32_t idx = tid + v * 256;
            if (idx < nqc * RV) stage[b][idx] = pre[v];
        }
    };
    auto read_record = [&](const uint4 *rec, uint32_t(&qw)[RS], int from, int to) {
#pragma unroll
        for (int v = from; v < to; v++) {
            const uint4 x = rec[v];
            qw[4 * v + 0] = x.x;
            qw[4 * v + 1] = x.y;
            qw[4 * v + 2] = x.z;
            qw[4 * v + 3] = x.w;
        }
    };
    auto load_tile = [&](uint32_t tile, uint4(&s)[PS * W]) {  // all planes of one wave tile: L2/HBM -> registers
        const uint4 *src = planes + (size_t)tile * (PS * W * 64) + lane;
#pragma unroll
        for (int i = 0; i < PS * W; i++) s[i] = src[i * 64];
    };
    // full comparison of one query against the 4 subjects this lane owns in `tile`
    auto full_compare = [&](const uint4(&s)[PS * W], uint32_t tile, const uint32_t(&qw)[RS], uint32_t q) {
        const uint32_t U = ~qw[BS];
        uint32_t d[4];
#pragma unroll
        for (int w = 0; w < W; w++) {
            uint32_t extra = 0;
#pragma unroll
            for (int p = PS; p < PQ; p++) extra |= qw[qslot(PQ, W, p, w)];
            uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
#pragma unroll
            for (int p = 0; p < PS; p++) {
                const uint4 v = s[p * W + w];
                const uint32_t qv = qw[qslot(PQ, W, p, w)];
                const bool first = p == 0 && PS == PQ;
                m0 = first ? (v.x ^ qv) : or_xor(m0, v.x, qv);
                m1 = first ? (v.y ^ qv) : or_xor(m1, v.y, qv);
                m2 = first ? (v.z ^ qv) : or_xor(m2, v.z, qv);
                m3 = first ? (v.w ^ qv) : or_xor(m3, v.w, qv);
            }
            d[0] = (w ? d[0] : 0u) + __builtin_popcount(m0);
            d[1] = (w ? d[1] : 0u) + __builtin_popcount(m1);
            d[2] = (w ? d[2] : 0u) + __builtin_popcount(m2);
            d[3] = (w ? d[3] : 0u) + __builtin_popcount(m3);
        }
        const uint32_t subj0 = tile * kWaveTile + lane * 4u;
        if (SEED) {
            uint32_t lo = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (subj0 + k < a.n_subjects) lo = min(lo, d[k]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
            if (lane == 0 && lo < U) atomicMin(a.thr + q, lo);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (d[k] <= U && subj0 + k < a.n_subjects) emit(a, rs, buf, q, subj0 + k, d[k]);
        }
    };
    // the filter plane of tile slot t folded over its words against the record's filter words
    auto fold = [&](int t, const uint32_t(&qw)[RS], uint32_t &m0, uint32_t &m1, uint32_t &m2, uint32_t &m3) {
        m0 = f[t][0].x ^ qw[0], m1 = f[t][0].y ^ qw[0];
        m2 = f[t][0].z ^ qw[0], m3 = f[t][0].w ^ qw[0];
#pragma unroll
        for (int w = 1; w < W; w++) {
            m0 = or_xor(m0, f[t][w].x, qw[qslot(PQ, W, FP, w)]);
            m1 = or_xor(m1, f[t][w].y, qw[qslot(PQ, W, FP, w)]);
            m2 = or_xor(m2, f[t][w].z, qw[qslot(PQ, W, FP, w)]);
            m3 = or_xor(m3, f[t][w].w, qw[qslot(PQ, W, FP, w)]);
        }
    };

    if (q0 < q1) {
        fetch(q0);
        commit(0, q0);
    }
    __syncthreads();
    bool filter_on = a.use_filter != 0;
    bool level1_on = true;
    uint32_t chunk_no = 0;
    for (uint32_t qc = q0; qc < q1; qc += kChunk, buf ^= 1, chunk_no++) {
        const uint32_t nqc = min((uint32_t)kChunk, q1 - qc);
        const bool more = qc + kChunk < q1;
        if (more) fetch(qc + kChunk);
        if (active) {
            const bool probe = a.use_filter && (filter_on || (chunk_no & 15u) == 0);
            if (probe) {
                const uint4 *rec = &stage[buf][0];
                uint32_t passes = 0, level1_passes = 0;
                // pinned to a scalar: hipcc cannot see through __ballot that these flags are wave-uniform
                const bool l1 = __builtin_amdgcn_readfirstlane((int)level1_on) != 0;
                for (uint32_t i = 0; i < nqc; i++, rec += RV) {
                    uint32_t qw[RS];
                    read_record(rec, qw, 0, HV);
                    const uint32_t nu = qw[BS];
                    bool go = true;
                    if (W > 1 && l1) {  // level 1: word 0 of the filter plane
                        uint32_t any1 = 0;
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            const uint32_t u0 = __builtin_popcount(f[t][0].x ^ qw[0]) + nu;
                            const uint32_t u1 = __builtin_popcount(f[t][0].y ^ qw[0]) + nu;
                            const uint32_t u2 = __builtin_popcount(f[t][0].z ^ qw[0]) + nu;
                            const uint32_t u3 = __builtin_popcount(f[t][0].w ^ qw[0]) + nu;
                            any1 = t ? or3(or3(u0, u1, u2), u3, any1) : (or3(u0, u1, u2) | u3);
                        }
                        go = __ballot((int32_t)any1 < 0) != 0ull;
                        level1_passes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(level1_passes + (go ? 1u : 0u)));
                    }
                    if (go) {  // level 2: the filter plane folded over its words, two subjects per popcount
                        uint32_t any = 0;
                        uint32_t tsign[T];
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            if (SUMFOLD) {
                                uint32_t t0 = nu, t1 = nu, t2 = nu, t3 = nu;
#pragma unroll
                                for (int w = 0; w < W; w++) {
                                    const uint32_t qv = qw[qslot(PQ, W, FP, w)];
                                    t0 += __builtin_popcount(f[t][w].x ^ qv);
                                    t1 += __builtin_popcount(f[t][w].y ^ qv);
                                    t2 += __builtin_popcount(f[t][w].z ^ qv);
                                    t3 += __builtin_popcount(f[t][w].w ^ qv);
                                }
                                tsign[t] = or3(t0, t1, t2) | t3;
                            } else {
                                uint32_t m0, m1, m2, m3;
                                fold(t, qw, m0, m1, m2, m3);
                                tsign[t] = kPair ? ((__builtin_popcount(m0 & m1) + nu) | (__builtin_popcount(m2 & m3) + nu))
                                                 : (or3(__builtin_popcount(m0) + nu, __builtin_popcount(m1) + nu,
                                                        __builtin_popcount(m2) + nu) |
                                                    (__builtin_popcount(m3) + nu));
                            }
                            any |= tsign[t];
                        }
                        if (__ballot((int32_t)any < 0) != 0ull) {  // level 3, rare: fetch planes, compare exactly
                            passes++;
                            read_record(rec, qw, HV, RV);
                            uint32_t live = 0;  // wave-uniform mask of the tiles that survived level 2
#pragma unroll
                            for (int t = 0; t < T; t++)
                                if (__ballot((int32_t)tsign[t] < 0) != 0ull) live |= 1u << t;
                            live = (uint32_t)__builtin_amdgcn_readfirstlane((int)live);
                            if (kPair) {
                                // The two-subject AND passes whenever ONE of the pair is close on this plane (a
                                // relative of the query next to anything): before fetching a tile, look at its
                                // subjects one by one.  Rare path: recomputing the masks is cheaper than keeping them.
#pragma unroll
                                for (int t = 0; t < T; t++) {
                                    if ((live >> t) & 1u) {
                                        uint32_t m0, m1, m2, m3;
                                        fold(t, qw, m0, m1, m2, m3);
                                        const uint32_t each = or3(__builtin_popcount(m0) + nu, __builtin_popcount(m1) + nu,
                                                                  __builtin_popcount(m2) + nu) |
                                                              (__builtin_popcount(m3) + nu);
                                        if (__ballot((int32_t)each < 0) == 0ull) live &= ~(1u << t);
                                    }
                                }
                                live = (uint32_t)__builtin_amdgcn_readfirstlane((int)live);
                            }
                            while (live) {  // ONE copy of the comparison code, whatever T is
                                const uint32_t t = (uint32_t)__builtin_ctz(live);
                                live &= live - 1;
                                if (tile0 + t < a.tile_end) {
                                    uint4 s[PS * W];
                                    load_tile(tile0 + t, s);
                                    full_compare(s, tile0 + t, qw, qc + i);
                                }
                            }
                        }
                    }
                }
                filter_on = passes * 4u <= nqc;
                level1_on = l1 ? level1_passes * 2u <= nqc : (chunk_no & 15u) == 15u;
            } else {
                // dense neighbourhoods: one pass over the chunk per tile, that tile's planes in registers
                for (int t = 0; t < T; t++) {
                    if (tile0 + t >= a.tile_end) break;
                    uint4 s[PS * W];
                    load_tile(tile0 + t, s);
                    const uint4 *rec = &stage[buf][0];
                    for (uint32_t i = 0; i < nqc; i++, rec += RV) {
                        uint32_t qw[RS];
                        read_record(rec, qw, 0, RV);
                        full_compare(s, tile0 + t, qw, qc + i);
                    }
                }
            }
        }
        if (more) commit(buf ^ 1, qc + kChunk);
        __syncthreads();
        if (!SEED && a.hits) flush_rows(a, rs, buf);
    }
    if (!SEED) finish_rows(a);
}

// ---------------------------------------------------------------------------------------------
// The scan of a SORTED store: zone level in front of levels 1-3.
//
// Big appends are sorted by their filter words when they are laid out (pack_rows in engine.hip), so the 256 subjects of
// a wave tile share their leading filter bits, and zone[tile] = {those bits, which bits they are} for filter words 0
// and 1 (zone_kernel).  For a query, popcount((Q_f ^ common) & shared) counts columns in which EVERY subject of the
// tile mismatches: a lower bound on all 256 distances at once.  The wave evaluates it for 64 queries at a time — lane i
// takes query i of the chunk, ~6 VALU ops per (64 queries x tile) — and only the (query, tile) pairs that survive enter
// level 1, one query at a time, the query's word and bound broadcast out of lane i by v_readlane.  On a 10M-row store
// a tile shares ~15 bits: at bound 5 about 1 pair in 6 survives, at bound 3 about 1 in 50.
// A lane keeps ONLY word 0 of the filter plane of its 16 subjects (4 wave tiles): that is all level 1 looks at; the
// other filter words (level 2, ~1 % of the survivors) and the other planes (level 3) come from L2/HBM when needed.
// Exact like every other level: nothing is skipped unless a lower bound already exceeds the query's bound.
// Where the prefilter stops paying for a wave (dense neighbourhoods) it falls back to the plain comparison with one
// tile's planes in registers, exactly like scan_lazy_kernel.
// ---------------------------------------------------------------------------------------------
// A qualifying pair appended straight to the caller's list: the lanes that got here together take consecutive rows with ONE
// global atomic (v_mbcnt ranks them).  For kernels whose waves never meet at a barrier (scan_zone_kernel<.., DIRECT>): rows
// are rare there (a fixed tight bound on a sorted store), so the workgroup-level LDS stage has nothing to batch.
__device__ __forceinline__ void emit_direct(const ScanArgs &a, uint32_t q, uint32_t pos, uint32_t dist) {
    smafa_hit h;
    h.query = q;
    h.subject = a.order[pos];
    h.dist = dist;
    const unsigned long long together = __ballot(1);
    const int first = __builtin_ctzll(together);
    unsigned long long g = 0;
    if ((int)__lane_id() == first) g = atomicAdd(a.count, (unsigned long long)__builtin_popcountll(together));
    g = shfl_u64(g, first) + lanes_below(together);
    if (g < a.cap) a.hits[g] = h;
}

#ifndef SMAFA_ZONE_TILES
#define SMAFA_ZONE_TILES 4
#endif
constexpr int kZoneTiles = SMAFA_ZONE_TILES;
// ... per shape and form.  The UNSTAGED five-plane two-word kernel (60-column amino acids, fixed bound) takes 2 tiles per wave at
// 7 waves per SIMD (72 VGPRs): without LDS staging and barriers there is little per-chunk work left to amortise over more tiles,
// and every resident wave more hides more of the survivor loop's dependent slow-class chain.  Same box, ms per launch, 10M x 10k /
// 50M x 125k (profiles/r04_zone_variants.txt): 5 waves x 4 tiles 1.795 / 53.5, 5 x 6 1.77 / 52.0, 6 x 3 1.71 / 50.6, 6 x 2 1.72 /
// 51.5, **7 x 2 1.68 / 49.3**, 7 x 3 1.76 / 64.6 (spills), 8 x 2 1.78 / 78.4 (spills), 4 x 6 1.95, 6 x 4 2.01.  Nucleotides keep 4
// tiles (3: 2.80, 6: 3.12 vs 2.70 ms).  SMAFA_ZONE_TILES != 4 overrides for every shape.
__host__ __device__ constexpr int zone_tiles(int ps, int w, bool direct) {
    return SMAFA_ZONE_TILES != 4 ? SMAFA_ZONE_TILES : (direct && ps == 5 && w == 2 ? 2 : 4);
}
#ifndef SMAFA_ZONE_WG_WAVES
#define SMAFA_ZONE_WG_WAVES 2
#endif
constexpr int kZoneWgWaves = SMAFA_ZONE_WG_WAVES;
#ifndef SMAFA_FEW_TILES
#define SMAFA_FEW_TILES 4
#endif
constexpr int kFewTiles = SMAFA_FEW_TILES;  // wave tiles per wave in scan_zone_few_kernel

// Waves per SIMD the register budget must allow.  The survivor loop is a chain of dependent slow-class instructions
// (v_readlane -> scalar-operand xor -> bcnt -> or -> cmp -> branch), so one more resident wave pays as long as the hot
// path does not spill: measured (profiles/r02_zone_variants.txt) aa 60 columns 2.55 -> 2.45 ms at 5 waves (3.06 at 6:
// spills), nt 60 columns 6.38 -> 5.91 ms at 6 waves.  What caps it is the dense fallback's tile (PS * W vectors).
#ifndef SMAFA_ZONE_WAVES_10
#define SMAFA_ZONE_WAVES_10 5  // waves per SIMD asked for where a tile holds 5..10 vectors (aa 60 columns: 10)
#endif
#ifndef SMAFA_ZONE_SGPR_ZONE
#define SMAFA_ZONE_SGPR_ZONE 1  // 1: word 0's zone words of the wave's tiles live in scalar registers (those of tiles 1.. end up
                                // in spilled VGPR lanes); 0: they are read out of their VGPR lane where they are used
#endif
#ifndef SMAFA_ZONE_VGPR_MASK
#define SMAFA_ZONE_VGPR_MASK 6  // word 0's zone masks also in vector registers for stores of up to this many vectors per tile
                                // (nucleotides: -2.3 %; the amino-acid kernel has no register to spare: profiles/r03_zone_variants.txt)
#endif
#ifndef SMAFA_ZONE_OPAQUE_BUF
#define SMAFA_ZONE_OPAQUE_BUF 1  // 1: the rare levels see the chunk parity through an opaque copy (their LDS addresses are formed there)
#endif
#ifndef SMAFA_ZONE_FULL_DMA
#define SMAFA_ZONE_FULL_DMA 1  // 1: a staged chunk is always 64 whole records (the record array is padded by one chunk)
#endif
#ifndef SMAFA_ZONE_NLIVE
#define SMAFA_ZONE_NLIVE 1  // 1: "tile slot t is inside the range" is ONE scalar (the number of live slots) instead of T lane masks
#endif
#ifndef SMAFA_ZONE_WAVES_4
#define SMAFA_ZONE_WAVES_4 6  // waves per SIMD asked for where a tile holds up to 4 vectors (the 2-bit nucleotide store)
#endif
#ifndef SMAFA_ZONE_WAVES_6
#define SMAFA_ZONE_WAVES_6 6  // ... for the three-plane two-word shape (nucleotides with N at 60 columns: 6 vectors per tile): 80 VGPRs +
                              // 8 bytes of scratch instead of 82 / none, one more resident wave: 3.09 -> 2.98 ms (7: the same;
                              // profiles/r04_zone_variants.txt)
#endif
#ifndef SMAFA_ZONE_WAVES_DIRECT_AA
#define SMAFA_ZONE_WAVES_DIRECT_AA 7  // the unstaged five-plane two-word kernel (see zone_tiles)
#endif
__host__ __device__ constexpr int zone_min_waves(int ps, int w, bool direct = false) {
    return (direct && ps == 5 && w == 2 && SMAFA_ZONE_TILES == 4) ? SMAFA_ZONE_WAVES_DIRECT_AA
           : ps * w <= 4 ? SMAFA_ZONE_WAVES_4 : (ps == 3 && w == 2) ? SMAFA_ZONE_WAVES_6 : ps * w <= 10 ? SMAFA_ZONE_WAVES_10 : 4;
}

// FIXED: one bound for every query (a.thr == NULL, the plain --max-divergence scan).  The chunks are then staged by
// LDS-DMA (global_load_lds_dwordx4: no register hop — the prefetch registers of the other form were being spilled
// across every chunk, 1.2 GB of scratch writes per launch), nothing has to be merged into the records, and ~bound is a
// scalar.  !FIXED (per-query bounds that tighten while the scan runs): register prefetch, ~bound merged at fetch time.
// DIRECT (FIXED only): no LDS staging and no barrier at all.  Lane i loads the head of query i of the chunk straight from
// the record array (8 bytes, the next chunk's one chunk ahead), the rare levels read the record through scalar loads, rows go
// straight to the list (emit_direct).  The staged form's waves wait for each other at a barrier per chunk, and the compiler
// guards every LDS read of a staged record with s_waitcnt vmcnt(0) — it cannot tell the buffer being read from the one the
// next chunk's LDS-DMA is landing in — so each wave stalled on its own prefetch once per chunk.
template <int PS, int PQ, int W, bool FIXED, bool DIRECT = false>
__global__ __launch_bounds__(kZoneWgWaves * 64, zone_min_waves(PS, W, DIRECT)) void scan_zone_kernel(const uint4 *__restrict__ planes,
                                                           const uint32_t *__restrict__ qrec, ScanArgs a) {
    static_assert(FIXED || !DIRECT, "per-query bounds keep the staged form");
    constexpr int T = zone_tiles(PS, W, DIRECT);
    constexpr int RS = qrec_stride(PQ, W);
    constexpr int RV = RS / 4;
    constexpr int WGW = kZoneWgWaves;  // waves per workgroup
    constexpr int NT = WGW * 64;
    constexpr int NV = (kChunk * RV + NT - 1) / NT;
    constexpr int FP = filter_plane(PQ);
    constexpr int BS = bound_slot(W);
    constexpr bool kPair = SMAFA_AND_PAIR && W > 1;
    __shared__ uint4 stage[2][kChunk * RV];
    __shared__ uint32_t nu_lds[2][kChunk];  // !FIXED: ~bound of the staged queries
    __shared__ RowStageT<(WGW == 1 ? 64 : kStageRows)> rs;
    int buf = 0;  // LDS buffer of the chunk being computed = parity of the row stage it appends to

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    if (tid == 0) rs.n[0] = rs.n[1] = 0;  // published by the barrier in front of the chunk loop
    const uint32_t wg_tile = blockIdx.x % a.n_wg_tiles;
    const uint32_t qblock = blockIdx.x / a.n_wg_tiles;
    const uint32_t tile0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.tile_begin + (wg_tile * WGW + wave) * T));
    const bool active = tile0 < a.tile_end;
    // tile slots of this wave that lie inside the range: one scalar (T lane masks, kept across the chunk loop, were being
    // spilled into VGPR lanes)
    const uint32_t n_live = active ? min((uint32_t)T, a.tile_end - tile0) : 0u;
    const uint32_t q0 = a.q_begin + qblock * a.qb_size;
    const uint32_t q1 = min(q0 + a.qb_size, a.q_end);

    // word 0 of the filter plane of this lane's 16 subjects; tile slots past the range copy tile_begin (valid memory)
    uint4 f0[T];
    auto load_filter = [&]() {
#pragma unroll
        for (int t = 0; t < T; t++) {
            const bool live = SMAFA_ZONE_NLIVE ? (uint32_t)t < n_live : tile0 + t < a.tile_end;
            f0[t] = planes[(size_t)(live ? tile0 + t : a.tile_begin) * (PS * W * 64) + (FP * W) * 64 + lane];
        }
    };
    load_filter();
    // zone words of this wave's tiles: lane t keeps tile slot t's four words for the whole kernel (4 VGPRs; sixteen
    // scalar registers ran out, and a load per tile and chunk sits on the critical path); v_readlane hands them out
    uint4 vz = make_uint4(0u, 0u, 0u, 0u);
    if (lane < (uint32_t)T && tile0 + lane < a.tile_end) vz = a.zone[tile0 + lane];
    // word 0's zone words go to scalar registers once (they are used by every chunk); word 1 rarely shares anything
    // (a 10M-row store spends its ~15 shared bits in word 0): its part of the check runs only for waves that need it
    uint32_t zc0[T], zm0[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        zc0[t] = (uint32_t)__builtin_amdgcn_readlane((int)vz.x, t);
        zm0[t] = (uint32_t)__builtin_amdgcn_readlane((int)vz.y, t);
    }
    // the masks also as (wave-uniform) VECTOR registers where there is room: (q ^ c) & m is then ONE v_bitop3 with a single
    // scalar operand — otherwise the mask is moved to a vector register in front of every use
    constexpr bool kMaskInVgpr = PS * W <= SMAFA_ZONE_VGPR_MASK;
    uint32_t zmv[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        zmv[t] = zm0[t];
        if (kMaskInVgpr) asm volatile("v_mov_b32 %0, %1" : "=v"(zmv[t]) : "s"(zm0[t]));
    }
    const bool zone_w1 = W > 1 && __ballot(vz.w != 0u) != 0ull;

    // Query chunks are staged by LDS-DMA (global_load_lds_dwordx4: global -> LDS with no register hop; a register
    // prefetch was being spilled across every chunk).  The records carry no bound in this kernel: FIXED takes the scalar
    // ~thr0, !FIXED keeps the chunk's per-query ~bound in nu_lds (read from thr at the top of the previous chunk, one
    // register per thread in flight, written at its end).
    auto dma = [&](int b, uint32_t qc) {  // lands at wave base + lane * 16 (lane-linear)
        // SMAFA_ZONE_FULL_DMA: always the whole chunk — the record array is padded by a chunk (engine.hip qset_fill), the heads
        // of queries past the block are masked where they are read: no per-chunk count / compare / exec mask around the DMA
        const uint32_t nqc = SMAFA_ZONE_FULL_DMA ? (uint32_t)kChunk : min((uint32_t)kChunk, q1 - qc);
        const uint4 *src = reinterpret_cast<const uint4 *>(qrec + (size_t)qc * RS);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const uint32_t idx = tid + v * NT;
            if (idx < nqc * RV)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + idx),
                                                 (__attribute__((address_space(3))) void *)(&stage[b][wave * 64 + v * NT]), 16, 0, 0);
        }
    };
    const uint32_t nu0 = ~a.thr0;  // FIXED: every query's ~bound
    auto load_bound = [&](uint32_t qc) -> uint32_t {  // !FIXED: this thread's query of the chunk at qc
        return tid < min((uint32_t)kChunk, q1 - qc) ? ~ld_relaxed(a.thr + qc + tid) : 0u;
    };
    uint32_t nu_next = 0;
    auto read_record = [&](const uint4 *rec, uint32_t(&qw)[RS]) {
#pragma unroll
        for (int v = 0; v < RV; v++) {
            const uint4 x = rec[v];
            qw[4 * v + 0] = x.x;
            qw[4 * v + 1] = x.y;
            qw[4 * v + 2] = x.z;
            qw[4 * v + 3] = x.w;
        }
    };
    // exact comparison of one query against the 4 subjects this lane owns in `tile`, the tile's words streamed from
    // L2/HBM one 32-column word at a time (PS vectors live, not PS * W)
    auto stream_compare = [&](uint32_t tile, const uint32_t(&qw)[RS], uint32_t q, int parity) {
        const uint32_t U = ~qw[BS];
        const uint4 *src = planes + (size_t)tile * (PS * W * 64) + lane;
        uint32_t d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int w = 0; w < W; w++) {
            uint32_t extra = 0;
#pragma unroll
            for (int p = PS; p < PQ; p++) extra |= qw[qslot(PQ, W, p, w)];
            uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
#pragma unroll
            for (int p = 0; p < PS; p++) {
                const uint4 v = src[(p * W + w) * 64];
                const uint32_t qv = qw[qslot(PQ, W, p, w)];
                m0 = or_xor(m0, v.x, qv);
                m1 = or_xor(m1, v.y, qv);
                m2 = or_xor(m2, v.z, qv);
                m3 = or_xor(m3, v.w, qv);
            }
            d[0] += __builtin_popcount(m0);
            d[1] += __builtin_popcount(m1);
            d[2] += __builtin_popcount(m2);
            d[3] += __builtin_popcount(m3);
        }
        const uint32_t subj0 = tile * kWaveTile + lane * 4u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (d[k] <= U && subj0 + k < a.n_subjects) {
                if (DIRECT) emit_direct(a, q, subj0 + k, d[k]);
                else emit(a, rs, parity, q, subj0 + k, d[k]);
            }
    };
    // DIRECT: the head [f0 f1] of this lane's query of a chunk, straight from the record array (records are padded by a
    // chunk: lanes past the block read zeros or a neighbour's head, and are masked where the head is used)
    auto load_head = [&](uint32_t qc) -> uint2 {
        return *reinterpret_cast<const uint2 *>(qrec + (size_t)(qc + lane) * RS);
    };
    uint2 head_next = make_uint2(0u, 0u);

    if (!DIRECT) {
        if (q0 < q1) {
            dma(0, q0);
            if (!FIXED && tid < (uint32_t)kChunk) nu_lds[0][tid] = load_bound(q0);
        }
        // The DMA'd chunk is published by the barrier: every wave waits for its own global_load_lds first (the barrier's
        // fence does not have to: gfx950's s_barrier has no implicit vmcnt wait).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else if (q0 < q1) {
        head_next = load_head(q0);
    }
    bool filter_on = a.use_filter != 0;
    uint32_t chunk_no = 0;
    for (uint32_t qc = q0; qc < q1; qc += kChunk, buf ^= 1, chunk_no++) {
        const uint32_t nqc = min((uint32_t)kChunk, q1 - qc);
        const bool more = qc + kChunk < q1;
        const uint2 head_cur = head_next;
        if (more) {  // in flight while this chunk is computed
            if (DIRECT) {
                head_next = load_head(qc + kChunk);
            } else {
                dma(buf ^ 1, qc + kChunk);  // every wave passed the barrier that ended the last use of that buffer
                if (!FIXED) nu_next = load_bound(qc + kChunk);
            }
        }
        if (active) {
            const bool probe = a.use_filter && (filter_on || (chunk_no & 15u) == 0);
            if (probe) {
                uint32_t passes = 0;  // (query, tile) pairs of this chunk that needed the exact comparison
                // ---- zone level: lane i holds [f0 f1 ~bound ..] of query i of the chunk
                uint4 head = make_uint4(0u, 0u, 0u, 0u);  // lanes past the chunk: ~bound = 0 never passes
                if (DIRECT) head.x = head_cur.x, head.y = head_cur.y;
                else if (lane < nqc) head = stage[buf][lane * RV];
                const uint32_t hq0 = head.x, hq1 = W > 1 ? head.y : 0u;
                const uint32_t hnu = lane < nqc ? (FIXED ? nu0 : nu_lds[buf][lane]) : 0u;
                // (opaque copy: the comparisons against it are redone per chunk — one s_cmp each — instead of being hoisted
                // out of the chunk loop as T lane masks that then live in spilled VGPR lanes)
                uint32_t nl = n_live;
                asm volatile("" : "+s"(nl));
                if (!SMAFA_ZONE_SGPR_ZONE) asm volatile("" : "+v"(vz.x), "+v"(vz.y));  // (read out of vz where they are used)
                // Tile by tile, unrolled (the rare levels' address math stays inside their branch thanks to the opaque
                // tile number below; a run-time tile loop cost 4 % on aa and 27 % on nt at bound 3 in per-tile
                // bookkeeping — profiles/r02_zone_variants.txt).
                // (Round 3: the body is a function of the compile-time tile slot, called under `if (nl > t)` — as a loop with a
                // `break` and a `continue` the compiler threaded a state variable through the four bodies: ~10 scalar
                // instructions of pure control flow between two tiles, which counts where few queries survive a tile —
                // nucleotides at bound 3: scalar instructions 1.5 per 1024 pairs against 1.7 vector ones.)
                auto do_tile = [&](auto slot) {
                    constexpr uint32_t t = decltype(slot)::value;
                    const uint32_t tile = tile0 + t;
                    const uint32_t zc = SMAFA_ZONE_SGPR_ZONE ? zc0[t] : (uint32_t)__builtin_amdgcn_readlane((int)vz.x, (int)t);
                    const uint32_t zm = SMAFA_ZONE_SGPR_ZONE ? zm0[t] : (uint32_t)__builtin_amdgcn_readlane((int)vz.y, (int)t);
                    uint32_t u = __builtin_popcount((hq0 ^ zc) & (kMaskInVgpr ? zmv[t] : zm)) + hnu;
                    if (zone_w1) {
                        const uint32_t zc1 = (uint32_t)__builtin_amdgcn_readlane((int)vz.z, (int)t);
                        const uint32_t zm1 = (uint32_t)__builtin_amdgcn_readlane((int)vz.w, (int)t);
                        u += __builtin_popcount((hq1 ^ zc1) & zm1);
                    }
                    unsigned long long m = __ballot((int32_t)u < 0);  // queries of the chunk this tile cannot exclude
                    const uint4 ft = f0[t];
                    while (m != 0ull) {
                        const int i = __builtin_ctzll(m);
                        asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(i));  // m &= ~(1 << i) in ONE scalar op (m & (m - 1): three)
                        // ---- level 1: word 0 of the filter plane.  The query's word and ~bound come out of lane i as
                        // scalar operands.  Measured alternatives (profiles/r02_zone_variants.txt): two survivors per
                        // iteration 13 % slower; the word via an LDS broadcast read (all-VGPR xors) 4 % slower, 16 %
                        // slower when software-pipelined; the word copied to a VGPR first (v_mov, all-VGPR xors): no change.
                        const uint32_t q0w = (uint32_t)__builtin_amdgcn_readlane((int)hq0, i);
                        const uint32_t nu = FIXED ? nu0 : (uint32_t)__builtin_amdgcn_readlane((int)hnu, i);
                        const uint32_t u0 = __builtin_popcount(ft.x ^ q0w) + nu;
                        const uint32_t u1 = __builtin_popcount(ft.y ^ q0w) + nu;
                        const uint32_t u2 = __builtin_popcount(ft.z ^ q0w) + nu;
                        const uint32_t u3 = __builtin_popcount(ft.w ^ q0w) + nu;
                        if (__ballot((int32_t)(or3(u0, u1, u2) | u3) < 0) == 0ull) continue;
                        // the rare levels get the tile number through an opaque copy: otherwise the compiler hoists
                        // their ~25 address computations out of this loop into the per-tile path every chunk pays
                        // for (6 % of the launch)
                        uint32_t tile_r = tile;
                        asm volatile("" : "+s"(tile_r));
#if SMAFA_ZONE_OPAQUE_BUF
                        // ... and the chunk parity likewise: the LDS addresses of the staged record and of the row stage
                        // (five scalar instructions) were being formed in front of EVERY tile's survivor loop
                        int buf_r = __builtin_amdgcn_readfirstlane(buf);
                        asm volatile("" : "+s"(buf_r));
#define SMAFA_ZONE_BUF buf_r
#else
#define SMAFA_ZONE_BUF buf
#endif
                        // ---- level 2 (rare): the filter plane folded over all its words, words 1.. from L2/HBM
                        uint32_t qw[RS];
                        if (DIRECT) read_record(reinterpret_cast<const uint4 *>(qrec + (size_t)(qc + (uint32_t)i) * RS), qw);
                        else read_record(&stage[SMAFA_ZONE_BUF][(uint32_t)i * RV], qw);
                        qw[BS] = nu;  // the staged record carries no bound
                        uint32_t m0 = ft.x ^ qw[0], m1 = ft.y ^ qw[0], m2 = ft.z ^ qw[0], m3 = ft.w ^ qw[0];
                        if (W > 1) {
                            const uint4 *src = planes + (size_t)tile_r * (PS * W * 64) + (FP * W) * 64 + lane;
#pragma unroll
                            for (int w = 1; w < W; w++) {
                                const uint4 v = src[w * 64];
                                m0 = or_xor(m0, v.x, qw[qslot(PQ, W, FP, w)]);
                                m1 = or_xor(m1, v.y, qw[qslot(PQ, W, FP, w)]);
                                m2 = or_xor(m2, v.z, qw[qslot(PQ, W, FP, w)]);
                                m3 = or_xor(m3, v.w, qw[qslot(PQ, W, FP, w)]);
                            }
                            if (kPair) {  // two subjects per popcount first (see scan_kernel)
                                const uint32_t sign = (__builtin_popcount(m0 & m1) + nu) | (__builtin_popcount(m2 & m3) + nu);
                                if (__ballot((int32_t)sign < 0) == 0ull) continue;
                            }
                            const uint32_t each = or3(__builtin_popcount(m0) + nu, __builtin_popcount(m1) + nu,
                                                      __builtin_popcount(m2) + nu) |
                                                  (__builtin_popcount(m3) + nu);
                            if (__ballot((int32_t)each < 0) == 0ull) continue;
                        }
                        // ---- level 3: all planes, exactly
                        passes++;
                        stream_compare(tile_r, qw, qc + (uint32_t)i, SMAFA_ZONE_BUF);
#undef SMAFA_ZONE_BUF
                    }
                };
                static_assert(T <= 8, "tile slots are spelled out below");
                if (SMAFA_ZONE_NLIVE ? nl > 0u : tile0 + 0u < a.tile_end) do_tile(std::integral_constant<uint32_t, 0>{});
                if (T > 1 && (SMAFA_ZONE_NLIVE ? nl > 1u : tile0 + 1u < a.tile_end)) do_tile(std::integral_constant<uint32_t, (T > 1 ? 1 : 0)>{});
                if (T > 2 && (SMAFA_ZONE_NLIVE ? nl > 2u : tile0 + 2u < a.tile_end)) do_tile(std::integral_constant<uint32_t, (T > 2 ? 2 : 0)>{});
                if (T > 3 && (SMAFA_ZONE_NLIVE ? nl > 3u : tile0 + 3u < a.tile_end)) do_tile(std::integral_constant<uint32_t, (T > 3 ? 3 : 0)>{});
                if (T > 4 && (SMAFA_ZONE_NLIVE ? nl > 4u : tile0 + 4u < a.tile_end)) do_tile(std::integral_constant<uint32_t, (T > 4 ? 4 : 0)>{});
                if (T > 5 && (SMAFA_ZONE_NLIVE ? nl > 5u : tile0 + 5u < a.tile_end)) do_tile(std::integral_constant<uint32_t, (T > 5 ? 5 : 0)>{});
                if (T > 6 && (SMAFA_ZONE_NLIVE ? nl > 6u : tile0 + 6u < a.tile_end)) do_tile(std::integral_constant<uint32_t, (T > 6 ? 6 : 0)>{});
                if (T > 7 && (SMAFA_ZONE_NLIVE ? nl > 7u : tile0 + 7u < a.tile_end)) do_tile(std::integral_constant<uint32_t, (T > 7 ? 7 : 0)>{});
                // an exact comparison from L2 costs ~60 VALU per (query, tile) pair, the dense walk below ~25 for EVERY
                // pair of the chunk (4 tiles x nqc): switch when more than ~2/5 of the pairs got that far
                filter_on = passes * 5u <= nqc * 8u;
            } else {
                // dense neighbourhoods: one pass over the chunk per tile, that tile's planes in registers
                for (uint32_t t = 0; t < (uint32_t)T; t++) {
                    const uint32_t tile = tile0 + t;
                    if (SMAFA_ZONE_NLIVE ? t >= n_live : tile >= a.tile_end) break;
                    uint4 s[PS * W];
                    uint32_t tile_d = tile;  // opaque: the walk's addresses are formed here, not kept (spilled) from the prologue
                    asm volatile("" : "+s"(tile_d));
                    const uint4 *src = planes + (size_t)(SMAFA_ZONE_NLIVE ? tile_d : tile) * (PS * W * 64) + lane;
#pragma unroll
                    for (int i = 0; i < PS * W; i++) s[i] = src[i * 64];
                    const uint4 *rec = DIRECT ? reinterpret_cast<const uint4 *>(qrec + (size_t)qc * RS) : &stage[buf][0];
                    for (uint32_t i = 0; i < nqc; i++, rec += RV) {
                        uint32_t qw[RS];
                        read_record(rec, qw);
                        const uint32_t U = FIXED ? a.thr0 : ~nu_lds[buf][i];
                        uint32_t d[4];
#pragma unroll
                        for (int w = 0; w < W; w++) {
                            uint32_t extra = 0;
#pragma unroll
                            for (int p = PS; p < PQ; p++) extra |= qw[qslot(PQ, W, p, w)];
                            uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
#pragma unroll
                            for (int p = 0; p < PS; p++) {
                                const uint4 v = s[p * W + w];
                                const uint32_t qv = qw[qslot(PQ, W, p, w)];
                                m0 = or_xor(m0, v.x, qv);
                                m1 = or_xor(m1, v.y, qv);
                                m2 = or_xor(m2, v.z, qv);
                                m3 = or_xor(m3, v.w, qv);
                            }
                            d[0] = (w ? d[0] : 0u) + __builtin_popcount(m0);
                            d[1] = (w ? d[1] : 0u) + __builtin_popcount(m1);
                            d[2] = (w ? d[2] : 0u) + __builtin_popcount(m2);
                            d[3] = (w ? d[3] : 0u) + __builtin_popcount(m3);
                        }
                        const uint32_t subj0 = tile * kWaveTile + lane * 4u;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (d[k] <= U && subj0 + k < a.n_subjects) {
                                if (DIRECT) emit_direct(a, qc + i, subj0 + k, d[k]);
                                else emit(a, rs, buf, qc + i, subj0 + k, d[k]);
                            }
                    }
                }
                load_filter();  // not kept across the walk (its registers hold the tile meanwhile): fetched again
            }
        }
        if (!DIRECT) {
            if (more && !FIXED && tid < (uint32_t)kChunk) nu_lds[buf ^ 1][tid] = nu_next;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next chunk's DMA has landed
            __syncthreads();
            if (a.hits) flush_rows(a, rs, buf);
        }
    }
    finish_rows(a);
}

// ---------------------------------------------------------------------------------------------
// Up to 64 queries against a sorted store (north_star's "each query is broadcast against all subjects", one query or
// a handful per pass).  Nothing is staged: lane i loads the head of query i straight from the record array, the wave
// checks its T tiles' zone words against all (up to) 64 queries at once, and only the tiles that some query survives
// are fetched at all — one pass reads the zone array (16 B per 256 subjects) plus the filter words of the surviving
// tiles, instead of streaming the whole filter plane.  Levels 1-3 as in scan_lazy_kernel; the rare levels read the
// query record from global memory.
// ---------------------------------------------------------------------------------------------
template <int PS, int PQ, int W>
__global__ __launch_bounds__(256) void scan_zone_few_kernel(const uint4 *__restrict__ planes,
                                                            const uint32_t *__restrict__ qrec, ScanArgs a) {
    constexpr int T = kFewTiles;
    constexpr int RS = qrec_stride(PQ, W);
    constexpr int RV = RS / 4;
    constexpr int FP = filter_plane(PQ);
    constexpr int BS = bound_slot(W);
    constexpr bool kPair = SMAFA_AND_PAIR && W > 1;
    __shared__ RowStage rs;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    if (tid == 0) rs.n[0] = rs.n[1] = 0;
    __syncthreads();
    const uint32_t tile0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.tile_begin + (blockIdx.x * kWgWaves + wave) * T));
    const uint32_t nq = a.q_end - a.q_begin;  // <= 64: the host picks this kernel
    if (tile0 < a.tile_end) {
        uint4 vz = make_uint4(0u, 0u, 0u, 0u);  // lane t: zone words of tile slot t
        if (lane < (uint32_t)T && tile0 + lane < a.tile_end) vz = a.zone[tile0 + lane];
        uint4 head = make_uint4(0u, 0u, 0u, 0u);  // lane i: [f0 f1 bound ..] of query i; lanes past nq never pass
        uint32_t hnu = 0;
        if (lane < nq) {
            head = reinterpret_cast<const uint4 *>(qrec + (size_t)(a.q_begin + lane) * RS)[0];
            hnu = ~(a.thr ? ld_relaxed(a.thr + a.q_begin + lane) : a.thr0);
        }
        const uint32_t hq0 = head.x, hq1 = W > 1 ? head.y : 0u;
        unsigned long long pass[T];
        uint4 f0[T];  // word 0 of the filter plane of the tiles some query survives
#pragma unroll
        for (int t = 0; t < T; t++) {
            const uint32_t zc0 = (uint32_t)__builtin_amdgcn_readlane((int)vz.x, t);
            const uint32_t zm0 = (uint32_t)__builtin_amdgcn_readlane((int)vz.y, t);
            uint32_t u = __builtin_popcount((hq0 ^ zc0) & zm0) + hnu;
            if (W > 1) {
                const uint32_t zc1 = (uint32_t)__builtin_amdgcn_readlane((int)vz.z, t);
                const uint32_t zm1 = (uint32_t)__builtin_amdgcn_readlane((int)vz.w, t);
                u += __builtin_popcount((hq1 ^ zc1) & zm1);
            }
            pass[t] = tile0 + t < a.tile_end ? __ballot((int32_t)u < 0) : 0ull;
            if (pass[t] != 0ull) {  // all surviving tiles' loads are in flight before the first is used
                f0[t] = planes[(size_t)(tile0 + t) * (PS * W * 64) + (FP * W) * 64 + lane];
            }
        }
#pragma unroll 1
        for (uint32_t t = 0; t < (uint32_t)T; t++) {
            unsigned long long m = 0ull;
            uint4 ft = f0[0];
#pragma unroll
            for (int k = 0; k < T; k++)
                if (t == (uint32_t)k) {
                    m = pass[k];
                    ft = f0[k];
                }
            const uint32_t tile = tile0 + t;
            while (m != 0ull) {
                const int i = __builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t q0w = (uint32_t)__builtin_amdgcn_readlane((int)hq0, i);
                const uint32_t nu = (uint32_t)__builtin_amdgcn_readlane((int)hnu, i);
                const uint32_t u0 = __builtin_popcount(ft.x ^ q0w) + nu;
                const uint32_t u1 = __builtin_popcount(ft.y ^ q0w) + nu;
                const uint32_t u2 = __builtin_popcount(ft.z ^ q0w) + nu;
                const uint32_t u3 = __builtin_popcount(ft.w ^ q0w) + nu;
                if (__ballot((int32_t)(or3(u0, u1, u2) | u3) < 0) == 0ull) continue;
                // levels 2 and 3 (rare): the whole record, from global memory
                const uint32_t q = a.q_begin + (uint32_t)i;
                uint32_t qw[RS];
                const uint4 *rec = reinterpret_cast<const uint4 *>(qrec + (size_t)q * RS);
#pragma unroll
                for (int v = 0; v < RV; v++) {
                    const uint4 x = rec[v];
                    qw[4 * v + 0] = x.x;
                    qw[4 * v + 1] = x.y;
                    qw[4 * v + 2] = x.z;
                    qw[4 * v + 3] = x.w;
                }
                qw[BS] = nu;  // the record's bound slot is filled in at staging time elsewhere; here from thr / thr0
                uint32_t m0 = ft.x ^ qw[0], m1 = ft.y ^ qw[0], m2 = ft.z ^ qw[0], m3 = ft.w ^ qw[0];
#pragma unroll
                for (int w = 1; w < W; w++) {  // filter words 1.. of the tile, from L2/HBM
                    const uint4 v = planes[(size_t)tile * (PS * W * 64) + (FP * W + w) * 64 + lane];
                    m0 = or_xor(m0, v.x, qw[qslot(PQ, W, FP, w)]);
                    m1 = or_xor(m1, v.y, qw[qslot(PQ, W, FP, w)]);
                    m2 = or_xor(m2, v.z, qw[qslot(PQ, W, FP, w)]);
                    m3 = or_xor(m3, v.w, qw[qslot(PQ, W, FP, w)]);
                }
                if (kPair) {
                    const uint32_t sign = (__builtin_popcount(m0 & m1) + nu) | (__builtin_popcount(m2 & m3) + nu);
                    if (__ballot((int32_t)sign < 0) == 0ull) continue;
                }
                if (W > 1) {
                    const uint32_t each = or3(__builtin_popcount(m0) + nu, __builtin_popcount(m1) + nu,
                                              __builtin_popcount(m2) + nu) |
                                          (__builtin_popcount(m3) + nu);
                    if (__ballot((int32_t)each < 0) == 0ull) continue;
                }
                // exact comparison against the 4 subjects this lane owns in the tile
                const uint4 *src = planes + (size_t)tile * (PS * W * 64) + lane;
                const uint32_t U = ~nu;
                uint32_t d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int w = 0; w < W; w++) {
                    uint32_t extra = 0;
#pragma unroll
                    for (int p = PS; p < PQ; p++) extra |= qw[qslot(PQ, W, p, w)];
                    uint32_t x0 = extra, x1 = extra, x2 = extra, x3 = extra;
#pragma unroll
                    for (int p = 0; p < PS; p++) {
                        const uint4 v = src[(p * W + w) * 64];
                        const uint32_t qv = qw[qslot(PQ, W, p, w)];
                        x0 = or_xor(x0, v.x, qv);
                        x1 = or_xor(x1, v.y, qv);
                        x2 = or_xor(x2, v.z, qv);
                        x3 = or_xor(x3, v.w, qv);
                    }
                    d[0] += __builtin_popcount(x0);
                    d[1] += __builtin_popcount(x1);
                    d[2] += __builtin_popcount(x2);
                    d[3] += __builtin_popcount(x3);
                }
                const uint32_t subj0 = tile * kWaveTile + lane * 4u;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (d[k] <= U && subj0 + k < a.n_subjects) emit(a, rs, 0, q, subj0 + k, d[k]);
            }
        }
    }
    __syncthreads();
    if (a.hits) flush_rows(a, rs, 0);
    finish_rows(a);
}

// ---------------------------------------------------------------------------------------------
// L > 64 (more than 2 words per plane) and L <= 32 (one word): the same three levels for any length.  A lane keeps
// the first FW <= 4 words of the filter plane of its 16 subjects (4 wave tiles); level 1 bounds word 0, level 2 the
// fold of the resident words (two subjects per popcount) — both exact lower bounds on the distance, as in
// scan_lazy_kernel.  The full comparison (level 3, rare while the bound is small) streams the
// surviving tile's planes from L2/HBM word by word and reads the query's words from LDS, so its registers do not
// grow with W; it stops as soon as every subject of the wave is past the bound.  W is a runtime argument: one
// instantiation per plane pair serves every length.  A chunk holds as many query records as fit 768 LDS vectors;
// the first vector of each record ([f0 f1 bound ..]) is staged a second time in a dense array, so the hot loop
// reads LDS at a compile-time stride whatever W is.
// ONE (W == 1): a single filter word prunes too little (15 % of the queries reach level 3 at L = 30, bound 5), so
// the second register word of a subject is word 0 of ANOTHER plane and level 2 bounds the two-plane fold — a
// mismatch in either plane is a mismatching column, so it is still a lower bound on the distance.
// ---------------------------------------------------------------------------------------------
constexpr int kWideTiles = 4;
constexpr int kWideStage = 768;  // uint4 per LDS buffer
// queries per tile load on the dense path: 4 * group distances live in registers next to the tile's planes
__host__ __device__ constexpr int wide_group(int ps) { return ps >= 5 ? 4 : ps == 3 ? 6 : 8; }
__host__ __device__ constexpr bool wide_fits(int planes, int words) { return qrec_stride(planes, words) / 4 <= kWideStage; }

template <int PS, int PQ, bool SEED, int FW, int WC>
__global__ __launch_bounds__(256, 4) void scan_wide_kernel(const uint4 *__restrict__ planes,
                                                           const uint32_t *__restrict__ qrec, ScanArgs a, uint32_t W_arg) {
    // WC > 0: the word count is a compile-time constant (33..128 columns): record offsets fold to immediates and the
    // dense walk keeps a tile's planes in registers, like scan_lazy_kernel; WC == 0: any W at run time
    const uint32_t W = WC > 0 ? (uint32_t)WC : W_arg;
    constexpr int T = kWideTiles;
    constexpr bool ONE = FW == 1;          // one-word store: the second register word is word 0 of another plane
    constexpr int NF = ONE ? 2 : FW;       // register words per subject
    constexpr int HV = NF == 4 ? 2 : 1;    // vectors of a record that hold them: [f0 f1 bound f2 | f3 ..]
    static_assert(FW >= 1 && FW <= 4, "1 (two planes of a one-word store) to 4 filter words resident");
    constexpr int kWideGroup = wide_group(PS);
    constexpr int NV = kWideStage / 256;
    constexpr int FP = filter_plane(PQ);
    constexpr int FP2 = FP == 0 ? 1 : 0;  // ONE: the second plane of level 2 (slot 2 of a one-word record)
    static_assert(FP < PS && FP2 < PS && PS <= PQ, "the filter planes must be ones the subjects store");
    const uint32_t RS = (uint32_t)qrec_stride(PQ, (int)W);
    const uint32_t RV = RS / 4;
    const uint32_t BS = (uint32_t)bound_slot((int)W);
    const uint32_t chunk = min((uint32_t)kChunk, (uint32_t)kWideStage / RV);  // >= 1: the host checks wide_fits()
    __shared__ uint4 stage[2][kWideStage];
    __shared__ uint4 heads[2][kChunk][HV];  // leading vector(s) of every staged record, at a compile-time stride
    __shared__ RowStage rs;
    int buf = 0;  // LDS buffer of the chunk being computed = parity of the row stage it appends to
    if (threadIdx.x == 0) rs.n[0] = rs.n[1] = 0;  // published by the barrier in front of the chunk loop

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t wg_tile = blockIdx.x % a.n_wg_tiles;
    const uint32_t qblock = blockIdx.x / a.n_wg_tiles;
    const uint32_t tile0 = a.tile_begin + (wg_tile * kWgWaves + wave) * T;
    const bool active = tile0 < a.tile_end;
    const size_t tile_stride = (size_t)PS * W * 64;

    uint4 f[T][NF];  // filter-plane words 0..FW-1; tile slots past the range copy tile_begin (valid memory)
    auto load_filter = [&]() {
#pragma unroll
        for (int t = 0; t < T; t++) {
            const bool live = tile0 + t < a.tile_end;
            const uint4 *src =
                planes + (size_t)(live ? tile0 + t : a.tile_begin) * tile_stride + (size_t)FP * W * 64 + lane;
            if (ONE) {
                f[t][0] = src[0];
                f[t][1] = planes[(size_t)(live ? tile0 + t : a.tile_begin) * tile_stride + (size_t)FP2 * 64 + lane];
            } else {
#pragma unroll
                for (int w = 0; w < NF; w++) f[t][w] = src[w * 64];
            }
        }
    };
    load_filter();
    // zone words of this wave's tiles (a.zone_on: a sorted store — see scan_zone_kernel): lane t holds tile slot t's
    uint4 vz = make_uint4(0u, 0u, 0u, 0u);
    const bool zone_on = !SEED && a.zone_on != 0;
    if (zone_on && lane < (uint32_t)T && tile0 + lane < a.tile_end) vz = a.zone[tile0 + lane];
    const uint32_t q0 = a.q_begin + qblock * a.qb_size;
    const uint32_t q1 = min(q0 + a.qb_size, a.q_end);

    uint4 pre[NV];
    auto fetch = [&](uint32_t qc) {
        const uint32_t nqc = min(chunk, q1 - qc);
        const uint4 *src = reinterpret_cast<const uint4 *>(qrec + (size_t)qc * RS);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const uint32_t idx = tid + v * 256;
            if (idx < nqc * RV) {
                uint4 x = src[idx];
                if (idx % RV == BS / 4) {
                    const uint32_t nu = ~(a.thr ? ld_relaxed(a.thr + qc + idx / RV) : a.thr0);
                    const uint32_t c = BS & 3u;
                    x.x = c == 0 ? nu : x.x;
                    x.y = c == 1 ? nu : x.y;
                    x.z = c == 2 ? nu : x.z;
                    x.w = c == 3 ? nu : x.w;
                }
                pre[v] = x;
            }
        }
    };
    auto commit = [&](int buf, uint32_t qc) {
        const uint32_t nqc = min(chunk, q1 - qc);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const uint32_t idx = tid + v * 256;
            if (idx < nqc * RV) {
                stage[buf][idx] = pre[v];
                if (idx % RV < (uint32_t)HV) heads[buf][idx / RV][idx % RV] = pre[v];
            }
        }
    };
    // report (or, in the seed pass, fold into the bound) the 4 subjects this lane owns in `tile`
    auto finish = [&](uint32_t tile, uint32_t q, uint32_t U, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
        const uint32_t subj0 = tile * kWaveTile + lane * 4u;
        const uint32_t d[4] = {d0, d1, d2, d3};
        if (SEED) {
            uint32_t lo = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (subj0 + k < a.n_subjects) lo = min(lo, d[k]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
            if (lane == 0 && lo < U) atomicMin(a.thr + q, lo);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (d[k] <= U && subj0 + k < a.n_subjects) emit(a, rs, buf, q, subj0 + k, d[k]);
        }
    };
    // level 3: full comparison of the query record at `rec` (LDS) against the 4 subjects this lane owns in `tile`
    auto wide_compare = [&](uint32_t tile, const uint32_t *rec, uint32_t q) {
        const uint32_t U = ~rec[BS];
        const uint4 *src = planes + (size_t)tile * tile_stride + lane;
        uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll 1
        for (uint32_t w = 0; w < W; w++) {
            uint32_t extra = 0;
#pragma unroll
            for (int p = PS; p < PQ; p++) extra |= rec[qslot(PQ, (int)W, p, (int)w)];
            uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
#pragma unroll
            for (int p = 0; p < PS; p++) {
                const uint4 v = src[((size_t)p * W + w) * 64];
                const uint32_t qv = rec[qslot(PQ, (int)W, p, (int)w)];
                m0 = or_xor(m0, v.x, qv);
                m1 = or_xor(m1, v.y, qv);
                m2 = or_xor(m2, v.z, qv);
                m3 = or_xor(m3, v.w, qv);
            }
            d0 += __builtin_popcount(m0);
            d1 += __builtin_popcount(m1);
            d2 += __builtin_popcount(m2);
            d3 += __builtin_popcount(m3);
            // distances only grow: once all 256 subjects of the wave are past the bound nothing can come out
            // (two-word steps with both loads in flight were measured: more registers, slower level 1)
            if (__ballot(min(min(d0, d1), min(d2, d3)) <= U) == 0ull) return;
        }
        finish(tile, q, U, d0, d1, d2, d3);
    };
    // Dense neighbourhoods (the prefilter stopped paying for this wave): every pair gets the exact comparison.
    // Streaming a tile from L2 once per query would cost P*W KB per (tile, query); instead a tile's words are
    // loaded once per group of kWideGroup queries and the group's 4 * kWideGroup distances live in registers.
    auto dense_walk = [&](const uint32_t *rec0, uint32_t nqc, uint32_t qc) {
        if constexpr (WC > 0) {
            // compile-time W: one pass over the chunk per tile, that tile's planes in registers (the filter words are
            // dead meanwhile and reloaded afterwards)
            for (uint32_t t = 0; t < (uint32_t)T; t++) {
                const uint32_t tile = tile0 + t;
                if (tile >= a.tile_end) break;
                const uint4 *src = planes + (size_t)tile * tile_stride + lane;
                uint4 s[PS * (WC > 0 ? WC : 1)];
#pragma unroll
                for (int i = 0; i < PS * WC; i++) s[i] = src[i * 64];
                const uint32_t *rec = rec0;
                for (uint32_t i = 0; i < nqc; i++, rec += RS) {
                    const uint32_t U = ~rec[BS];
                    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                    for (int w = 0; w < WC; w++) {
                        uint32_t extra = 0;
#pragma unroll
                        for (int p = PS; p < PQ; p++) extra |= rec[qslot(PQ, WC, p, w)];
                        uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
#pragma unroll
                        for (int p = 0; p < PS; p++) {
                            const uint4 v = s[p * WC + w];
                            const uint32_t qv = rec[qslot(PQ, WC, p, w)];
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
                    if (__ballot(min(min(d0, d1), min(d2, d3)) <= U) != 0ull) finish(tile, qc + i, U, d0, d1, d2, d3);
                }
            }
            return;
        }
        for (uint32_t t = 0; t < (uint32_t)T; t++) {
            const uint32_t tile = tile0 + t;
            if (tile >= a.tile_end) break;
            const uint4 *src = planes + (size_t)tile * tile_stride + lane;
            for (uint32_t g = 0; g < nqc; g += kWideGroup) {
                uint32_t d[kWideGroup][4];
#pragma unroll
                for (int j = 0; j < kWideGroup; j++) d[j][0] = d[j][1] = d[j][2] = d[j][3] = 0;
#pragma unroll 1
                for (uint32_t w = 0; w < W; w++) {
                    uint4 v[PS];
#pragma unroll
                    for (int p = 0; p < PS; p++) v[p] = src[((size_t)p * W + w) * 64];
#pragma unroll
                    for (int j = 0; j < kWideGroup; j++) {
                        const uint32_t *rec = rec0 + min(g + j, nqc - 1u) * RS;  // past the end: a copy, not reported
                        uint32_t extra = 0;
#pragma unroll
                        for (int p = PS; p < PQ; p++) extra |= rec[qslot(PQ, (int)W, p, (int)w)];
                        uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
#pragma unroll
                        for (int p = 0; p < PS; p++) {
                            const uint32_t qv = rec[qslot(PQ, (int)W, p, (int)w)];
                            m0 = or_xor(m0, v[p].x, qv);
                            m1 = or_xor(m1, v[p].y, qv);
                            m2 = or_xor(m2, v[p].z, qv);
                            m3 = or_xor(m3, v[p].w, qv);
                        }
                        d[j][0] += __builtin_popcount(m0);
                        d[j][1] += __builtin_popcount(m1);
                        d[j][2] += __builtin_popcount(m2);
                        d[j][3] += __builtin_popcount(m3);
                    }
                }
                for (uint32_t j = 0; j < (uint32_t)kWideGroup && g + j < nqc; j++) {
                    const uint32_t U = ~rec0[(g + j) * RS + BS];
                    // dynamic j: pick the group's row with selects (the array stays in registers)
                    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
#pragma unroll
                    for (int jj = 0; jj < kWideGroup; jj++) {
                        const bool mine = (uint32_t)jj == j;
                        e0 = mine ? d[jj][0] : e0;
                        e1 = mine ? d[jj][1] : e1;
                        e2 = mine ? d[jj][2] : e2;
                        e3 = mine ? d[jj][3] : e3;
                    }
                    if (__ballot(min(min(e0, e1), min(e2, e3)) <= U) != 0ull) finish(tile, qc + g + j, U, e0, e1, e2, e3);
                }
            }
        }
    };

    if (q0 < q1) {
        fetch(q0);
        commit(0, q0);
    }
    __syncthreads();
    bool filter_on = a.use_filter != 0;
    bool level1_on = true;
    uint32_t chunk_no = 0;
    for (uint32_t qc = q0; qc < q1; qc += chunk, buf ^= 1, chunk_no++) {
        const uint32_t nqc = min(chunk, q1 - qc);
        const bool more = qc + chunk < q1;
        if (more) fetch(qc + chunk);
        if (active) {
            // pinned to scalars: hipcc cannot see through __ballot that these flags are wave-uniform
            const bool probe =
                __builtin_amdgcn_readfirstlane((int)(a.use_filter && (filter_on || (chunk_no & 15u) == 0))) != 0;
            const uint32_t *rec0 = reinterpret_cast<const uint32_t *>(&stage[buf][0]);
            uint32_t passes = 0, level1_passes = 0;
            const bool l1 = __builtin_amdgcn_readfirstlane((int)level1_on) != 0;
            if (!probe) {
                dense_walk(rec0, nqc, qc);
                load_filter();  // not kept across the walk: its registers hold the group's distances meanwhile
            }
            // ---- zone level (sorted stores): lane i takes query i of the chunk; zq[t] = the queries tile slot t cannot
            // exclude.  Off: every query, every tile.
            unsigned long long zq[T], todo = nqc >= 64u ? ~0ull : (1ull << nqc) - 1ull;
#pragma unroll
            for (int t = 0; t < T; t++) zq[t] = todo;
            if (probe && zone_on) {
                uint4 head = make_uint4(0u, 0u, 0u, 0u);  // lanes past the chunk: ~bound = 0 never passes
                if (lane < nqc) head = heads[buf][lane][0];
                const uint32_t hnu = ONE ? head.y : head.z;
                todo = 0ull;
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const uint32_t zc0 = (uint32_t)__builtin_amdgcn_readlane((int)vz.x, t);
                    const uint32_t zm0 = (uint32_t)__builtin_amdgcn_readlane((int)vz.y, t);
                    uint32_t u = __builtin_popcount((head.x ^ zc0) & zm0) + hnu;
                    if (!ONE) {
                        const uint32_t zc1 = (uint32_t)__builtin_amdgcn_readlane((int)vz.z, t);
                        const uint32_t zm1 = (uint32_t)__builtin_amdgcn_readlane((int)vz.w, t);
                        u += __builtin_popcount((head.y ^ zc1) & zm1);
                    }
                    zq[t] = __ballot((int32_t)u < 0);
                    todo |= zq[t];
                }
            }
            // one query of the chunk against the wave's T tiles; GATED: only the tile slots in `tiles` (the zone level's
            // survivors) — the ungated form keeps the four tiles' level 1 free of branches (a gated loop for every store
            // cost the unsorted ones 20 %: one-word stores, 10M x 20 aa, bound 3: 7.1 -> 8.5 ms)
            auto step = [&](uint32_t i, uint32_t tiles, auto gated_tag) {
                constexpr bool GATED = decltype(gated_tag)::value;
                const uint32_t *rec = rec0 + i * RS;
                uint32_t live = 0;
                {
                    const uint4 head = heads[buf][i][0];  // one LDS read at a constant stride
                    // [f0 f1 bound f2 | f3 ..]; one-word records: [f0 bound g0 ..] (g = plane FP2)
                    const uint32_t qw0 = head.x, nu = ONE ? head.y : head.z;
                    if (l1) {  // level 1: word 0 of the filter plane
                        uint32_t any1 = 0;
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            if (GATED && !((tiles >> t) & 1u)) continue;
                            const uint32_t u0 = __builtin_popcount(f[t][0].x ^ qw0) + nu;
                            const uint32_t u1 = __builtin_popcount(f[t][0].y ^ qw0) + nu;
                            const uint32_t u2 = __builtin_popcount(f[t][0].z ^ qw0) + nu;
                            const uint32_t u3 = __builtin_popcount(f[t][0].w ^ qw0) + nu;
                            any1 = (!GATED && t == 0) ? (or3(u0, u1, u2) | u3) : or3(or3(u0, u1, u2), u3, any1);
                        }
                        const bool go = __ballot((int32_t)any1 < 0) != 0ull;
                        level1_passes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(level1_passes + (go ? 1u : 0u)));
                        if (!go) return;
                    }
                    // level 2: the resident words folded, two subjects per popcount
                    uint32_t qf[NF];
                    qf[0] = qw0;
                    qf[1] = ONE ? head.z : head.y;
                    if (NF >= 3) qf[NF >= 3 ? 2 : 0] = head.w;
                    if (NF >= 4) qf[NF >= 4 ? 3 : 0] = heads[buf][i][HV - 1].x;
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        if (GATED && !((tiles >> t) & 1u)) continue;
                        uint32_t m0 = f[t][0].x ^ qf[0], m1 = f[t][0].y ^ qf[0];
                        uint32_t m2 = f[t][0].z ^ qf[0], m3 = f[t][0].w ^ qf[0];
#pragma unroll
                        for (int w = 1; w < NF; w++) {
                            m0 = or_xor(m0, f[t][w].x, qf[w]);
                            m1 = or_xor(m1, f[t][w].y, qf[w]);
                            m2 = or_xor(m2, f[t][w].z, qf[w]);
                            m3 = or_xor(m3, f[t][w].w, qf[w]);
                        }
                        const uint32_t sign =
                            SMAFA_WIDE_AND_PAIR ? ((__builtin_popcount(m0 & m1) + nu) | (__builtin_popcount(m2 & m3) + nu))
                                           : (or3(__builtin_popcount(m0) + nu, __builtin_popcount(m1) + nu,
                                                  __builtin_popcount(m2) + nu) |
                                              (__builtin_popcount(m3) + nu));
                        if (__ballot((int32_t)sign < 0) != 0ull) live |= 1u << t;
                    }
                    live = (uint32_t)__builtin_amdgcn_readfirstlane((int)live);
                    if (SMAFA_WIDE_AND_PAIR && live != 0) {  // subjects one by one before a tile is streamed (see scan_lazy_kernel)
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            if ((live >> t) & 1u) {
                                uint32_t m0 = f[t][0].x ^ qf[0], m1 = f[t][0].y ^ qf[0];
                                uint32_t m2 = f[t][0].z ^ qf[0], m3 = f[t][0].w ^ qf[0];
#pragma unroll
                                for (int w = 1; w < NF; w++) {
                                    m0 = or_xor(m0, f[t][w].x, qf[w]);
                                    m1 = or_xor(m1, f[t][w].y, qf[w]);
                                    m2 = or_xor(m2, f[t][w].z, qf[w]);
                                    m3 = or_xor(m3, f[t][w].w, qf[w]);
                                }
                                const uint32_t each = or3(__builtin_popcount(m0) + nu, __builtin_popcount(m1) + nu,
                                                          __builtin_popcount(m2) + nu) |
                                                      (__builtin_popcount(m3) + nu);
                                if (__ballot((int32_t)each < 0) == 0ull) live &= ~(1u << t);
                            }
                        }
                        live = (uint32_t)__builtin_amdgcn_readfirstlane((int)live);
                    }
                    if (live == 0) return;
                    passes++;
                }
                while (live) {  // level 3: exact, tile by tile — ONE copy of the comparison code
                    const uint32_t t = (uint32_t)__builtin_ctz(live);
                    live &= live - 1;
                    if (tile0 + t < a.tile_end) wide_compare(tile0 + t, rec, qc + i);
                }
            };
            if (probe && zone_on) {
                while (todo != 0ull) {
                    const uint32_t i = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    uint32_t tiles = 0;  // the tile slots this query still has to look at
#pragma unroll
                    for (int t = 0; t < T; t++) tiles |= (uint32_t)((zq[t] >> i) & 1ull) << t;
                    step(i, tiles, std::true_type{});
                }
            } else if (probe) {
                for (uint32_t i = 0; i < nqc; i++) step(i, (1u << T) - 1u, std::false_type{});
            }
            if (probe) {
                passes = (uint32_t)__builtin_amdgcn_readfirstlane((int)passes);
                filter_on = passes * 4u <= nqc;
                level1_on = l1 ? level1_passes * 2u <= nqc : (chunk_no & 15u) == 15u;
            }
        }
        if (more) commit(buf ^ 1, qc + chunk);
        __syncthreads();
        if (!SEED && a.hits) flush_rows(a, rs, buf);
    }
    if (!SEED) finish_rows(a);
}

// Any (planes, words) shape — the fallback for L > 128 when the bound is too loose for scan_wide_kernel's
// prefilter to prune (or the prefilter is switched off).  A lane keeps word 0
// of the prefilter plane of its 16 subjects (4 wave tiles) in registers and applies the level-1 bound of the
// specialised kernels (exact: popcount over the first 32 columns of one plane <= distance); only for queries
// that survive it are the subjects' words re-read from L2/HBM, plane by plane, for the full comparison.
constexpr int kGenericTiles = 4;

__global__ __launch_bounds__(256) void scan_generic_kernel(const uint4 *__restrict__ planes,
                                                           const uint32_t *__restrict__ qrec, ScanArgs a, uint32_t PS,
                                                           uint32_t PQ, uint32_t W, uint32_t QS) {
    __shared__ RowStage rs;
    if (threadIdx.x == 0) rs.n[0] = rs.n[1] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t wg_tile = blockIdx.x % a.n_wg_tiles;
    const uint32_t qblock = blockIdx.x / a.n_wg_tiles;
    const uint32_t tile0 = a.tile_begin + (wg_tile * kWgWaves + wave) * kGenericTiles;
    const bool active = tile0 < a.tile_end;  // idle waves still take part in the barriers of the row flushes
    const uint32_t FP = (uint32_t)filter_plane((int)PQ);
    const size_t tile_stride = (size_t)PS * W * 64;
    uint4 f[kGenericTiles];
#pragma unroll
    for (int t = 0; t < kGenericTiles; t++)
        f[t] = tile0 + t < a.tile_end ? planes[(size_t)(tile0 + t) * tile_stride + (size_t)FP * W * 64 + lane]
                                      : make_uint4(0, 0, 0, 0);
    const uint32_t q0 = a.q_begin + qblock * a.qb_size;
    const uint32_t q1 = min(q0 + a.qb_size, a.q_end);
    int parity = 0;
    for (uint32_t qc = q0; qc < q1; qc += kChunk, parity ^= 1) {
        const uint32_t qe = min(qc + (uint32_t)kChunk, q1);
        for (uint32_t q = qc; active && q < qe; q++) {
            const uint32_t U = a.thr ? ld_relaxed(a.thr + q) : a.thr0;
            const uint32_t *qr = qrec + (size_t)q * QS;
            bool go = true;
            if (a.use_filter) {
                const uint32_t q0w = qr[0], nu = ~U;  // slot 0 = word 0 of the prefilter plane
                uint32_t any = 0;
#pragma unroll
                for (int t = 0; t < kGenericTiles; t++)
                    any |= or3(__builtin_popcount(f[t].x ^ q0w) + nu, __builtin_popcount(f[t].y ^ q0w) + nu,
                               __builtin_popcount(f[t].z ^ q0w) + nu) |
                           (__builtin_popcount(f[t].w ^ q0w) + nu);
                go = __ballot((int32_t)any < 0) != 0ull;  // wave-uniform
            }
            if (!go) continue;
            for (uint32_t t = 0; t < (uint32_t)kGenericTiles; t++) {
                const uint32_t tile = tile0 + t;
                if (tile >= a.tile_end) break;
                const uint4 *src = planes + (size_t)tile * tile_stride + lane;
                const uint32_t subj0 = tile * kWaveTile + lane * 4u;
                uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                for (uint32_t w = 0; w < W; w++) {
                    uint32_t extra = 0;
                    for (uint32_t p = PS; p < PQ; p++) extra |= qr[qslot((int)PQ, (int)W, (int)p, (int)w)];
                    uint32_t m0 = extra, m1 = extra, m2 = extra, m3 = extra;
                    for (uint32_t p = 0; p < PS; p++) {
                        const uint4 v = src[(p * W + w) * 64];
                        const uint32_t qv = qr[qslot((int)PQ, (int)W, (int)p, (int)w)];
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
                if (d0 <= U && subj0 + 0 < a.n_subjects) emit(a, rs, parity, q, subj0 + 0, d0);
                if (d1 <= U && subj0 + 1 < a.n_subjects) emit(a, rs, parity, q, subj0 + 1, d1);
                if (d2 <= U && subj0 + 2 < a.n_subjects) emit(a, rs, parity, q, subj0 + 2, d2);
                if (d3 <= U && subj0 + 3 < a.n_subjects) emit(a, rs, parity, q, subj0 + 3, d3);
            }
        }
        __syncthreads();
        if (a.hits) flush_rows(a, rs, parity);
    }
    finish_rows(a);
}

// The k >= 2 modes (max_num_hits = k, src/lib.rs:250-256) start from an upper bound of every query's k-th smallest distance:
// the k-th smallest distance within ANY subset of the subjects bounds the k-th smallest over all of them from above.  This
// kernel counts the distances of wave tiles [0, n_tiles) — the seed (4 tiles) or the whole sample of the store (engine.hip
// scan_range) — for EVERY pair, in LDS histograms: one workgroup = 4 waves x kSeedQueries queries (blockIdx.x % n_chunks = the
// chunk of queries) walking its share of the tiles (blockIdx.x / n_chunks = tile group), a wave keeps one tile's planes in
// registers per step and compares it with the chunk's queries in full; ds_add per pair, no global atomic, no row, no bound.
//   cnt == NULL (one tile group):  thr[q] = min(thr0, first d whose cumulative count reaches k) straight from LDS;
//   cnt != NULL:                   the histograms are added to cnt[q][d] (what the counting launches of the scan kernel would
//                                  have counted, exactly) and kth_from_counts_kernel derives the bounds.
// (Before: counting launches of the scan kernel with every bound at thr0 — each counted pair went through global atomics:
// 9 ms per 10 000 queries for the first 256 subjects alone at k = 5, 12 ms for a 1/32 sample at k = 50; profiles/r04_kth.txt.)
// PS_/PQ_/W_ = 0: run-time values (any shape); the common shapes are instantiated with compile-time values.
constexpr int kSeedQueries = 32;
constexpr int kSeedBins = 256;  // sequences of up to 255 columns (longer ones keep the counting launches)
template <int PS_, int PQ_, int W_>
__global__ __launch_bounds__(256) void kth_seed_kernel(const uint4 *__restrict__ planes, const uint32_t *__restrict__ qrec,
                                                       uint32_t QS, uint32_t ps_rt, uint32_t pq_rt, uint32_t w_rt,
                                                       uint32_t n_tiles, uint32_t n_subjects, uint32_t q_begin, uint32_t q_end,
                                                       uint32_t n_chunks, uint32_t n_groups, uint32_t k, uint32_t thr0,
                                                       uint32_t *__restrict__ thr, uint32_t *__restrict__ cnt, uint32_t cnt_stride) {
    const uint32_t PS = PS_ ? (uint32_t)PS_ : ps_rt, PQ = PQ_ ? (uint32_t)PQ_ : pq_rt, W = W_ ? (uint32_t)W_ : w_rt;
    constexpr int kMaxRegs = PS_ && W_ ? PS_ * W_ : 1;  // the tile in registers only for compile-time shapes
    __shared__ uint32_t hist[kSeedQueries][kSeedBins];
    // Running bound per query: the k-th smallest distance among the pairs this workgroup has counted so far (recomputed after
    // every 4-tile step).  Pairs above it cannot be among any k smallest, so they are not counted: after the first steps an
    // ds_add is rare (counting EVERY pair, ~16 lanes per instruction hit the same bin: the 1/16 sample of the metric store
    // took 6.4 ms that way, 4x the comparison itself).  The counts stay complete up to the bound, which never drops below
    // this workgroup's own k-th distance, itself an upper bound of the sample's and of the store's.
    __shared__ uint32_t bound_lds[kSeedQueries];
    // compile-time shapes: the chunk's query records are staged in LDS once (a record read is then a broadcast ds_read, not a
    // scalar load from global memory whose latency the four waves of a workgroup could not hide: 2.2 -> 1.x ms for the 1/32 sample)
    constexpr bool kStaged = PS_ && W_;
    constexpr int kRecWords = kStaged ? qrec_stride(PQ_ ? PQ_ : 1, W_ ? W_ : 1) : 4;
    __shared__ uint32_t qstage[kSeedQueries * kRecWords];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = blockIdx.x % n_chunks, group = blockIdx.x / n_chunks;
    const uint32_t q0 = q_begin + chunk * kSeedQueries;
    const uint32_t nq = min((uint32_t)kSeedQueries, q_end - q0);
    for (uint32_t i = tid; i < (uint32_t)(kSeedQueries * kSeedBins); i += 256u) (&hist[0][0])[i] = 0u;
    if (tid < (uint32_t)kSeedQueries) bound_lds[tid] = thr0;
    if (kStaged)
        for (uint32_t i = tid; i < nq * (uint32_t)kRecWords; i += 256u) qstage[i] = qrec[(size_t)q0 * QS + i];
    __syncthreads();
    // tile groups are strided sets of 4-tile steps: step s of group g covers tiles 4 * (g + s * n_groups) .. + 3
    auto update_bounds = [&]() {  // every thread of the workgroup: first d whose cumulative count reaches k, per query
        __syncthreads();
        if (tid < nq) {
            uint32_t seen = 0;
            const uint32_t old = bound_lds[tid];
            for (uint32_t d = 0; d <= old && d < (uint32_t)kSeedBins; d++) {
                seen += hist[tid][d];
                if (seen >= k) {
                    bound_lds[tid] = d;
                    break;
                }
            }
        }
        __syncthreads();
    };
    for (uint32_t step = group; step * kWgWaves < n_tiles; step += n_groups) {  // (uniform trip count per workgroup)
        const uint32_t tile = min(step * kWgWaves + wave, n_tiles - 1u);
        const bool live = step * kWgWaves + wave < n_tiles;  // a wave past the range still meets the others at the barriers
        const uint4 *t = planes + (size_t)tile * ((size_t)PS * W * 64) + lane;
        uint4 s[kMaxRegs];
        if (PS_ && W_) {
#pragma unroll
            for (int i = 0; i < kMaxRegs; i++) s[i] = t[i * 64];
        }
        const uint32_t pos = tile * kWaveTile + lane * 4u;
        for (uint32_t qi = 0; qi < nq; qi++) {
            const uint32_t *rec_g = qrec + (size_t)(q0 + qi) * QS;
            const uint32_t *rec_l = &qstage[kStaged ? qi * (uint32_t)kRecWords : 0u];
#define rec (kStaged ? rec_l : rec_g)
            uint4 d = make_uint4(0, 0, 0, 0);
            for (uint32_t w = 0; w < W; w++) {  // (compile-time trip counts unroll by themselves; the run-time form cannot)
                uint32_t extra = 0;  // query bits in planes no subject has: a mismatch against every subject
                for (uint32_t p = PS; p < PQ; p++) extra |= rec[qslot((int)PQ, (int)W, (int)p, (int)w)];
                uint4 m = make_uint4(extra, extra, extra, extra);
                for (uint32_t p = 0; p < PS; p++) {
                    const uint4 v = (PS_ && W_) ? s[(PS_ && W_) ? p * W + w : 0] : t[(p * W + w) * 64];
                    const uint32_t qv = rec[qslot((int)PQ, (int)W, (int)p, (int)w)];
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
#undef rec
            const uint32_t bnd = live ? bound_lds[qi] : 0u;
            if (pos + 0 < n_subjects && d.x <= bnd && live) atomicAdd(&hist[qi][d.x], 1u);
            if (pos + 1 < n_subjects && d.y <= bnd && live) atomicAdd(&hist[qi][d.y], 1u);
            if (pos + 2 < n_subjects && d.z <= bnd && live) atomicAdd(&hist[qi][d.z], 1u);
            if (pos + 3 < n_subjects && d.w <= bnd && live) atomicAdd(&hist[qi][d.w], 1u);
        }
        // the bound is brought up to date every 4th step only (a stale, looser bound merely counts a few pairs more): the walk
        // over a histogram is a chain of ~L dependent LDS reads by 32 threads while the other 224 wait at the barrier — as
        // long as a whole step's comparisons
        if ((step / n_groups) % 4u == 3u) update_bounds();
    }
    update_bounds();  // (also the barrier between the last step's ds_adds and the reads below)
    if (cnt) {  // this group's share of the counts: complete up to its own final bound (bins above it hold what was counted
                // while the bound was still looser — partial, and never needed: the sample's k-th distance is at most this bound)
        const uint32_t bins = min(min(thr0 + 1u, cnt_stride), (uint32_t)kSeedBins);
        for (uint32_t i = tid; i < nq * bins; i += 256u) {
            const uint32_t qi = i / bins, d = i % bins;
            const uint32_t v = hist[qi][d];
            if (v && d <= bound_lds[qi]) atomicAdd(cnt + (size_t)(q0 + qi) * cnt_stride + d, v);
        }
    } else if (tid < nq) {
        thr[q0 + tid] = bound_lds[tid];
    }
}

// The literal get_distances seam (src/lib.rs:71-89): every subject's distance to ONE query, written at the subject's
// index (out has n_subjects entries; the store keeps its subjects in sorted positions, `order` maps back).
__global__ __launch_bounds__(256) void distances_kernel(const uint4 *__restrict__ planes, uint32_t n_wave_tiles,
                                                        uint32_t n_subjects, uint32_t PS, uint32_t PQ, uint32_t W,
                                                        const uint32_t *__restrict__ qrec,
                                                        const uint32_t *__restrict__ order, uint32_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * kWgWaves + (threadIdx.x >> 6);
    if (tile >= n_wave_tiles) return;
    const uint4 *t = planes + (size_t)tile * ((size_t)PS * W * 64) + lane;
    uint4 d = make_uint4(0, 0, 0, 0);
    for (uint32_t w = 0; w < W; w++) {
        uint32_t extra = 0;
        for (uint32_t p = PS; p < PQ; p++) extra |= qrec[qslot((int)PQ, (int)W, (int)p, (int)w)];
        uint4 m = make_uint4(extra, extra, extra, extra);
        for (uint32_t p = 0; p < PS; p++) {
            const uint4 v = t[(p * W + w) * 64];
            const uint32_t qv = qrec[qslot((int)PQ, (int)W, (int)p, (int)w)];
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
    const uint32_t pos = tile * kWaveTile + lane * 4u;
    if (pos + 0 < n_subjects) out[order[pos + 0]] = d.x;
    if (pos + 1 < n_subjects) out[order[pos + 1]] = d.y;
    if (pos + 2 < n_subjects) out[order[pos + 2]] = d.z;
    if (pos + 3 < n_subjects) out[order[pos + 3]] = d.w;
}

// ---------------------------------------------------------------------------------------------
// Packing: code bytes -> bit-planes by wave64 ballot.
// Lane c of a wave handles packed column 64h + c of one row: it reads the SOURCE column perm[64h + c] of that row and
// re-codes the byte through tab[source column][code] (the store's layout: best columns first, best-balanced bit in
// plane 0 — choose_layout() in engine.hip; distances are invariant under both).  __ballot(bit p of the code) IS the
// 64-column slice of plane p (low half = word 2h, high half = word 2h+1).  A wave packs 64 consecutive rows, parks
// row i's words in lane i, then stores each (plane, word) as one coalesced 256-byte row.
//   mode 0: subject tile layout   out[((tile*P + p)*W + w)*256 + (row & 255)]
//   mode 1: query record layout   out[row*QS + qslot(p, w)]
// `first` = position of the first packed row (appends start mid-tile).  The row packed at position first + i is row
// src[i] of `codes` (src == NULL: row i) — the sorted order of an append — and order[first + i] = first + src[i] is the
// subject index a scan reports for it.
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint8_t *codes, const uint32_t *src, uint64_t first,
                                                        uint64_t n, uint32_t L, uint32_t W, uint32_t *out, int mode,
                                                        uint32_t QS, const uint32_t *__restrict__ perm,
                                                        const uint8_t *__restrict__ tab, uint32_t *order) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t group = (uint64_t)blockIdx.x * kWgWaves + (threadIdx.x >> 6);
    const uint64_t base = (first / 64 + group) * 64;  // position of lane 0's slot
    const uint64_t end = first + n;
    if (base >= end) return;
    const uint64_t my_row = base + lane;
    const bool mine_valid = my_row >= first && my_row < end;
    // source row of the position this lane parks (and, via readlane, of every position the wave packs)
    uint32_t my_src = 0;
    if (mine_valid) my_src = src ? src[my_row - first] : (uint32_t)(my_row - first);
    if (mine_valid && order) order[my_row] = (uint32_t)first + my_src;
    for (uint32_t h = 0; h * 2 < W; h++) {
        const uint32_t col = 64 * h + lane;
        const uint32_t scol = col < L ? perm[col] : 0u;
        const uint8_t *tcol = tab + (size_t)scol * 32;
        uint32_t lo[P], hi[P];
#pragma unroll
        for (int p = 0; p < P; p++) lo[p] = hi[p] = 0;
#pragma unroll 8
        for (uint32_t i = 0; i < 64; i++) {
            const uint64_t row = base + i;  // wave-uniform
            const uint32_t srow = (uint32_t)__builtin_amdgcn_readlane((int)my_src, (int)i);
            uint32_t code = 0;
            if (row >= first && row < end && col < L) code = tcol[codes[(size_t)srow * L + scol] & 31u];
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
                        out[my_row * QS + qslot(P, (int)W, p, (int)w)] = v;
                    }
                }
            }
        }
    }
}

// Sort key of a subject row: its filter-plane words 0 and 1 (after the layout's column order and re-coding), each
// bit-reversed (column 0 = most significant: the layout puts the most informative columns first) and run through the
// inverse Gray code, so that rows whose keys are close in the sorted order differ in few LEADING filter bits even
// across a power-of-two boundary (a plain binary order loses every bit above the boundary a tile straddles).
__device__ __forceinline__ uint32_t gray_rank(uint32_t x) {
    x = __brev(x);
    x ^= x >> 1;
    x ^= x >> 2;
    x ^= x >> 4;
    x ^= x >> 8;
    x ^= x >> 16;
    return x;
}

__global__ void row_keys_kernel(const uint8_t *__restrict__ codes, uint64_t n, uint32_t L, const uint32_t *__restrict__ perm,
                                const uint8_t *__restrict__ tab, unsigned long long *__restrict__ keys,
                                uint32_t *__restrict__ iota) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *row = codes + (size_t)i * L;
    uint32_t w0 = 0, w1 = 0;
    const uint32_t c0 = min(L, 32u), c1 = min(L, 64u);
    for (uint32_t j = 0; j < c0; j++) {
        const uint32_t sc = perm[j];
        w0 |= (uint32_t)(tab[sc * 32 + (row[sc] & 31u)] & 1u) << j;
    }
    for (uint32_t j = 32; j < c1; j++) {
        const uint32_t sc = perm[j];
        w1 |= (uint32_t)(tab[sc * 32 + (row[sc] & 31u)] & 1u) << (j - 32);
    }
    keys[i] = ((unsigned long long)gray_rank(w0) << 32) | gray_rank(w1);
    iota[i] = (uint32_t)i;
}

// Re-sorting a store in place (engine.hip resort_store): the key of the row at a position, read back from the filter
// plane (the same key row_keys_kernel computes from the code bytes), and the move of every row to its new position.
__global__ void position_keys_kernel(const uint32_t *__restrict__ planes, uint32_t PS, uint32_t W, uint64_t n,
                                     unsigned long long *__restrict__ keys, uint32_t *__restrict__ iota) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *t = planes + (i >> 8) * ((size_t)PS * W * 256) + (i & 255u);  // plane 0 = the filter plane
    const uint32_t w0 = t[0], w1 = W > 1 ? t[256] : 0u;
    keys[i] = ((unsigned long long)gray_rank(w0) << 32) | gray_rank(w1);
    iota[i] = (uint32_t)i;
}

__global__ void permute_rows_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                    const uint32_t *__restrict__ from, const uint32_t *__restrict__ order_in,
                                    uint32_t *__restrict__ order_out, uint64_t n, uint32_t PW) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t old = from[p];
    order_out[p] = order_in[old];
    const uint32_t *s = in + (size_t)(old >> 8) * ((size_t)PW * 256) + (old & 255u);
    uint32_t *d = out + (p >> 8) * ((size_t)PW * 256) + (p & 255u);
    for (uint32_t j = 0; j < PW; j++) d[(size_t)j * 256] = s[(size_t)j * 256];
}

// zone[tile] = {c0, m0, c1, m1}: m_w = the bits of filter word w on which all subjects of the wave tile agree, c_w =
// their common value there.  For a query word q, popcount((q ^ c_w) & m_w) mismatching columns are shared by every
// subject of the tile.  One wave per tile; positions past n_subjects (the padding of the last tile) do not count.
__global__ __launch_bounds__(256) void zone_kernel(const uint4 *__restrict__ planes, uint32_t PS, uint32_t W, uint32_t L,
                                                   uint32_t tile_begin, uint32_t tile_end, uint32_t n_subjects,
                                                   uint4 *__restrict__ zone) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = tile_begin + blockIdx.x * kWgWaves + (threadIdx.x >> 6);
    if (tile >= tile_end) return;
    const uint4 *t = planes + (size_t)tile * ((size_t)PS * W * 64) + lane;  // plane 0 is the filter plane
    const uint32_t pos = tile * kWaveTile + lane * 4u;
    uint32_t land[2] = {0xffffffffu, 0xffffffffu}, lor[2] = {0u, 0u};
    for (uint32_t w = 0; w < 2 && w < W; w++) {
        const uint4 v = t[w * 64];
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (pos + k < n_subjects) {
                land[w] &= x[k];
                lor[w] |= x[k];
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int w = 0; w < 2; w++) {
            land[w] &= (uint32_t)__shfl_xor((int)land[w], off, 64);
            lor[w] |= (uint32_t)__shfl_xor((int)lor[w], off, 64);
        }
    }
    if (lane == 0) {
        uint4 z;
        // bits past the last column are zero in every subject and every query: they never mismatch, and counting them as
        // shared would make the store look better sorted than it is (use_zone works from these words)
        const uint32_t cols0 = L >= 32u ? 0xffffffffu : (1u << L) - 1u;
        const uint32_t cols1 = L >= 64u ? 0xffffffffu : L > 32u ? (1u << (L - 32u)) - 1u : 0u;
        z.y = ~(land[0] ^ lor[0]) & cols0;
        z.x = land[0] & z.y;
        z.w = W > 1 ? (~(land[1] ^ lor[1]) & cols1) : 0u;
        z.z = W > 1 ? (land[1] & z.w) : 0u;
        if (lor[0] == 0u && land[0] == 0xffffffffu) z = make_uint4(0u, 0u, 0u, 0u);  // an empty tile shares nothing
        zone[tile] = z;
    }
}

// Re-layout the subject block from p_old to p_new planes per subject (new planes zero): the N-free
// nucleotide store gains its third plane the first time a subject with an N is appended.
__global__ void replane_kernel(const uint32_t *src, uint32_t *dst, uint64_t n_tiles, uint32_t p_old, uint32_t p_new,
                               uint32_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one u32 of dst
    const uint64_t per_tile = (uint64_t)p_new * W * 256;
    if (i >= n_tiles * per_tile) return;
    const uint64_t tile = i / per_tile, r = i % per_tile;
    const uint32_t p = (uint32_t)(r / (W * 256u)), rest = (uint32_t)(r % (W * 256u));
    dst[i] = p < p_old ? src[tile * (uint64_t)p_old * W * 256 + (uint64_t)p * W * 256 + rest] : 0u;
}

// After the launches of a tightening scan (max_num_hits = k): the scan appended every pair that was within the bound
// of its query AT THAT TIME to the scratch list; only rows within the FINAL bound can be printed, so this pass keeps
// `dist <= thr[query]` and drops the rest — typically 50-100 appended rows per query shrink to one or two before
// anything is sorted or crosses PCIe.  Kept rows land in `out` in arbitrary order through one wave-aggregated atomic
// per wave (*out_count zeroed by the host); *out_count keeps counting past cap.  If the scratch list itself
// overflowed (rows were dropped: the answer is incomplete) nothing is appended and *out_count = cap + 1, the
// "did not fit" signal of the scan API.
__global__ __launch_bounds__(256) void filter_rows_kernel(const smafa_hit *list, const unsigned long long *list_count,
                                                          unsigned long long list_cap, smafa_hit *out,
                                                          unsigned long long cap, unsigned long long *out_count,
                                                          const uint32_t *thr) {
    const unsigned long long total = *list_count;
    if (total > list_cap) {  // every workgroup sees the same counter: nobody appends, one thread reports
        if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = cap + 1;
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long base = (unsigned long long)blockIdx.x * blockDim.x; base < total; base += stride) {
        const unsigned long long i = base + threadIdx.x;  // uniform trip count per workgroup
        smafa_hit h = {0, 0, 0};
        bool keep = false;
        if (i < total) {
            h = list[i];
            keep = h.dist <= thr[h.query];
        }
        const unsigned long long mask = __ballot(keep);
        if (mask == 0ull) continue;
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(out_count, (unsigned long long)__builtin_popcountll(mask));
        first = shfl_u64(first, 0);
        const unsigned long long slot = first + lanes_below(mask);
        if (keep && slot < cap) out[slot] = h;
    }
}

// After a counting pass of the k >= 2 modes: every pair within its query's bound was counted in cnt[q][dist], and the
// bound never dropped below the k-th smallest distance, so the counts up to that distance are complete.  The exact
// k-th distance is the first d whose cumulative count reaches k (fewer than k pairs in range: the bound stays).
__global__ void kth_from_counts_kernel(const uint32_t *cnt, uint32_t cnt_stride, uint32_t k, uint32_t *thr,
                                       uint32_t q_begin, uint32_t nq) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint32_t q = q_begin + i;
    const uint32_t *c = cnt + (size_t)q * cnt_stride;
    const uint32_t bound = thr[q];
    uint32_t seen = 0;
    for (uint32_t d = 0; d <= bound && d < cnt_stride; d++) {
        seen += c[d];
        if (seen >= k) {
            thr[q] = d;
            return;
        }
    }
}

// Empirical HBM read-stream ceiling of the device ("measure an empirical read-stream ceiling on the box with
// a trivial sum kernel over a >= 4 GB buffer"): 16-byte loads per lane, grid-stride, 8 loads in flight per lane.
__global__ __launch_bounds__(256) void hbm_read_probe_kernel(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + 7 * stride < n; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n; i += stride) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;  // keeps the loads alive
}

// The same bytes the way a streaming scan reads them at its best: every workgroup owns one contiguous span, loads carry the
// non-temporal hint (tools/experiments/ubench_stream.hip: 0.87 of 8 TB/s against 0.63-0.76 for the grid-stride form above)
__global__ __launch_bounds__(256) void hbm_read_probe_span_kernel(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t acc = 0;
    size_t i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = ld_nt(p + i + u * 256);
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < hi; i += 256) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void fill_u32_kernel(uint32_t *p, uint32_t v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace smafa
