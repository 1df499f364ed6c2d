// engine.hip — device half of the C ABI (include/smafa_amd.h): HBM-resident subject store, packed
// query sets, scan launches, row collection and ordering.  Host-only entry points (FASTX, DB file,
// selection, drivers) live in host/*.cpp.  No CPU fallback anywhere: without a HIP device every call
// here fails.
#include <hipcub/hipcub.hpp>
#include <cmath>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"
#include "host/packed.h"
#include "kernels.hip.h"
#include "index.hip.h"

namespace smafa {

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return set_error(SMAFA_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// planes per query record: 3 code bits for ACGTN, 5 for the amino-acid codes 0..27
static int query_planes_for(int alphabet) { return alphabet == SMAFA_ALPHABET_AA ? 5 : 3; }

// row counter (u64) at byte 0, ticket counters of finish_rows from byte 256 on
constexpr size_t kCtrBytes = 256 + (size_t)(1 + kTicketGroups) * kTicketStride * sizeof(uint32_t);

// grow-only device buffer (scratch that is reused across calls: hipMalloc/hipFree cost far more than a scan)
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SMAFA_OK;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(bytes, 4096);
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return SMAFA_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace smafa

using namespace smafa;

static std::atomic<uint64_t> g_qset_serial{1};

struct smafa_qset {
    smafa_db *db = nullptr;
    uint64_t nq = 0;
    uint64_t serial = 0;  // unique per fill of the set: a recreated set at a recycled address is a different set
    DevBuf qrec, thr, cnt;  // packed records, per-query bounds, per-query distance histograms (k >= 2 only)
};

struct smafa_db {
    int device = 0;
    int alphabet = 0;
    uint32_t L = 0, W = 0, QS = 0;
    uint32_t P = 0;   // planes stored per SUBJECT: 5 (aa), 3 (nt), or 2 while no nucleotide subject holds an N
    uint32_t PQ = 0;  // planes per QUERY record: 5 (aa) or 3 (nt)
    uint64_t n = 0;          // subjects stored
    uint64_t cap_tiles = 0;  // allocated wave tiles
    uint32_t *d_planes = nullptr;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    uint32_t last_launches = 0;
    // totals over the scans of the most recent host-buffer call (smafa_scan_hits: the near-hit ladder and the tightening
    // path are several scans): smafa_last_call_stats
    float call_ms = 0.f;
    uint32_t call_launches = 0, call_scans = 0;
    // Where the scan launches of this handle really went: the HIP device current on the launching thread at each launch
    // (smafa_launch_device).  A member of a multi-device group whose worker thread forgot hipSetDevice would show here.
    int launch_device = -1;
    uint64_t launches_off_device = 0;
    double life_ms = 0.0;        // ... and over the handle's life (the cluster driver's debug line)
    uint64_t life_launches = 0;
    // smafa_scan_each: the K one-query launches captured once as a HIP graph and replayed
    hipGraphExec_t each_graph = nullptr;
    // (the captured kernel nodes bake in the set's device record buffer: the key names the set by its serial and that pointer,
    // not by its host address alone — a destroyed set's address is handed out again by the next `new`)
    struct EachKey { const void *qs, *qrec, *hits, *counts; uint64_t qs_serial, cap, nq, generation; uint32_t max_div, qb; int zone; bool filter; } each_key{};
    uint32_t qb_override = 0;
    bool use_filter = true;  // exact lower-bound prefilter in the scan kernel (SMAFA_FILTER=0 disables)
    uint32_t tiles_override = 0;  // SMAFA_TILES
    bool lazy = true;             // filter-plane-resident kernel where it applies (SMAFA_LAZY=0 disables)
    bool wide_one = true;         // one-word stores (L <= 32) through scan_wide_kernel's two-plane level 2 (SMAFA_WIDE_ONE=0: lazy kernel)
    uint32_t wide_from = 5;       // words per plane from which scan_wide_kernel replaces the per-length kernels (SMAFA_WIDE_FROM)
    // what the last launch used (smafa_last_scan_plan)
    uint32_t plan_lazy = 0, plan_tiles = 1, plan_qblocks = 1;
    char plan_kernel[96] = "";  // the instantiation of the last launch, as rocprofv3 names it (smafa_last_scan_kernel)
    int n_cu = 256;
    // scratch of the host-buffer API, kept across calls
    DevBuf upload;            // staging for code rows on their way to the pack kernel
    DevBuf hits, count;       // rows and their counter
    DevBuf scratch;           // row list of the tightening modes (everything appended while bounds were still running)
    DevBuf ctrs;              // [0]: the scan kernels' row counter (u64), [256]: finished-workgroup tickets (u32); both zero between scans
    DevBuf keys_a, keys_b, sort_tmp;
    DevBuf idx_a, idx_b;      // sort payload of an append: source row per sorted position
    // ---- layout of the packed store (fixed when the first rows arrive)
    bool layout_set = false;
    std::vector<uint32_t> perm;  // packed column j holds source column perm[j]
    std::vector<uint8_t> tab;    // [source column][code] -> stored code
    DevBuf d_perm, d_tab;
    uint32_t *d_order = nullptr;  // position -> subject index (cap_tiles * 256 entries)
    uint4 *d_zone = nullptr;      // per wave tile: shared filter bits (cap_tiles entries)
    struct Run { uint64_t rows; bool sorted; };
    std::vector<Run> runs;        // the appends the store consists of (each sorted within itself or not); kept in the packed file
    // What the zone level can prune with, MEASURED: shared filter bits per wave tile (both words), read back after every
    // append; hist[b] = tiles sharing b bits.  Sorting gives ~log2(rows / 256) on unrelated sequences, related ones share
    // more, many small appends share few — use_zone() works from this, not from assumptions about the data.
    std::vector<uint8_t> tile_bits;
    uint64_t zone_hist[65] = {0};
    uint64_t rows_since_sort = 0;  // rows appended since the whole store was last in one sorted run
    uint32_t resorts = 0;          // full re-sorts so far (resort_store)
    bool resort = true;            // SMAFA_RESORT=0: never
    double prune_p = 2e-3;         // prefilter_prunes: largest level-1 pass probability per subject (SMAFA_PRUNE_P)
    uint64_t resort_min = 32768;   // stores below this many rows are left alone (SMAFA_RESORT_MIN)
    double zone_loose = 0.3;      // pass share below which the zone kernel also takes bounds level 1 cannot prune at (SMAFA_ZONE_LOOSE)
    int zone = 1;                 // zone level of the filter-plane-resident kernel: 1 = where it prunes (use_zone), 0 = never
                                  // (SMAFA_ZONE=0), 2 = whenever that kernel runs (SMAFA_ZONE=2, tests)
    bool sort_rows = true;        // sort big appends by their filter words (SMAFA_SORT=0: keep the append order)
    smafa_qset scratch_q;     // query set of smafa_scan_hits / smafa_distances
    smafa_qset scratch_q2;    // the compacted batch of queries the near-hit probe did not finish
    smafa_qset scratch_q3;    // the sample of open queries the later steps of the ladder are planned from
    bool two_phase = true;    // near-hit probe before the tightening path (SMAFA_TWO_PHASE=0 disables)
    bool fold3 = true;        // scan_kernel's all-planes-but-the-last bound for launches whose bound starts above 32 (SMAFA_FOLD3=0)
    bool stream_nt = true;    // one-query-block launches of scan_lazy_kernel load their filter words non-temporally (SMAFA_STREAM_NT=0)
    uint32_t count_first_k = 3;  // smallest k whose loose-bound scans count first and append second (SMAFA_COUNT_FIRST_K)
    bool ladder_probe = true;    // the ladder's first step is asked of a 256-query sample before the whole batch pays for it (SMAFA_LADDER_PROBE=0)
    bool zone_direct = true;     // fixed-bound zone launches without LDS staging and barriers (SMAFA_ZONE_DIRECT=0: the staged form)
    bool lazy_fold = true;       // the filter-plane-resident kernel also at the bounds only its level 2 rejects at (SMAFA_LAZY_FOLD=0)
    bool kth_hist_seed = true;   // k >= 2: the seed bound from an LDS histogram over the first tiles (SMAFA_KTH_HIST_SEED=0: a counting launch)
    uint32_t kth_sample_min_tiles = 4096;  // stores below this many wave tiles (1M subjects) count everything first (SMAFA_KTH_SAMPLE_MIN_TILES)
    uint32_t kth_sample_div = 32;  // ... counting only the first 1/32 of the tiles, the rest counted and appended in one pass (SMAFA_KTH_SAMPLE=0: count everything first)
    // rows of a smafa_scan_hits call that ended in SMAFA_ERR_CAPACITY, kept for the caller's "grow and retry":
    // the retry with the same arguments against the same store is answered without scanning again
    std::vector<smafa_hit> retry_rows;
    std::vector<uint8_t> retry_codes;  // the query bytes of that call (compared exactly; the key only screens)
    uint64_t retry_key = 0, retry_nq = 0, generation = 0;  // generation: bumped whenever the subjects change
    uint32_t retry_div = 0, retry_k = 0;
    bool retry_valid = false;
    // ---- block index of the resident store (index.hip.h): built on request (smafa_db_build_index) or on demand (mode 2), tied to
    // the store's state — any append, re-sort or re-plane leaves it stale and the scan kernels run until it is built again
    struct BlockIndex {
        bool valid = false;
        uint64_t generation = 0, n = 0;
        uint32_t resorts = 0, P = 0;
        uint32_t B = 0, dir_bits = 0;
        uint16_t col_begin[kIndexMaxBlocks + 1] = {0};
        uint64_t max_run[kIndexMaxBlocks] = {0};  // longest run of equal keys of block b
        double mean_run[kIndexMaxBlocks] = {0};   // sum(run^2) / n: candidates a query drawn like the store's rows meets there
        double build_ms = 0.0;
        DevBuf kp, dir, stats, rows;
    } index;
    int index_mode = 1;             // 0: never probed; 1: probed where a built index pays; 2: also built by the first scan that could
                                    // use one; 3: ... built once such scans have cost what the build would (SMAFA_INDEX)
    double index_debt_ms = 0.0;     // mode 3: estimated kernel time of the eligible scans since the store last changed
    uint64_t index_debt_generation = 0;
    uint64_t index_failed_generation = 0;  // generation + 1 of the store whose automatic build failed (0: none)
    uint64_t index_max_run = 4096;  // a block whose longest run exceeds this is never probed (SMAFA_INDEX_MAX_RUN)
    double index_cand_per_subject = -1.0;  // candidates per query the probes may expect, per stored subject (SMAFA_INDEX_CAND;
                                           // < 0: by the bound's class — index_cand_limit)
    uint64_t index_min_rows = 65536;  // mode 2 builds an index for stores of at least this many subjects (SMAFA_INDEX_MIN_ROWS)
    uint32_t index_probes = 0;      // launches answered by the index over the handle's life (smafa_index_info)
    size_t tile_words() const { return (size_t)P * W * kWaveTile; }
    uint64_t hits_cap() const { return hits.cap / sizeof(smafa_hit); }
};

namespace smafa {

static int use_device(const smafa_db *db) {
    HIP_TRY(hipSetDevice(db->device));
    return SMAFA_OK;
}

// ---- layout of a store: which source column sits in which packed column, and how each column's codes are re-coded.
// Both leave every distance unchanged (a distance counts columns whose codes differ; neither the order of the columns
// nor a per-column injective renaming of the codes changes that), so they are free parameters of the HBM layout, chosen
// for the prefilter: plane 0 is the plane its lower bounds look at, packed columns 0..31 its first level.
//   * per column, the letters are split into two sets of nearly equal total frequency (greedy, by descending count);
//     the first set gets even codes, the second odd ones: bit 0 of the stored code then flips for a mismatch as often
//     as this column's letter distribution allows.  Nucleotides: A C G T are permuted among 0..3 and N stays 4 (the
//     2-plane form of an N-free store holds bits 0 and 1); of the three A/C/G/T pairings, {A,C}|{G,T} is preferred
//     while it is within 5 % of the best — it is the one that sees transitions, the commonest real substitutions.
//   * columns are ordered by that flip probability, 2p(1-p), best first: conserved columns end up in the last words.
// Decided once per handle, from a sample of the first rows it receives (compute_layout, host/layout.cpp).
static int choose_layout(smafa_db *db, const uint8_t *codes, uint64_t n) {
    compute_layout(db->alphabet, db->L, codes, n, db->perm, db->tab);  // host/layout.cpp
    int rc = db->d_perm.ensure(db->perm.size() * sizeof(uint32_t));
    if (!rc) rc = db->d_tab.ensure(db->tab.size());
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(db->d_perm.p, db->perm.data(), db->perm.size() * sizeof(uint32_t), hipMemcpyHostToDevice, db->stream));
    HIP_TRY(hipMemcpyAsync(db->d_tab.p, db->tab.data(), db->tab.size(), hipMemcpyHostToDevice, db->stream));
    HIP_TRY(hipStreamSynchronize(db->stream));  // the vectors may be reallocated later; the copy is done now
    db->layout_set = true;
    return SMAFA_OK;
}

template <int P>
static void launch_pack(smafa_db *db, const uint8_t *d_codes, const uint32_t *d_src, uint64_t first, uint64_t n,
                        uint32_t *d_out, int mode, uint32_t *d_order) {
    const uint64_t groups = (first + n + 63) / 64 - first / 64;
    const uint32_t blocks = (uint32_t)((groups + kWgWaves - 1) / kWgWaves);
    hipLaunchKernelGGL(pack_rows_kernel<P>, dim3(blocks), dim3(256), 0, db->stream, d_codes, d_src, first, n, db->L, db->W,
                       d_out, mode, db->QS, db->d_perm.as<uint32_t>(), db->d_tab.as<uint8_t>(), d_order);
}

// fold the zone words of tiles [t0, t0 + count) into the store's shared-bit statistics (a re-computed tile replaces its
// earlier entry)
static void note_zone_words(smafa_db *db, uint32_t t0, const uint4 *z, size_t count) {
    if (db->tile_bits.size() < (size_t)t0 + count) db->tile_bits.resize((size_t)t0 + count, 255);
    for (size_t i = 0; i < count; i++) {
        uint8_t &slot = db->tile_bits[t0 + i];
        if (slot != 255) db->zone_hist[slot]--;
        slot = (uint8_t)(__builtin_popcount(z[i].y) + __builtin_popcount(z[i].w));
        db->zone_hist[slot]++;
    }
}

constexpr uint64_t kSortMin = 4096;  // appends of fewer rows keep their order (their tiles share few bits anyway)

// Upload code rows and pack them with the ballot kernel.  mode 0: subjects, at positions first.. of the store — big
// appends are first sorted by their filter words (row_keys_kernel + a device radix sort), so that the subjects of a
// wave tile share their leading filter bits (the zone level of the scan), and the zone words of the touched tiles are
// recomputed; mode 1: query records, in the caller's order.
static int pack_rows(smafa_db *db, const uint8_t *codes, uint64_t first, uint64_t n, uint32_t *d_out, int mode) {
    if (n == 0) return SMAFA_OK;
    if (n > 0xfffffff0ull) return set_error(SMAFA_ERR_INVALID, "too many rows in one call");
    if (!db->layout_set) {
        int lrc = choose_layout(db, codes, mode == 0 ? n : 0);  // a query batch says nothing about the subjects
        if (lrc) return lrc;
    }
    const size_t bytes = (size_t)n * db->L;
    int rc = db->upload.ensure(bytes);
    if (rc) return rc;
    uint8_t *d_codes = db->upload.as<uint8_t>();
    HIP_TRY(hipMemcpyAsync(d_codes, codes, bytes, hipMemcpyHostToDevice, db->stream));
    const uint32_t *d_src = nullptr;
    // (the device radix sort counts its items in an int: appends of 2^31 rows and more keep their order)
    const bool sorted = mode == 0 && db->sort_rows && n >= kSortMin && n < (1ull << 31);
    if (sorted) {
        rc = db->keys_a.ensure(n * sizeof(unsigned long long));
        if (!rc) rc = db->keys_b.ensure(n * sizeof(unsigned long long));
        if (!rc) rc = db->idx_a.ensure(n * sizeof(uint32_t));
        if (!rc) rc = db->idx_b.ensure(n * sizeof(uint32_t));
        if (rc) return rc;
        unsigned long long *ka = db->keys_a.as<unsigned long long>(), *kb = db->keys_b.as<unsigned long long>();
        uint32_t *ia = db->idx_a.as<uint32_t>(), *ib = db->idx_b.as<uint32_t>();
        hipLaunchKernelGGL(row_keys_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, db->stream, d_codes, n, db->L,
                           db->d_perm.as<uint32_t>(), db->d_tab.as<uint8_t>(), ka, ia);
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ka, kb, ia, ib, (int)n, 0, 64, db->stream));
        rc = db->sort_tmp.ensure(tmp_bytes);
        if (rc) return rc;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(db->sort_tmp.p, tmp_bytes, ka, kb, ia, ib, (int)n, 0, 64, db->stream));
        d_src = ib;
    }
    const uint32_t planes = mode == 0 ? db->P : db->PQ;  // subjects may be stored with fewer planes than queries
    uint32_t *d_order = mode == 0 ? db->d_order : nullptr;
    if (planes == 5) launch_pack<5>(db, d_codes, d_src, first, n, d_out, mode, d_order);
    else if (planes == 3) launch_pack<3>(db, d_codes, d_src, first, n, d_out, mode, d_order);
    else launch_pack<2>(db, d_codes, d_src, first, n, d_out, mode, d_order);
    HIP_TRY(hipGetLastError());
    if (mode == 0) {
        const uint32_t t0 = (uint32_t)(first / kWaveTile), t1 = (uint32_t)((first + n + kWaveTile - 1) / kWaveTile);
        hipLaunchKernelGGL(zone_kernel, dim3((t1 - t0 + kWgWaves - 1) / kWgWaves), dim3(256), 0, db->stream,
                           reinterpret_cast<const uint4 *>(db->d_planes), db->P, db->W, db->L, t0, t1, (uint32_t)(first + n), db->d_zone);
        HIP_TRY(hipGetLastError());
        db->runs.push_back({n, sorted});
        std::vector<uint4> z(t1 - t0);
        HIP_TRY(hipMemcpyAsync(z.data(), db->d_zone + t0, z.size() * sizeof(uint4), hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        note_zone_words(db, t0, z.data(), z.size());
    }
    // the caller's host buffer is borrowed for the call only, and `upload` is reused by the next call
    HIP_TRY(hipStreamSynchronize(db->stream));
    return SMAFA_OK;
}

// A store that grew by many appends — a DB loaded in pieces, cluster's centroid set — is a patchwork of runs that were
// sorted one by one (or not at all, below kSortMin rows): its tiles share far fewer filter bits than a store of that size
// can.  Before a scan, once the rows appended since the last full sort make up a quarter of the store, the whole store is
// sorted again ON THE DEVICE: keys read back from the filter plane, one radix sort, every row moved to its new position
// (planes and order[]), zone words recomputed.  Each re-sort is paid for by the growth since the last one (geometric:
// <= 4 row moves per row over the store's life); subject indices (order[]) do not change, only positions do.
// No room for a second copy of the planes: the store stays as it is.
static int resort_store(smafa_db *db) {
    const uint64_t n = db->n;
    const double t_begin = now_seconds();
    db->rows_since_sort = 0;  // whatever happens below: not again before the store has grown
    int rc = db->keys_a.ensure(n * sizeof(unsigned long long));
    if (!rc) rc = db->keys_b.ensure(n * sizeof(unsigned long long));
    if (!rc) rc = db->idx_a.ensure(n * sizeof(uint32_t));
    if (!rc) rc = db->idx_b.ensure(n * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t *d_new = nullptr, *d_order = nullptr;
    uint4 *d_zone = nullptr;  // the new zone words go to their own buffer: the live ones stay valid until the swap below
    const size_t bytes = db->cap_tiles * db->tile_words() * sizeof(uint32_t);
    const size_t order_bytes = db->cap_tiles * kWaveTile * sizeof(uint32_t);
    const size_t zone_bytes = db->cap_tiles * sizeof(uint4);
    if (hipMalloc(&d_new, bytes) != hipSuccess || hipMalloc(&d_order, order_bytes) != hipSuccess ||
        hipMalloc(&d_zone, zone_bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (d_new) (void)hipFree(d_new);
        if (d_order) (void)hipFree(d_order);
        return SMAFA_OK;
    }
    auto fail = [&](int code) {
        (void)hipFree(d_new);
        (void)hipFree(d_order);
        (void)hipFree(d_zone);
        return code;
    };
    unsigned long long *ka = db->keys_a.as<unsigned long long>(), *kb = db->keys_b.as<unsigned long long>();
    uint32_t *ia = db->idx_a.as<uint32_t>(), *ib = db->idx_b.as<uint32_t>();
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipError_t e = hipMemsetAsync(d_new, 0, bytes, db->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_order, 0, order_bytes, db->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_zone, 0, zone_bytes, db->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(position_keys_kernel, dim3(blocks), dim3(256), 0, db->stream, db->d_planes, db->P, db->W, n, ka, ia);
        e = hipGetLastError();
    }
    size_t tmp_bytes = 0;
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ka, kb, ia, ib, (int)n, 0, 64, db->stream);
    if (e == hipSuccess && (rc = db->sort_tmp.ensure(tmp_bytes)) != 0) return fail(rc);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(db->sort_tmp.p, tmp_bytes, ka, kb, ia, ib, (int)n, 0, 64, db->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(permute_rows_kernel, dim3(blocks), dim3(256), 0, db->stream, db->d_planes, d_new, ib, db->d_order,
                           d_order, n, db->P * db->W);
        e = hipGetLastError();
    }
    const uint32_t n_tiles = (uint32_t)((n + kWaveTile - 1) / kWaveTile);
    std::vector<uint4> z(n_tiles);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(zone_kernel, dim3((n_tiles + kWgWaves - 1) / kWgWaves), dim3(256), 0, db->stream,
                           reinterpret_cast<const uint4 *>(d_new), db->P, db->W, db->L, 0u, n_tiles, (uint32_t)n, d_zone);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(z.data(), d_zone, z.size() * sizeof(uint4), hipMemcpyDeviceToHost, db->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(db->stream);
    if (e != hipSuccess) return fail(set_error(SMAFA_ERR_DEVICE, "re-sorting the store failed: %s", hipGetErrorString(e)));
    // planes, order and zone words change hands together: a failure above leaves the handle exactly as it was
    (void)hipFree(db->d_planes);
    (void)hipFree(db->d_order);
    (void)hipFree(db->d_zone);
    db->d_planes = d_new;
    db->d_order = d_order;
    db->d_zone = d_zone;
    db->tile_bits.clear();
    for (uint64_t &h : db->zone_hist) h = 0;
    note_zone_words(db, 0, z.data(), z.size());
    db->runs.assign(1, {n, true});
    db->resorts++;
    log_line(2, "store of %llu rows sorted again on the device in %.2f ms (re-sort %u)", (unsigned long long)n,
             (now_seconds() - t_begin) * 1e3, db->resorts);
    if (n > (1u << 20))
        for (DevBuf *b : {&db->keys_a, &db->keys_b, &db->idx_a, &db->idx_b, &db->sort_tmp}) b->release();
    return SMAFA_OK;
}

static int maybe_resort(smafa_db *db) {
    if (!db->sort_rows || !db->resort || db->runs.size() < 2 || db->n < db->resort_min || db->n >= (1ull << 31)) return SMAFA_OK;
    if (db->rows_since_sort * 4 < db->n) return SMAFA_OK;
    return resort_store(db);
}

static int validate_codes(const smafa_db *db, const uint8_t *codes, uint64_t n, uint8_t *max_code = nullptr) {
    const uint8_t lim = db->alphabet == SMAFA_ALPHABET_AA ? 28 : 5;
    const size_t total = (size_t)n * db->L;
    uint8_t worst = 0;
    for (size_t i = 0; i < total; i++) worst = codes[i] > worst ? codes[i] : worst;
    if (worst >= lim) return set_error(SMAFA_ERR_INVALID, "code byte %u outside the alphabet (max %u)", worst, lim - 1);
    if (max_code) *max_code = worst;
    return SMAFA_OK;
}

// 2-plane (N-free) nucleotide store -> 3 planes, in place of the old block
static int upgrade_planes(smafa_db *db, uint32_t p_new) {
    if (db->P >= p_new) return SMAFA_OK;
    if (db->cap_tiles) {
        uint32_t *d_new = nullptr;
        const size_t words = db->cap_tiles * (size_t)p_new * db->W * kWaveTile;
        HIP_TRY(hipMalloc(&d_new, words * sizeof(uint32_t)));
        hipLaunchKernelGGL(replane_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, db->stream, db->d_planes,
                           d_new, db->cap_tiles, db->P, p_new, db->W);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(db->stream));
        HIP_TRY(hipFree(db->d_planes));
        db->d_planes = d_new;
    }
    db->P = p_new;
    return SMAFA_OK;
}

static int reserve_tiles(smafa_db *db, uint64_t need_tiles) {
    if (need_tiles <= db->cap_tiles) return SMAFA_OK;
    uint64_t ncap = db->cap_tiles ? db->cap_tiles * 2 : 64;
    if (ncap < need_tiles) ncap = need_tiles;
    ncap = (ncap + kWgWaves - 1) / kWgWaves * kWgWaves;
    uint32_t *d_new = nullptr, *d_order = nullptr;
    uint4 *d_zone = nullptr;
    const size_t bytes = ncap * db->tile_words() * sizeof(uint32_t);
    HIP_TRY(hipMalloc(&d_new, bytes));
    HIP_TRY(hipMalloc(&d_order, ncap * kWaveTile * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d_zone, ncap * sizeof(uint4)));
    // zero-fill: padding subjects of the last tile read as all-zero planes and are masked by position; a zero zone
    // entry shares no bits (prunes nothing)
    HIP_TRY(hipMemsetAsync(d_new, 0, bytes, db->stream));
    HIP_TRY(hipMemsetAsync(d_order, 0, ncap * kWaveTile * sizeof(uint32_t), db->stream));
    HIP_TRY(hipMemsetAsync(d_zone, 0, ncap * sizeof(uint4), db->stream));
    if (db->d_planes) {
        HIP_TRY(hipMemcpyAsync(d_new, db->d_planes, db->cap_tiles * db->tile_words() * sizeof(uint32_t),
                               hipMemcpyDeviceToDevice, db->stream));
        HIP_TRY(hipMemcpyAsync(d_order, db->d_order, db->cap_tiles * kWaveTile * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                               db->stream));
        HIP_TRY(hipMemcpyAsync(d_zone, db->d_zone, db->cap_tiles * sizeof(uint4), hipMemcpyDeviceToDevice, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        HIP_TRY(hipFree(db->d_planes));
        HIP_TRY(hipFree(db->d_order));
        HIP_TRY(hipFree(db->d_zone));
    }
    db->d_planes = d_new;
    db->d_order = d_order;
    db->d_zone = d_zone;
    db->cap_tiles = ncap;
    return SMAFA_OK;
}

// fill a query set (its buffers grow as needed) from host code rows
static int qset_fill(smafa_qset *qs, smafa_db *db, const uint8_t *query_codes, uint64_t n_queries) {
    qs->db = db;
    qs->nq = n_queries;
    qs->serial = g_qset_serial.fetch_add(1);
    // whole 64-query chunks plus one: a chunk staged by scan_zone_kernel's LDS-DMA is always 64 records from wherever its
    // query block starts (SMAFA_ZONE_FULL_DMA), so up to 63 records past the last query are read (zeros, never used)
    const uint64_t padded = std::max<uint64_t>((n_queries + 63) / 64 * 64, 64) + 64;
    int rc = qs->qrec.ensure(padded * db->QS * sizeof(uint32_t));
    if (!rc) rc = qs->thr.ensure(padded * sizeof(uint32_t));
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(qs->qrec.p, 0, padded * db->QS * sizeof(uint32_t), db->stream));
    return pack_rows(db, query_codes, 0, n_queries, qs->qrec.as<uint32_t>(), 1);
}

// remember which instantiation ran, spelled the way rocprofv3 lists it
static void note_kernel(const smafa_db *db, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void note_kernel(const smafa_db *db, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(const_cast<smafa_db *>(db)->plan_kernel, sizeof db->plan_kernel, fmt, ap);
    va_end(ap);
}

template <int PS, int PQ, int W, int T>
static void launch_lazy_t(const smafa_db *db, const uint32_t *d_qrec, const ScanArgs &a, uint32_t grid) {
    const bool seed = a.hits == nullptr && a.k_tight == 1;
    // two words per plane, bound 13..17: level 2 sums the filter plane's per-word popcounts (the OR-fold rejects nothing there)
    const bool sumfold = W == 2 && !seed && a.thr0 > 12u && a.thr0 <= 17u;  // (fold_rejects sends bounds up to 14 here)
    note_kernel(db, "smafa::scan_lazy_kernel<%d, %d, %d, %d, %s, %s>", PS, PQ, W, T, seed ? "true" : "false", sumfold ? "true" : "false");
    const uint4 *planes = reinterpret_cast<const uint4 *>(db->d_planes);
    if (seed)
        hipLaunchKernelGGL((scan_lazy_kernel<PS, PQ, W, T, true>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
    else if (W == 2 && sumfold)
        hipLaunchKernelGGL((scan_lazy_kernel<PS, PQ, W, T, false, W == 2>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
    else
        hipLaunchKernelGGL((scan_lazy_kernel<PS, PQ, W, T, false>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
}

// sorted store: the zone level in front (scan_zone_kernel; up to 64 queries per launch: scan_zone_few_kernel)
template <int PS, int PQ, int W>
static void launch_zone_t(const smafa_db *db, const uint32_t *d_qrec, const ScanArgs &a, uint32_t grid) {
    const uint4 *planes = reinterpret_cast<const uint4 *>(db->d_planes);
    if (a.q_end - a.q_begin <= 64u) {
        note_kernel(db, "smafa::scan_zone_few_kernel<%d, %d, %d>", PS, PQ, W);
        hipLaunchKernelGGL((scan_zone_few_kernel<PS, PQ, W>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
        return;
    }
    const bool fixed = a.thr == nullptr;  // one bound for every query: LDS-DMA staging, scalar bound
    const bool direct = fixed && db->zone_direct && a.hits != nullptr;  // ... or no staging at all (SMAFA_ZONE_DIRECT)
    note_kernel(db, "smafa::scan_zone_kernel<%d, %d, %d, %s, %s>", PS, PQ, W, fixed ? "true" : "false", direct ? "true" : "false");
    if (direct)
        hipLaunchKernelGGL((scan_zone_kernel<PS, PQ, W, true, true>), dim3(grid), dim3(kZoneWgWaves * 64), 0, db->stream, planes, d_qrec, a);
    else if (fixed)
        hipLaunchKernelGGL((scan_zone_kernel<PS, PQ, W, true>), dim3(grid), dim3(kZoneWgWaves * 64), 0, db->stream, planes, d_qrec, a);
    else
        hipLaunchKernelGGL((scan_zone_kernel<PS, PQ, W, false>), dim3(grid), dim3(kZoneWgWaves * 64), 0, db->stream, planes, d_qrec, a);
}

template <int PS, int PQ, int W, int T>
static void launch_scan_t(const smafa_db *db, const uint32_t *d_qrec, const ScanArgs &a, uint32_t grid) {
    // the seed pass of the running-minimum mode (no append) has its own instantiation
    const bool seed = a.hits == nullptr && a.k_tight == 1;
    // Two words per plane: level 2 by the bound of the launch.  Up to 12 the OR-fold of the filter plane's words (one
    // popcount per subject); 13..17 the filter plane's per-word popcounts summed (FOLD 1: the OR-fold rejects nothing
    // there); 18..32, stores of 3 planes and more, the same over two planes (FOLD 2: flat 14.7 ms from 18 to 28 where the full
    // comparison costs 31, 10 000 queries x 10M aa); beyond that nothing rejects.
    int fold = 0;
    if (SMAFA_SUM_FOLD && W == 2 && !seed && a.use_filter) {
        if (a.thr0 > 12u && a.thr0 <= 17u) fold = 1;
        else if (PS >= 3 && a.thr0 >= 18u && a.thr0 <= 32u) fold = 2;
        else if (PS >= 4 && a.thr0 > 32u && db->fold3) fold = 3;  // all planes but the last (the k-th modes without a bound)
        else if (PS == 3 && a.thr0 > 32u && db->fold3) fold = 2;  // three planes: "all but the last" IS the two-plane form
    }
    note_kernel(db, "smafa::scan_kernel<%d, %d, %d, %d, %s, %d>", PS, PQ, W, T, seed ? "true" : "false", fold);
    const uint4 *planes = reinterpret_cast<const uint4 *>(db->d_planes);
    if (seed)
        hipLaunchKernelGGL((scan_kernel<PS, PQ, W, T, true, 0>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
    else if (W == 2 && fold == 1)
        hipLaunchKernelGGL((scan_kernel<PS, PQ, W, T, false, (W == 2 ? 1 : 0)>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
    else if (W == 2 && PS >= 3 && fold == 2)
        hipLaunchKernelGGL((scan_kernel<PS, PQ, W, T, false, (W == 2 && PS >= 3 ? 2 : 0)>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
    else if (W == 2 && PS >= 4 && fold == 3)
        hipLaunchKernelGGL((scan_kernel<PS, PQ, W, T, false, (W == 2 && PS >= 4 ? 3 : 0)>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
    else
        hipLaunchKernelGGL((scan_kernel<PS, PQ, W, T, false, 0>), dim3(grid), dim3(256), 0, db->stream, planes, d_qrec, a);
}

// wave tiles per wave (4*T subjects per lane).  With the cheap first-level bound the per-query work that does
// not depend on the subject count (LDS read, OR tree, compare, branches, loop bookkeeping) is what T amortises:
// measured 2 beats 1 for every store with W <= 2 even where it costs occupancy (profiles/r01_variant_tiles*.txt).
// SMAFA_TILES=1|2|4 overrides (4: 2-plane store only).
// The filter-plane-resident kernel wins where the prefilter prunes (sparse hits: +18 % aa, 5x less HBM traffic)
// and loses 10-100 % where it cannot (profiles/r01_lazy_vs_resident.txt, r01_length_probe.txt).  Chosen per launch
// from the initial bound, so best-hit scans without --max-divergence (bound = L) and short sequences with a loose
// bound keep the all-planes kernel.
// Level 1 looks at cols = min(32, L) columns of one plane, where unrelated sequences differ in about half: a subject
// passes it with probability P(Binomial(cols, 1/2) <= bound).  The filter-plane-resident kernels pay off while a
// wave's 1024 subjects rarely produce a pass, i.e. while that tail stays below ~2e-3 — for cols = 32 this is
// bound <= 7, the measured crossover (profiles/r01_lazy_vs_resident.txt); short sequences need a tighter bound
// (cols = 20: bound <= 3; cols = 12: bound 0 — L = 12 with bound 2 ran 3x slower through these kernels).
static bool prefilter_prunes(const smafa_db *db, uint32_t bound) {
    const uint32_t cols = std::min<uint32_t>(32u, db->L);
    if (bound >= cols) return false;
    double term = 1.0, tail = 0.0;  // C(cols, k), summed for k = 0..bound
    for (uint32_t k = 0; k <= bound; k++) {
        tail += term;
        term = term * (double)(cols - k) / (double)(k + 1);
    }
    for (uint32_t i = 0; i < cols; i++) tail *= 0.5;
    return tail <= db->prune_p;
}

// Two words per plane: beyond the bounds level 1 prunes at, level 2 still rejects nearly every pair while the bound is well
// below what unrelated sequences score on it — the OR of the two words' mismatch bits has ~3/4 of the second word's columns
// set (+ half of the first word's columns that have no partner): bound <= half of that (L = 60: 12); then the per-word sums up
// to 14 (SUMFOLD).  There the filter-plane-resident kernel — 16 subjects per lane, one plane streamed — beats the all-planes
// one, which only ever uses its other planes for the pairs that pass: 10 000 queries x 10M aa, bound 8 / 9 / 10 / 12:
// 8.7 / 9.5 / 8.9 / 9.6 -> 7.8 / 8.3 / 8.4 / 9.1 ms, bound 14: 12.2 -> 11.3 (tools/bound_probe.py, profiles/r04_bound_probe.txt).
// SMAFA_LAZY_FOLD=0: off.
static bool fold_rejects(const smafa_db *db, uint32_t bound) {
    if (!db->lazy_fold || db->W != 2 || db->L < 56) return false;  // (measured at 60 columns; shorter second words: not claimed)
    const uint32_t second = db->L - 32u;                                  // columns that have a partner in the other word
    const uint32_t unrelated = (3u * second + 2u * (32u - second)) / 4u;  // expected popcount of the OR-fold (L = 60: 23)
    // OR-fold up to 12, the per-word sums (SUMFOLD) at 13 and 14; from 15 on too many wave steps pass level 2 and fetch their
    // tiles from L2 (bound 16: 18.1 ms against 13.8 for the all-planes kernel, whose tiles are resident)
    return bound * 2u <= unrelated + 1u || bound <= 14u;
}

static bool use_lazy(const smafa_db *db, uint32_t thr0) {
    const bool wide = db->W >= db->wide_from || (db->W == 1 && db->wide_one);
    return db->lazy && db->use_filter && db->W <= 4 && !wide && (prefilter_prunes(db, thr0) || fold_rejects(db, thr0));
}

// More than four words per plane (L > 128): scan_wide_kernel under the same rule — its levels 1 and 2 are the lazy
// kernel's, with 16 subjects per lane whatever the length.  One-word stores (L <= 32) take it too: its level 2
// folds a second plane, which a single filter word needs (up to 1.8x on sparse hits, equal elsewhere).  At W = 3, 4
// it is 8-25 % faster than the per-length kernels on sparse hits but 1.3-2x slower on dense or closely related
// stores (tools/dense_check.py: their level 2 folds every filter word and their full comparison keeps the tile in
// registers), so those lengths keep them; SMAFA_WIDE_FROM=3 switches them over (profiles/r01_wide_vs_lazy.txt).
// Bound too loose, or prefilter off: scan_kernel (W <= 4) / scan_generic_kernel.
static bool use_wide(const smafa_db *db, uint32_t thr0) {
    const bool one = db->W == 1 && db->wide_one;
    return db->lazy && db->use_filter && (db->W >= db->wide_from || one) && prefilter_prunes(db, thr0) &&
           wide_fits((int)db->PQ, (int)db->W);
}

// P(Binomial(bits, 1/2) <= bound): how often `bits` shared filter bits of a tile fail to exclude a random query
static double binom_tail(uint32_t bits, uint32_t bound) {
    if (bound >= bits) return 1.0;
    double term = 1.0, tail = 0.0;
    for (uint32_t k = 0; k <= bound; k++) {
        tail += term;
        term = term * (double)(bits - k) / (double)(k + 1);
    }
    for (uint32_t i = 0; i < bits; i++) tail *= 0.5;
    return tail;
}

// Does the zone level pay?  A tile that shares b filter bits lets a query unrelated to it through with probability
// P(Binomial(b, 1/2) <= bound); the expected share of (query, tile) pairs that pass follows from the store's measured
// shared-bit histogram (zone_hist).  Measured (tools/zone_threshold.sh, profiles/r02_zone_threshold.txt): the zone
// kernel wins while that share stays below ~0.6 — 1M rows at bound 5 (~12 bits, 0.39): 0.51 vs 0.64 ms; 250k rows at
// bound 5 (~10 bits, 0.62): 0.225 vs 0.208 ms; 10M rows at bound 7 (~15 bits, 0.50): 5.8 vs 6.5 ms.
static double zone_pass_share(const smafa_db *db, uint32_t thr0) {
    double tiles = 0.0, pass = 0.0;
    for (uint32_t b = 0; b <= 64; b++) {
        if (!db->zone_hist[b]) continue;
        tiles += (double)db->zone_hist[b];
        pass += (double)db->zone_hist[b] * binom_tail(b, thr0);
    }
    return tiles > 0.0 ? pass / tiles : 1.0;
}

// `prunes`: does level 1 (word 0 of the filter plane) prune at this bound (prefilter_prunes)?  Where it does not — short
// sequences, loose bounds: every (query, tile) pair that passes the zone level goes on to the exact comparison — the
// zone level has to exclude more on its own to beat the all-planes kernel: SMAFA_ZONE_LOOSE (default 0.3).
static bool zone_pays(const smafa_db *db, uint32_t thr0, bool prunes) {
    if (!db->lazy || !db->use_filter) return false;
    if (db->zone != 1) return db->zone == 2;
    // (five planes of four words: the survivors' levels 2-3 fetch 20 vectors per tile from L2 — the crossover comes
    // earlier: aa 128 columns at bound 7, share 0.5: 11.7 ms vs 9.5 ms without the zone level; 80 columns: 6.6 vs 7.8)
    const double pays = db->W <= 4 && db->P * db->W >= 20 ? 0.4 : 0.6;  // (scan_wide_kernel's own zone level: 0.6)
    return zone_pass_share(db, thr0) < (prunes ? pays : db->zone_loose);
}
// up to 128 columns: scan_zone_kernel; longer: the zone level inside scan_wide_kernel (ScanArgs::zone_on)
static bool use_zone(const smafa_db *db, uint32_t thr0, bool prunes) { return db->W <= 4 && zone_pays(db, thr0, prunes); }

static uint32_t tiles_per_wave(const smafa_db *db, bool lazy, uint32_t thr0) {
    if (lazy) return db->W >= 3 ? 2u : 4u;  // every filter word resident: 8 subjects per lane from 3 words on
    if (db->W > 2) return 1;
    if (db->tiles_override == 4) return db->P == 2 ? 4u : 2u;
    if (db->tiles_override == 1 || db->tiles_override == 2) return db->tiles_override;
    // Two tiles per wave share the per-query work of the bound levels between 8 subjects per lane — which pays while those
    // levels reject most pairs.  Where (nearly) every pair gets the full comparison — prefilter off, or a bound above 16 with
    // four and more planes (FOLD 2 / 3) — one tile per wave is faster: 80 registers less, more waves resident, and for a one-query
    // pass shorter waves that keep the memory pipeline full.  10 000 x 10M aa: prefilter off 30.5 -> 25.4 ms, bound 24
    // 15.0 -> 14.4 ms, best hit without a bound 23.1 -> 21.8 ms, bound 8 the other way (8.3 -> 9.6 ms: stays at two); one query
    // streaming every plane of the 10M store: 70.3 -> 58.4 us = 0.72 -> 0.86 of HBM peak (profiles/r03_stream_nt.txt).
    // (nucleotide stores the same way, less to gain: best hit without a bound, half the queries unrelated, 13.6 -> 12.8 ms)
    if (!db->use_filter || thr0 > 16u) return 1u;
    return 2u;
}

template <int PS, int PQ>
static void launch_wide_t(const smafa_db *db, const uint32_t *d_qrec, const ScanArgs &a, uint32_t grid) {
    const bool seed = a.hits == nullptr && a.k_tight == 1;
    const uint4 *planes = reinterpret_cast<const uint4 *>(db->d_planes);
    // resident filter words per subject: 1 = one-word store (plus word 0 of a second plane), else 3 (a fourth
    // pushes the kernel past 128 VGPRs: measured spills, and one wave per SIMD less)
    const uint32_t fw = db->W == 1 ? 1u : 3u;
    const uint32_t wc = (db->W == 3 || db->W == 4) ? db->W : 0u;  // compile-time word count: register-resident dense walk
#define SMAFA_WIDE(FW_, WC_)                                                                                     \
    if (fw == FW_ && wc == WC_) {                                                                               \
        note_kernel(db, "smafa::scan_wide_kernel<%d, %d, %s, %d, %d>%s", PS, PQ, seed ? "true" : "false", FW_, WC_, \
                    a.zone_on ? " (zone level on)" : "");                                                       \
        if (seed)                                                                                               \
            hipLaunchKernelGGL((scan_wide_kernel<PS, PQ, true, FW_, WC_>), dim3(grid), dim3(256), 0, db->stream, \
                               planes, d_qrec, a, db->W);                                                       \
        else                                                                                                    \
            hipLaunchKernelGGL((scan_wide_kernel<PS, PQ, false, FW_, WC_>), dim3(grid), dim3(256), 0, db->stream, \
                               planes, d_qrec, a, db->W);                                                       \
        return;                                                                                                 \
    }
    SMAFA_WIDE(1, 0) SMAFA_WIDE(3, 0) SMAFA_WIDE(3, 3) SMAFA_WIDE(3, 4)
#undef SMAFA_WIDE
}

static void launch_scan(const smafa_db *db, const uint32_t *d_qrec, const ScanArgs &a, uint32_t grid, uint32_t T,
                        bool lazy, bool zone) {
    if (lazy && !zone && (db->W >= db->wide_from || (db->W == 1 && db->wide_one))) {  // above 64 columns, or up to 32
        if (db->P == 2) return launch_wide_t<2, 3>(db, d_qrec, a, grid);
        if (db->P == 3) return launch_wide_t<3, 3>(db, d_qrec, a, grid);
        return launch_wide_t<5, 5>(db, d_qrec, a, grid);
    }
    if (lazy && zone) {  // sorted store, bound the zone level prunes at
#define SMAFA_ZONE(PS_, PQ_, W_)                                 \
    if (db->P == PS_ && db->PQ == PQ_ && db->W == W_) {          \
        launch_zone_t<PS_, PQ_, W_>(db, d_qrec, a, grid);        \
        return;                                                  \
    }
        SMAFA_ZONE(2, 3, 1) SMAFA_ZONE(3, 3, 1) SMAFA_ZONE(5, 5, 1) SMAFA_ZONE(2, 3, 2) SMAFA_ZONE(3, 3, 2) SMAFA_ZONE(5, 5, 2)
        SMAFA_ZONE(2, 3, 3) SMAFA_ZONE(3, 3, 3) SMAFA_ZONE(5, 5, 3) SMAFA_ZONE(2, 3, 4) SMAFA_ZONE(3, 3, 4) SMAFA_ZONE(5, 5, 4)
#undef SMAFA_ZONE
    }
    if (lazy) {  // filter-plane-resident kernel
#define SMAFA_LAZY(PS_, PQ_, W_, T_)                                  \
    if (db->P == PS_ && db->PQ == PQ_ && db->W == W_ && T == T_) {    \
        launch_lazy_t<PS_, PQ_, W_, T_>(db, d_qrec, a, grid);         \
        return;                                                       \
    }
        SMAFA_LAZY(2, 3, 2, 4) SMAFA_LAZY(3, 3, 2, 4) SMAFA_LAZY(5, 5, 2, 4)
        SMAFA_LAZY(2, 3, 1, 4) SMAFA_LAZY(3, 3, 1, 4) SMAFA_LAZY(5, 5, 1, 4)
        SMAFA_LAZY(2, 3, 3, 2) SMAFA_LAZY(3, 3, 3, 2) SMAFA_LAZY(5, 5, 3, 2)
        SMAFA_LAZY(2, 3, 4, 2) SMAFA_LAZY(3, 3, 4, 2) SMAFA_LAZY(5, 5, 4, 2)
#undef SMAFA_LAZY
    }
#define SMAFA_CASE(PS_, PQ_, W_, T_)                          \
    if (db->P == PS_ && db->PQ == PQ_ && db->W == W_ && T == T_) { \
        launch_scan_t<PS_, PQ_, W_, T_>(db, d_qrec, a, grid); \
        return;                                               \
    }
    SMAFA_CASE(2, 3, 1, 1) SMAFA_CASE(2, 3, 2, 1) SMAFA_CASE(2, 3, 3, 1) SMAFA_CASE(2, 3, 4, 1)
    SMAFA_CASE(3, 3, 1, 1) SMAFA_CASE(3, 3, 2, 1) SMAFA_CASE(3, 3, 3, 1) SMAFA_CASE(3, 3, 4, 1)
    SMAFA_CASE(5, 5, 1, 1) SMAFA_CASE(5, 5, 2, 1) SMAFA_CASE(5, 5, 3, 1) SMAFA_CASE(5, 5, 4, 1)
    SMAFA_CASE(2, 3, 1, 2) SMAFA_CASE(2, 3, 2, 2) SMAFA_CASE(3, 3, 1, 2) SMAFA_CASE(3, 3, 2, 2)
    SMAFA_CASE(5, 5, 1, 2) SMAFA_CASE(5, 5, 2, 2)
    SMAFA_CASE(2, 3, 1, 4) SMAFA_CASE(2, 3, 2, 4)
#undef SMAFA_CASE
    note_kernel(db, "smafa::scan_generic_kernel");
    hipLaunchKernelGGL(scan_generic_kernel, dim3(grid), dim3(256), 0, db->stream,
                       reinterpret_cast<const uint4 *>(db->d_planes), d_qrec, a, db->P, db->PQ, db->W, db->QS);
}

// queries per workgroup pass: big enough that the tile load is amortised (the scan is then bound by
// VALU issue, not HBM), small enough that the grid has many more workgroups than the chip has slots.
static uint32_t choose_query_block(const smafa_db *db, uint32_t n_wg_tiles, uint32_t nq) {
    if (db->qb_override) return std::min<uint32_t>(std::max<uint32_t>(db->qb_override, 1u), std::max(nq, 1u));
    const uint32_t slots = (uint32_t)db->n_cu * 6u;  // 6 workgroups of 4 waves per CU at the kernel's register budget
    const uint32_t want_items = slots * 16u;
    uint32_t nqb = (want_items + n_wg_tiles - 1) / n_wg_tiles;
    const uint32_t max_nqb = std::max(1u, nq / 256u);
    nqb = std::max(1u, std::min(nqb, max_nqb));
    uint32_t qb = (nq + nqb - 1) / nqb;
    return std::max(qb, 1u);
}

// ---------------------------------------------------------------------------------------------
// The block index (index.hip.h).
static bool index_current(const smafa_db *db) {
    const auto &ix = db->index;
    return ix.valid && ix.generation == db->generation && ix.n == db->n && ix.resorts == db->resorts && ix.P == db->P;
}

static void index_drop(smafa_db *db) {
    db->index.valid = false;
    for (DevBuf *b : {&db->index.kp, &db->index.dir, &db->index.stats, &db->index.rows}) b->release();
}

// Build (or rebuild) the index with `blocks` blocks: serves every fixed bound up to blocks - 1.
static int index_build(smafa_db *db, uint32_t blocks) {
    auto &ix = db->index;
    ix.valid = false;
    if (db->W > (uint32_t)kIndexMaxWords)
        return set_error(SMAFA_ERR_INVALID, "the block index takes rows of up to %d columns (this store: %u)", kIndexMaxWords * 32, db->L);
    if (blocks < 1 || blocks > (uint32_t)kIndexMaxBlocks || blocks > db->L)
        return set_error(SMAFA_ERR_INVALID, "the block index takes 1..%u blocks for this store (asked: %u)",
                         std::min<uint32_t>(kIndexMaxBlocks, db->L), blocks);
    if (db->n == 0 || db->n >= (1ull << 31)) return set_error(SMAFA_ERR_INVALID, "the block index takes 1..2^31-1 subjects");
    int rc = use_device(db);
    if (rc) return rc;
    rc = maybe_resort(db);  // (positions are final afterwards: an index built before a due re-sort would be stale at once)
    if (rc) return rc;
    const double t_begin = now_seconds();
    const uint32_t n = (uint32_t)db->n;
    uint32_t dir_bits = 8;
    while (dir_bits < 22 && (1ull << (dir_bits + 2)) < n) dir_bits++;
    const size_t dir_entries = ((size_t)1 << dir_bits) + 1;
    const size_t row_bytes = (size_t)index_row_vectors((int)db->P, (int)db->W) * sizeof(uint4);
    rc = ix.kp.ensure((size_t)blocks * n * sizeof(uint2));
    if (!rc) rc = ix.rows.ensure((size_t)n * row_bytes);
    if (!rc) rc = ix.dir.ensure((size_t)blocks * dir_entries * sizeof(uint32_t));
    if (!rc) rc = ix.stats.ensure((size_t)kIndexMaxBlocks * 2 * sizeof(unsigned long long));
    if (!rc) rc = db->keys_a.ensure((size_t)n * sizeof(uint32_t));
    if (!rc) rc = db->keys_b.ensure((size_t)n * sizeof(uint32_t));
    if (!rc) rc = db->idx_a.ensure((size_t)n * sizeof(uint32_t));
    if (!rc) rc = db->idx_b.ensure((size_t)n * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t *ka = db->keys_a.as<uint32_t>(), *ia = db->idx_a.as<uint32_t>();
    uint32_t *kb = db->keys_b.as<uint32_t>(), *ib = db->idx_b.as<uint32_t>();
    size_t tmp_bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ka, kb, ia, ib, (int)n, 0, 32, db->stream));
    rc = db->sort_tmp.ensure(tmp_bytes);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ix.stats.p, 0, (size_t)kIndexMaxBlocks * 2 * sizeof(unsigned long long), db->stream));
    const uint32_t grid = (n + 255u) / 256u;
    for (uint32_t b = 0; b <= blocks; b++) ix.col_begin[b] = (uint16_t)((uint64_t)b * db->L / blocks);
    hipLaunchKernelGGL(index_rows_kernel, dim3(grid), dim3(256), 0, db->stream, db->d_planes, db->P, db->W, n, ix.rows.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    for (uint32_t b = 0; b < blocks; b++) {
        hipLaunchKernelGGL(index_keys_kernel, dim3(grid), dim3(256), 0, db->stream, db->d_planes, db->P, db->W, n,
                           (uint32_t)ix.col_begin[b], (uint32_t)ix.col_begin[b + 1], ka, ia);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(db->sort_tmp.p, tmp_bytes, ka, kb, ia, ib, (int)n, 0, 32, db->stream));
        hipLaunchKernelGGL(index_dir_kernel, dim3((uint32_t)((dir_entries + 255) / 256)), dim3(256), 0, db->stream, kb, n, dir_bits,
                           ix.dir.as<uint32_t>() + (size_t)b * dir_entries);
        hipLaunchKernelGGL(index_stats_kernel, dim3(std::min<uint32_t>(grid, 2048u)), dim3(256), 0, db->stream, kb, n,
                           ix.stats.as<unsigned long long>() + (size_t)b * 2);
        hipLaunchKernelGGL(index_interleave_kernel, dim3(grid), dim3(256), 0, db->stream, kb, ib, n, ix.kp.as<uint2>() + (size_t)b * n);
        HIP_TRY(hipGetLastError());
    }
    unsigned long long st[kIndexMaxBlocks * 2] = {0};
    HIP_TRY(hipMemcpyAsync(st, ix.stats.p, (size_t)blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, db->stream));
    HIP_TRY(hipStreamSynchronize(db->stream));
    for (uint32_t b = 0; b < blocks; b++) {
        ix.max_run[b] = st[2 * b];
        ix.mean_run[b] = (double)st[2 * b + 1] / (double)n;
    }
    ix.B = blocks;
    ix.dir_bits = dir_bits;
    ix.generation = db->generation;
    ix.n = db->n;
    ix.resorts = db->resorts;
    ix.P = db->P;
    ix.build_ms = (now_seconds() - t_begin) * 1e3;
    ix.valid = true;
    if (n > (1u << 20))
        for (DevBuf *b : {&db->keys_a, &db->keys_b, &db->idx_a, &db->idx_b, &db->sort_tmp}) b->release();
    log_line(2, "block index of %u rows: %u blocks, %.1f MB, built on the device in %.2f ms", n, blocks,
             (double)(ix.kp.cap + ix.dir.cap + ix.rows.cap) / 1e6, ix.build_ms);
    return SMAFA_OK;
}

// Candidates per query the probes of a scan with this bound may expect before the scan kernels are the better choice.
// Measured on 10M-subject stores (profiles/r04_index.txt): a candidate costs ~0.025 ns (its row of the index's row-major copy,
// one or two cache lines; gathered from the bit-planes it was 0.18 ns for 60 amino-acid columns).  A scanned subject costs the
// batched kernels, per query: the zone kernel ~8e-14 s x the share of (query, tile) pairs that pass the zone level (level 1
// over the tile for each of them: aa bound 5, share 0.2: 1.7e-14 s; bound 3: 3.7e-15; nt bound 3, share 0.025: 2.9e-15; bound 7:
// 5.3e-14), the filter-plane-resident kernel at the bounds its folds reject 8e-14 s (aa 8..12, nt 9), the all-planes forms
// 1.2e-13 s and more.  The candidates allowed are 0.45 of break-even — queries are not spread like the store's own rows — and an
// underestimate of the scan only leaves a scan where the index would have been faster.
static double index_cand_limit(const smafa_db *db, uint32_t thr0) {
    double per_subject = db->index_cand_per_subject;
    if (per_subject < 0.0) {
        const bool prunes = prefilter_prunes(db, thr0);
        const double scan_s = db->W <= 4 && use_zone(db, thr0, prunes) ? 8e-14 * std::min(1.0, std::max(0.02, zone_pass_share(db, thr0)))
                              : fold_rejects(db, thr0) || prunes        ? 8e-14
                                                                        : 1.2e-13;
        per_subject = 0.45 * scan_s / 0.025e-9;
    }
    return std::max(16.0, per_subject * (double)db->n);
}

// Which blocks a fixed-bound scan would probe, and whether that beats the scan kernels: bound + 1 blocks out of the usable
// ones (longest run within index_max_run), the ones with the fewest expected candidates.
static bool index_plan(const smafa_db *db, uint32_t thr0, uint32_t nq, uint8_t *probe_block) {
    const auto &ix = db->index;
    if (!db->index_mode || !db->use_filter || !index_current(db) || nq <= 64u || thr0 + 1u > ix.B) return false;
    uint8_t usable[kIndexMaxBlocks];
    uint32_t nu = 0;
    for (uint32_t b = 0; b < ix.B; b++)
        if (ix.max_run[b] <= db->index_max_run) usable[nu++] = (uint8_t)b;
    if (nu < thr0 + 1u) return false;
    std::sort(usable, usable + nu, [&](uint8_t x, uint8_t y) { return ix.mean_run[x] < ix.mean_run[y] || (ix.mean_run[x] == ix.mean_run[y] && x < y); });
    double expected = 0.0;
    for (uint32_t j = 0; j <= thr0; j++) expected += ix.mean_run[usable[j]];
    if (expected > index_cand_limit(db, thr0)) return false;
    for (uint32_t j = 0; j <= thr0; j++) probe_block[j] = usable[j];
    return true;
}

static int index_probe(smafa_db *db, smafa_qset *qs, uint32_t q_begin, uint32_t q_end, uint32_t thr0, const uint8_t *probe_block,
                       smafa_hit *d_rows, uint64_t rows_cap, unsigned long long *d_count) {
    const auto &ix = db->index;
    IndexArgs x;
    x.kp = ix.kp.as<uint2>();
    x.rows = ix.rows.as<uint4>();
    x.dir = ix.dir.as<uint32_t>();
    x.n = (uint32_t)ix.n;
    x.dir_bits = ix.dir_bits;
    x.n_probes = thr0 + 1u;
    x.bound = thr0;
    x.L = db->L;
    x.QS = db->QS;
    x.q_begin = q_begin;
    x.q_end = q_end;
    for (uint32_t j = 0; j < (uint32_t)kIndexMaxBlocks; j++) {
        x.probe_block[j] = j <= thr0 ? probe_block[j] : 0;
        x.probe_cols[j] = (uint32_t)ix.col_begin[x.probe_block[j]] | ((uint32_t)ix.col_begin[x.probe_block[j] + 1] << 16);
    }
    ScanArgs a{};
    a.n_subjects = (uint32_t)db->n;
    a.hits = d_rows;
    a.cap = rows_cap;
    a.count = d_count;  // zeroed by the caller; rows are reserved straight from it and it IS the result
    a.order = db->d_order;
    const uint64_t groups = (uint64_t)(q_end - q_begin) * x.n_probes;
    const uint64_t grid = (groups * kIndexGroup + kIndexWg - 1u) / kIndexWg;
    if (grid > 0x7fffffffull) return set_error(SMAFA_ERR_INVALID, "index probe grid too large (%llu workgroups)", (unsigned long long)grid);
    int cur_dev = -1;
    if (hipGetDevice(&cur_dev) == hipSuccess) {
        db->launch_device = cur_dev;
        if (cur_dev != db->device) db->launches_off_device++;
    }
    const uint32_t *qrec = qs->qrec.as<uint32_t>();
    bool launched = false;
#define SMAFA_PROBE(PS_, PQ_, W_)                                                                                             \
    if (!launched && db->P == PS_ && db->PQ == PQ_ && db->W == W_) {                                                          \
        hipLaunchKernelGGL((index_probe_kernel<PS_, PQ_, W_>), dim3((uint32_t)grid), dim3(kIndexWg), 0, db->stream, db->d_planes, qrec, x, a); \
        note_kernel(db, "smafa::index_probe_kernel<%d, %d, %d>", PS_, PQ_, W_);                                               \
        launched = true;                                                                                                      \
    }
    SMAFA_PROBE(2, 3, 1) SMAFA_PROBE(3, 3, 1) SMAFA_PROBE(5, 5, 1) SMAFA_PROBE(2, 3, 2) SMAFA_PROBE(3, 3, 2) SMAFA_PROBE(5, 5, 2)
    SMAFA_PROBE(2, 3, 3) SMAFA_PROBE(3, 3, 3) SMAFA_PROBE(5, 5, 3) SMAFA_PROBE(2, 3, 4) SMAFA_PROBE(3, 3, 4) SMAFA_PROBE(5, 5, 4)
#undef SMAFA_PROBE
    if (!launched) return set_error(SMAFA_ERR_INVALID, "no index probe for %u/%u planes x %u words", db->P, db->PQ, db->W);
    HIP_TRY(hipGetLastError());
    db->plan_lazy = 0;
    db->plan_tiles = 0;
    db->plan_qblocks = 1;
    db->last_launches++;
    db->index_probes++;
    return SMAFA_OK;
}

// one kernel launch: queries [q_begin, q_end) x wave tiles [tile_begin, tile_end).  Rows go to `d_rows` (room for
// `rows_cap`) through the handle's counter; publish != NULL: the launch is the whole scan — its last workgroup writes the
// row total to *publish and leaves the counters at zero.
static int launch_tiles(smafa_db *db, smafa_qset *qs, uint32_t q_begin, uint32_t q_end, uint32_t tile_begin,
                        uint32_t tile_end, uint32_t k_tight, uint32_t thr0, smafa_hit *d_rows, uint64_t rows_cap,
                        unsigned long long *publish, bool per_query_bounds = false, unsigned long long *own_counter = nullptr) {
    ScanArgs a;
    const bool specialised = db->W <= 4;  // else scan_wide_kernel / scan_generic_kernel
    const bool wide = use_wide(db, thr0);
    bool lazy = wide || (specialised && use_lazy(db, thr0));
    const bool seed = d_rows == nullptr && k_tight == 1;  // the seed pass covers a few tiles: no zone level
    // a sorted store whose tiles share enough bits takes the zone kernel at any length up to 128 columns — also where
    // scan_wide_kernel would otherwise run (one-word stores)
    const bool zone = specialised && !seed && use_zone(db, thr0, prefilter_prunes(db, thr0));
    lazy = lazy || zone;  // (the plan reported by smafa_last_scan_plan: a filter-plane-resident kernel)
    // (the unstaged form of the zone kernel — launch_zone_t's `direct` — has its own tile count per shape)
    const bool zone_is_direct = !(k_tight || per_query_bounds) && db->zone_direct && d_rows != nullptr;
    const uint32_t T = zone ? (q_end - q_begin <= 64u ? (uint32_t)kFewTiles : (uint32_t)zone_tiles((int)db->P, (int)db->W, zone_is_direct))
                     : wide ? (uint32_t)kWideTiles : specialised ? tiles_per_wave(db, lazy, thr0) : (uint32_t)kGenericTiles;
    a.tile_begin = tile_begin;
    a.tile_end = tile_end;
    const uint32_t wg_waves = (zone && q_end - q_begin > 64u) ? (uint32_t)kZoneWgWaves : (uint32_t)kWgWaves;
    a.n_wg_tiles = (tile_end - tile_begin + wg_waves * T - 1) / (wg_waves * T);
    a.n_subjects = (uint32_t)db->n;
    a.q_begin = q_begin;
    a.q_end = q_end;
    // (the block size is chosen per 4-wave share of the store whatever the workgroup size, so that it does not change
    // with kZoneWgWaves: profiles/r02_zone_variants.txt)
    a.qb_size = choose_query_block(db, (tile_end - tile_begin + kWgWaves * T - 1) / (kWgWaves * T), q_end - q_begin);
    // fixed common bound: no per-query array, no fill launch; per_query_bounds: fixed bounds read from thr
    a.thr = (k_tight || per_query_bounds) ? qs->thr.as<uint32_t>() : nullptr;
    a.thr0 = thr0;
    a.cnt = qs->cnt.as<uint32_t>();
    a.cnt_stride = db->L + 1;
    a.k_tight = k_tight;
    a.use_filter = db->use_filter ? 1u : 0u;
    a.hits = d_rows;
    a.cap = rows_cap;
    a.count = db->ctrs.as<unsigned long long>();
    a.done = publish ? reinterpret_cast<uint32_t *>(db->ctrs.as<uint8_t>() + 256) : nullptr;
    a.publish = publish;
    if (own_counter) {  // the caller zeroed a counter of its own for this launch: rows are reserved from it and it IS the result —
        a.count = own_counter;  // nobody takes a ticket (smafa_scan_each: a workgroup's returning ticket atomic at the end of
        a.done = nullptr;       // a one-query pass cost the streaming form 13 % — profiles/r03_stream_nt.txt)
        a.publish = nullptr;
    }
    a.order = db->d_order;
    a.zone = db->d_zone;
    a.zone_on = (wide && !zone && !seed && db->W > 4 && zone_pays(db, thr0, true)) ? 1u : 0u;
    const uint64_t n_qblocks = (q_end - q_begin + a.qb_size - 1) / a.qb_size;
    // non-temporal loads (scan_lazy_kernel's filter words, scan_kernel's tiles): where a cached copy is never read again — one
    // query block, or more bytes per query block than the 256 MiB Infinity Cache holds until the next one comes round
    // (bytes one query block reads: the filter plane's words for the filter-plane-resident kernels, whole tiles for scan_kernel)
    const uint64_t range_bytes = (uint64_t)(tile_end - tile_begin) * (lazy ? db->W : db->P * db->W) * 1024u;
    a.stream_once = (db->stream_nt && (n_qblocks == 1 || range_bytes >= (256ull << 20))) ? 1u : 0u;
    const uint64_t grid = n_qblocks * a.n_wg_tiles;
    if (grid > 0x7fffffffull)
        return set_error(SMAFA_ERR_INVALID, "scan grid too large (%llu workgroups)", (unsigned long long)grid);
    int cur_dev = -1;
    if (hipGetDevice(&cur_dev) == hipSuccess) {
        db->launch_device = cur_dev;
        if (cur_dev != db->device) db->launches_off_device++;
    }
    launch_scan(db, qs->qrec.as<uint32_t>(), a, (uint32_t)grid, T, lazy, zone);
    db->plan_lazy = lazy ? 1u : 0u;
    db->plan_tiles = T;
    db->plan_qblocks = (uint32_t)n_qblocks;
    HIP_TRY(hipGetLastError());
    db->last_launches++;
    return SMAFA_OK;
}

// Scan queries [q_begin, q_end) of a resident set against the whole store; rows and their count stay on
// the device.
// k_tight = 0: ONE launch (up to 64 queries: a one-workgroup kernel zeroes *d_count first and nobody takes a ticket), fixed
// bound max_div: the kernel appends straight into the caller's list and — big batches — its last
// workgroup publishes the total in *d_count (which may exceed cap: the rows past it are dropped, the count is exact).
// k_tight >= 1: the bound of each query is lowered to its running k-th smallest distance while the scan proceeds.
// Workgroups of one launch run side by side and would all start from the loose initial bound, so the store is walked
// in segments that grow 8x per launch (bounds tighten between launches), after a seed launch over the first segment
// that only lowers the bounds and appends nothing.  Rows are appended to a scratch list while the bounds are still
// running; filter_rows_kernel then keeps the ones within the final bounds.
static int scan_range(smafa_db *db, smafa_qset *qs, uint32_t q_begin, uint32_t q_end, uint32_t max_div,
                      uint32_t k_tight, smafa_hit *d_hits, uint64_t cap, unsigned long long *d_count) {
    const uint32_t nq = q_end - q_begin;
    db->last_launches = 0;
    db->timed = false;
    if (nq == 0 || db->n == 0) {
        HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), db->stream));
        return SMAFA_OK;
    }
    int prc = maybe_resort(db);
    if (prc) return prc;
    unsigned long long *d_ctr = db->ctrs.as<unsigned long long>();
    const uint32_t thr0 = std::min<uint32_t>(max_div, db->L);  // a distance never exceeds seq_len
    const uint32_t n_tiles = (uint32_t)((db->n + kWaveTile - 1) / kWaveTile);
    if (k_tight == 0) {
        // A store with a current block index answers a tight fixed bound from it: bound + 1 probes per query instead of a
        // pass over every tile (index.hip.h).  Mode 2 builds the index the first time such a scan arrives.
        if (db->index_mode >= 2 && db->use_filter && nq > 64u && db->W <= (uint32_t)kIndexMaxWords && thr0 + 1u <= std::min<uint32_t>(kIndexMaxBlocks, db->L) &&
            db->n >= db->index_min_rows && db->n < (1ull << 31) && (!index_current(db) || db->index.B < thr0 + 1u)) {
            bool build = db->index_mode == 2;
            if (!build) {
                // rent or buy: the scans that could have used an index are charged at the batched kernels' measured rate
                // (1.7e-12 ms per pair and stored vector: profiles/r04_bench_full.json), the build at ~1 ms per block and 10M
                // subjects (profiles/r04_index.txt); the index is built once the rent paid equals its price
                if (db->index_debt_generation != db->generation) db->index_debt_ms = 0.0, db->index_debt_generation = db->generation;
                db->index_debt_ms += (double)nq * (double)db->n * (double)(db->P * db->W) * 1.7e-12;
                build = db->index_debt_ms >= 0.3 + (double)(thr0 + 1u) * (double)db->n * 1.0e-7;
            }
            if (build && db->index_failed_generation != db->generation + 1u) {
                // (a build that fails — no room for another 8 B x blocks + a row copy per subject — is not the scan's failure:
                // the scan kernels answer, and the automatic modes do not try again until the store changes)
                if (index_build(db, thr0 + 1u) != SMAFA_OK) {
                    log_line(1, "block index not built (%s): scanning as before", smafa_last_error());
                    index_drop(db);
                    db->index_failed_generation = db->generation + 1u;
                }
            }
        }
        uint8_t probe_block[kIndexMaxBlocks];
        if (index_plan(db, thr0, nq, probe_block)) {
            hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, db->stream, (uint32_t *)d_count, 0u, (uint64_t)2);
            HIP_TRY(hipEventRecord(db->ev0, db->stream));
            int rc = index_probe(db, qs, q_begin, q_end, thr0, probe_block, d_hits, cap, d_count);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(db->ev1, db->stream));
            db->timed = true;
            return SMAFA_OK;
        }
        // A handful of queries against a big store is a grid of many short-lived workgroups: there the ticket every workgroup
        // takes at its end (to find the last one, which publishes the total) costs more than a tiny fill kernel in front of the launch —
        // the rows are then reserved straight from *d_count (one-query pass over the 50M store: 65 -> 57 us streaming,
        // 22.4 -> 18.5 us with the zone level; profiles/r03_stream_nt.txt).  Big batches keep the one-launch form.
        const bool own = nq <= 64u;
        if (own) hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, db->stream, (uint32_t *)d_count, 0u, (uint64_t)2);
        HIP_TRY(hipEventRecord(db->ev0, db->stream));
        int rc = launch_tiles(db, qs, q_begin, q_end, 0, n_tiles, 0, thr0, d_hits, cap, own ? nullptr : d_count, false,
                              own ? d_count : nullptr);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(db->ev1, db->stream));
        db->timed = true;
        return SMAFA_OK;
    }
    // Tightening modes append every pair that is within its query's bound at that moment — 50-100 rows per query
    // when the bound starts loose — of which the filter pass keeps the ones within the final bound: the scratch list
    // is sized for the appended volume, the caller's buffer only has to hold what is kept.
    // (k >= 2 with a loose bound counts first and appends only the final rows, straight into the caller's list.)
    const bool count_first = k_tight >= db->count_first_k && !prefilter_prunes(db, thr0);
    // Counting first costs TWO passes over every pair (count, then append with the exact bounds).  On a big store the first
    // one is cut to a SAMPLE — the first 1/32 of the tiles: the k-th smallest distance within any subset of the subjects is an
    // upper bound of the k-th smallest over all of them — and the rest of the store is scanned ONCE, counting, tightening and
    // appending to the scratch list from that bound on; the counts are then complete up to the k-th distance (the bound never
    // dropped below it), so kth_from_counts_kernel gives the exact bounds, the sample's own tiles are scanned again with
    // those fixed (appending), and filter_rows_kernel keeps what is within them: (1 + 1/32) passes instead of 2
    // (10 000 queries x 10M aa, k = 5 / 50, sample 1/8: 34 / 39 ms, 1/16: 30 / 35, 1/32: 28 / 35; profiles/r04_kth.txt).
    // Rows parked meanwhile: ~k x (segment / store seen before it) per segment, a few k per query (SMAFA_KTH_SAMPLE=0: two passes).
    const uint32_t sample_tiles = (count_first && db->kth_sample_div && n_tiles >= db->kth_sample_min_tiles &&
                                   n_tiles / db->kth_sample_div >= 1u &&
                                   (uint64_t)k_tight * 4u <= (uint64_t)(n_tiles / db->kth_sample_div) * kWaveTile)
                                      ? n_tiles / db->kth_sample_div : 0u;
    uint64_t scratch_rows = 0;
    if (!count_first || sample_tiles) {
        const uint64_t per_query = sample_tiles ? 16ull * k_tight + 128u : 128u;
        scratch_rows = std::max<uint64_t>(2 * cap, std::min<uint64_t>((uint64_t)nq * per_query, 1ull << 27));
        int src = db->scratch.ensure(scratch_rows * sizeof(smafa_hit));
        if (src) return src;
    }
    smafa_hit *d_scratch = db->scratch.as<smafa_hit>();
    HIP_TRY(hipMemsetAsync(d_ctr, 0, sizeof(unsigned long long), db->stream));
    hipLaunchKernelGGL(fill_u32_kernel, dim3((nq + 255) / 256), dim3(256), 0, db->stream, qs->thr.as<uint32_t>() + q_begin,
                       thr0, (uint64_t)nq);
    const size_t cnt_stride = db->L + 1;
    auto zero_cnt = [&]() -> int {
        if (k_tight < 2) return SMAFA_OK;
        int rc = qs->cnt.ensure(std::max<uint64_t>(qs->nq, 64) * cnt_stride * sizeof(uint32_t));
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(qs->cnt.as<uint32_t>() + (size_t)q_begin * cnt_stride, 0,
                               (size_t)nq * cnt_stride * sizeof(uint32_t), db->stream));
        return SMAFA_OK;
    };
    HIP_TRY(hipEventRecord(db->ev0, db->stream));
    int rc = zero_cnt();
    // seed: k = 1 reduces each wave's minimum before its single atomicMin, so a whole workgroup tile is
    // cheap; the k >= 2 seed counts every pair in its histogram, so keep it to one wave tile
    const uint32_t seed_tiles = std::min<uint32_t>(k_tight == 1 ? kWgWaves : 1, n_tiles);
    // k >= 2: the k-th smallest distance within the first (up to) 1024 subjects, from an LDS histogram per query
    // (kth_seed_kernel: no global atomics; the counting launch it replaces put every pair of its tile through them)
    const bool hist_seed = k_tight >= 2 && db->L < (uint32_t)kSeedBins && db->kth_hist_seed;
    // ... and where a sample is counted first (sample_tiles), the same kernel counts the WHOLE sample: every pair of it, in LDS
    // histograms that are added to cnt[q][d] — exactly what the counting launches would have counted, without their global
    // atomics per pair and their growing segments (k = 50, 10 000 queries: 12 ms -> ~2 ms for a 1/32 sample)
    const bool hist_sample = hist_seed && sample_tiles != 0;
    auto launch_hist = [&](uint32_t tiles, uint32_t *d_cnt) {
        const uint32_t n_chunks = (nq + kSeedQueries - 1) / kSeedQueries;
        const uint32_t steps = (tiles + kWgWaves - 1) / kWgWaves;
        // enough workgroups to fill the chip (8 per CU), never more tile groups than 4-tile steps
        const uint32_t n_groups = d_cnt ? std::max(1u, std::min(steps, ((uint32_t)db->n_cu * 8u + n_chunks - 1) / n_chunks)) : 1u;
        const dim3 grid(n_chunks * n_groups), block(256);
        const uint4 *planes = reinterpret_cast<const uint4 *>(db->d_planes);
        const uint32_t *qrec = qs->qrec.as<uint32_t>();
        uint32_t *thr = qs->thr.as<uint32_t>();
#define SMAFA_SEED(PS_, PQ_, W_)                                                                                              \
    if (db->P == PS_ && db->PQ == PQ_ && db->W == W_) {                                                                       \
        hipLaunchKernelGGL((kth_seed_kernel<PS_, PQ_, W_>), grid, block, 0, db->stream, planes, qrec, db->QS, db->P, db->PQ,    \
                           db->W, tiles, (uint32_t)db->n, q_begin, q_end, n_chunks, n_groups, k_tight, thr0, thr, d_cnt,      \
                           (uint32_t)cnt_stride);                                                                             \
        return;                                                                                                               \
    }
        SMAFA_SEED(2, 3, 1) SMAFA_SEED(3, 3, 1) SMAFA_SEED(5, 5, 1) SMAFA_SEED(2, 3, 2) SMAFA_SEED(3, 3, 2) SMAFA_SEED(5, 5, 2)
        SMAFA_SEED(2, 3, 3) SMAFA_SEED(3, 3, 3) SMAFA_SEED(5, 5, 3) SMAFA_SEED(2, 3, 4) SMAFA_SEED(3, 3, 4) SMAFA_SEED(5, 5, 4)
#undef SMAFA_SEED
        hipLaunchKernelGGL((kth_seed_kernel<0, 0, 0>), grid, block, 0, db->stream, planes, qrec, db->QS, db->P, db->PQ, db->W, tiles,
                           (uint32_t)db->n, q_begin, q_end, n_chunks, n_groups, k_tight, thr0, thr, d_cnt, (uint32_t)cnt_stride);
    };
    if (!rc && hist_sample) {
        launch_hist(sample_tiles, qs->cnt.as<uint32_t>());
        HIP_TRY(hipGetLastError());
    } else if (!rc && hist_seed) {
        launch_hist(std::min<uint32_t>(kWgWaves, n_tiles), nullptr);
        HIP_TRY(hipGetLastError());
    } else {
        if (!rc) rc = launch_tiles(db, qs, q_begin, q_end, 0, seed_tiles, k_tight, thr0, nullptr, 0, nullptr);
        if (!rc) rc = zero_cnt();  // the seed's subjects are counted again below
    }
    uint32_t begin = 0, len = hist_seed ? 8u * kWgWaves : kWgWaves;  // (the histogram seed already stands for the first 4 tiles)
    const uint32_t count_end = sample_tiles ? sample_tiles : n_tiles;  // the tiles that are only counted
    if (hist_sample) begin = count_end;  // counted already, every pair of them
    while (!rc && begin < count_end) {
        const uint32_t end = (uint32_t)std::min<uint64_t>((uint64_t)begin + len, count_end);
        rc = launch_tiles(db, qs, q_begin, q_end, begin, end, k_tight, thr0, count_first ? nullptr : d_scratch,
                          count_first ? 0 : scratch_rows, nullptr);
        begin = end;
        len = len > (1u << 28) ? len : len * 8;
    }
    if (rc) return rc;
    if (sample_tiles) {
        // bounds from the sample's counts (a query with fewer than k subjects in range so far keeps its bound)
        hipLaunchKernelGGL(kth_from_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, db->stream, qs->cnt.as<uint32_t>(),
                           (uint32_t)cnt_stride, k_tight, qs->thr.as<uint32_t>(), q_begin, nq);
        // the rest of the store, once: count, tighten, append to the scratch list — in segments growing 4x, so that the
        // workgroups of one launch do not all start from the sample's bound
        len = 3u * sample_tiles;
        while (!rc && begin < n_tiles) {
            const uint32_t end = (uint32_t)std::min<uint64_t>((uint64_t)begin + len, n_tiles);
            rc = launch_tiles(db, qs, q_begin, q_end, begin, end, k_tight, thr0, d_scratch, scratch_rows, nullptr);
            begin = end;
            len = len > (1u << 28) ? len : len * 4;
        }
        if (rc) return rc;
        // every pair within the k-th distance has been counted once: the exact bounds; then the sample's own tiles with
        // those bounds fixed, appending to the same list
        hipLaunchKernelGGL(kth_from_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, db->stream, qs->cnt.as<uint32_t>(),
                           (uint32_t)cnt_stride, k_tight, qs->thr.as<uint32_t>(), q_begin, nq);
        rc = launch_tiles(db, qs, q_begin, q_end, 0, sample_tiles, 0, thr0, d_scratch, scratch_rows, nullptr, true);
        if (rc) return rc;
        // (falls through to the filter pass below: rows within the exact bounds go to the caller's list)
    } else if (count_first) {
        hipLaunchKernelGGL(kth_from_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, db->stream, qs->cnt.as<uint32_t>(),
                           (uint32_t)cnt_stride, k_tight, qs->thr.as<uint32_t>(), q_begin, nq);
        rc = launch_tiles(db, qs, q_begin, q_end, 0, n_tiles, 0, thr0, d_hits, cap, d_count, true);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(db->ev1, db->stream));
        db->timed = true;
        return SMAFA_OK;
    }
    HIP_TRY(hipEventRecord(db->ev1, db->stream));  // smafa_last_scan_ms: the scan kernels, not the filter pass below
    db->timed = true;
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), db->stream));
    hipLaunchKernelGGL(filter_rows_kernel, dim3((uint32_t)std::min<uint64_t>(1024, (scratch_rows + 255) / 256)), dim3(256), 0,
                       db->stream, d_scratch, d_ctr, (unsigned long long)scratch_rows, d_hits, (unsigned long long)cap,
                       d_count, qs->thr.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(d_ctr, 0, sizeof(unsigned long long), db->stream));  // counters are zero between scans
    return SMAFA_OK;
}

// the stream has just been synchronised after a scan_range: fold its kernel time into the call's totals
static void note_call_scan(smafa_db *db) {
    db->call_scans++;
    db->call_launches += db->last_launches;
    float ms = 0.f;
    if (db->timed && hipEventElapsedTime(&ms, db->ev0, db->ev1) == hipSuccess) db->call_ms += ms;
    db->life_ms += ms;
    db->life_launches += db->last_launches;
}

static bool hit_less(const smafa_hit &x, const smafa_hit &y) {
    if (x.query != y.query) return x.query < y.query;
    if (x.dist != y.dist) return x.dist < y.dist;
    return x.subject < y.subject;
}

// ---- ordering rows on the device: (query, dist, subject) packed into one 64-bit radix key -------------
__global__ void rows_to_keys_kernel(const smafa_hit *rows, uint64_t n, uint32_t q_begin, uint32_t dist_bits,
                                    unsigned long long *keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const smafa_hit h = rows[i];
    keys[i] = ((unsigned long long)(h.query - q_begin) << (32 + dist_bits)) | ((unsigned long long)h.dist << 32) | h.subject;
}

__global__ void keys_to_rows_kernel(const unsigned long long *keys, uint64_t n, uint32_t q_begin, uint32_t dist_bits,
                                    smafa_hit *rows) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    smafa_hit h;
    h.subject = (uint32_t)k;
    h.dist = (uint32_t)(k >> 32) & ((1u << dist_bits) - 1u);
    h.query = (uint32_t)(k >> (32 + dist_bits)) + q_begin;
    rows[i] = h;
}

// Sort db->hits[0..count) by (query, dist, subject) in place with a device radix sort; returns false (and leaves
// the rows untouched) when the key does not fit 64 bits — the caller then orders them on the host.
static int sort_rows_on_device(smafa_db *db, uint64_t count, uint32_t q_begin, uint32_t q_end, bool *sorted) {
    *sorted = false;
    uint32_t dist_bits = 1;
    while ((1u << dist_bits) <= db->L) dist_bits++;
    uint32_t q_bits = 1;
    while (q_bits < 32 && (1ull << q_bits) < (uint64_t)(q_end - q_begin)) q_bits++;
    if (dist_bits + q_bits > 32 || count > 0x7fffffffull) return SMAFA_OK;
    int rc = db->keys_a.ensure(count * sizeof(unsigned long long));
    if (!rc) rc = db->keys_b.ensure(count * sizeof(unsigned long long));
    if (rc) return rc;
    unsigned long long *ka = db->keys_a.as<unsigned long long>(), *kb = db->keys_b.as<unsigned long long>();
    const uint32_t blocks = (uint32_t)((count + 255) / 256);
    hipLaunchKernelGGL(rows_to_keys_kernel, dim3(blocks), dim3(256), 0, db->stream, db->hits.as<smafa_hit>(), count, q_begin,
                       dist_bits, ka);
    size_t tmp_bytes = 0;
    const int end_bit = (int)(32 + dist_bits + q_bits);
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, ka, kb, (int)count, 0, end_bit, db->stream));
    rc = db->sort_tmp.ensure(tmp_bytes);
    if (rc) return rc;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(db->sort_tmp.p, tmp_bytes, ka, kb, (int)count, 0, end_bit, db->stream));
    hipLaunchKernelGGL(keys_to_rows_kernel, dim3(blocks), dim3(256), 0, db->stream, kb, count, q_begin, dist_bits,
                       db->hits.as<smafa_hit>());
    HIP_TRY(hipGetLastError());
    *sorted = true;
    return SMAFA_OK;
}

// Append db->hits[0..count) — rows of queries [q_begin, q_end) — to `out`, ordered by (query, dist, subject).
static int fetch_rows(smafa_db *db, uint64_t count, uint32_t q_begin, uint32_t q_end, std::vector<smafa_hit> &out) {
    if (count == 0) return SMAFA_OK;
    bool sorted = false;
    if (count >= 4096) {  // small lists are cheaper to order on the host
        int rc = sort_rows_on_device(db, count, q_begin, q_end, &sorted);
        if (rc) return rc;
    }
    const double t0 = now_seconds();
    HIP_TRY(hipStreamSynchronize(db->stream));
    const double t1 = now_seconds();
    const size_t old = out.size();
    out.resize(old + count);
    const double t2 = now_seconds();
    HIP_TRY(hipMemcpyAsync(out.data() + old, db->hits.p, count * sizeof(smafa_hit), hipMemcpyDeviceToHost, db->stream));
    HIP_TRY(hipStreamSynchronize(db->stream));
    const double t3 = now_seconds();
    if (!sorted) std::sort(out.begin() + old, out.end(), hit_less);  // each range ordered => `out` ordered
    if (count >= (1u << 20))
        log_line(2, "%llu rows: scan + device sort %.2f ms, host buffer %.2f ms, copy back %.2f ms", (unsigned long long)count,
                 (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    return SMAFA_OK;
}

// Collect all qualifying rows of queries [q_begin, q_end) into `out` (host).  First the plain bound when there
// is one (rows within max_div are usually few); if they do not fit, tighten to the k-th smallest distance; if
// that still does not fit, halve the query range.  Ranges are visited in ascending query order and each is
// ordered by (query, dist, subject) — on the device when it is big enough to matter — so `out` is ordered.
static int collect_range(smafa_db *db, smafa_qset *qs, uint32_t q_begin, uint32_t q_end, uint32_t max_div,
                         uint32_t max_num_hits, std::vector<smafa_hit> &out) {
    const uint32_t k_tight = max_num_hits == SMAFA_NONE ? 0u : max_num_hits;
    unsigned long long count = 0;
    bool done = false;
    for (int attempt = 0; attempt < 2 && !done; attempt++) {
        uint32_t k;
        if (attempt == 0) {
            if (max_div == SMAFA_NONE && k_tight) continue;  // no bound at all: go straight to tightening
            k = 0;
        } else {
            if (!k_tight) break;
            k = k_tight;
            // counting first, the scan returns k rows per query plus ties: make room for them up front (up to 64M rows)
            if (k >= db->count_first_k && !prefilter_prunes(db, std::min<uint32_t>(max_div, db->L))) {
                const uint64_t want = std::min<uint64_t>((uint64_t)(q_end - q_begin) * ((uint64_t)k + 64u), 1ull << 26);
                if (db->hits_cap() < want) {
                    int erc = db->hits.ensure(want * sizeof(smafa_hit));
                    if (erc) return erc;
                }
            }
        }
        int rc = scan_range(db, qs, q_begin, q_end, max_div, k, db->hits.as<smafa_hit>(), db->hits_cap(),
                            db->count.as<unsigned long long>());
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(&count, db->count.p, sizeof count, hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        note_call_scan(db);
        done = count <= db->hits_cap();
    }
    if (!done) {
        // a fixed-bound scan reports its exact total: make room for it and scan once more (up to 128M rows = 1.5 GB)
        if (!k_tight && count <= (1ull << 27)) {
            int rc = db->hits.ensure(count * sizeof(smafa_hit));
            if (rc) return rc;
            return collect_range(db, qs, q_begin, q_end, max_div, max_num_hits, out);
        }
        if (q_end - q_begin > 1) {
            const uint32_t mid = q_begin + (q_end - q_begin) / 2;
            int rc = collect_range(db, qs, q_begin, mid, max_div, max_num_hits, out);
            if (rc) return rc;
            return collect_range(db, qs, mid, q_end, max_div, max_num_hits, out);
        }
        // one query with more rows than the buffer: it can have at most one row per subject
        int rc = db->hits.ensure(std::max<uint64_t>(count, db->n) * sizeof(smafa_hit));
        if (rc) return rc;
        return collect_range(db, qs, q_begin, q_end, max_div, max_num_hits, out);
    }
    return fetch_rows(db, count, q_begin, q_end, out);
}

void db_life_stats(const smafa_db *db, double *kernel_ms, uint64_t *launches) {
    *kernel_ms = db ? db->life_ms : 0.0;
    *launches = db ? db->life_launches : 0;
}

void warm_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return;
    if (hipSetDevice(device) != hipSuccess) return;
    (void)hipFree(nullptr);  // forces the context
}

// Rows past n in the last tile keep their old bits; every kernel tests subject < n_subjects before it reports or
// tightens a bound, exactly as it does for the zero padding of a fresh store.
int db_clear(smafa_db *db) {
    if (!db) return set_error(SMAFA_ERR_INVALID, "db_clear: NULL handle");
    db->n = 0;
    db->runs.clear();
    db->rows_since_sort = 0;
    db->tile_bits.clear();
    for (uint64_t &h : db->zone_hist) h = 0;
    db->generation++;
    return SMAFA_OK;
}

int scan_to_host(smafa_db *db, const uint8_t *query_codes, uint64_t n_queries, uint32_t max_div,
                 uint32_t max_num_hits, std::vector<smafa_hit> &out) {
    out.clear();
    db->call_ms = 0.f;
    db->call_launches = db->call_scans = 0;
    if (n_queries == 0) return SMAFA_OK;
    if (n_queries > 0xfffffff0ull) return set_error(SMAFA_ERR_INVALID, "too many queries in one batch");
    int rc = use_device(db);
    if (rc) return rc;
    rc = db->count.ensure(sizeof(unsigned long long));
    if (!rc && db->hits_cap() < (1ull << 22)) rc = db->hits.ensure((1ull << 22) * sizeof(smafa_hit));  // 4M rows = 48 MiB
    if (rc) return rc;
    rc = validate_codes(db, query_codes, n_queries);
    if (rc) return rc;
    rc = qset_fill(&db->scratch_q, db, query_codes, n_queries);
    if (rc) return rc;
    // k-th-distance modes whose bound is loose or absent (`smafa query` without --max-divergence — the reference's
    // default): the scan would run the all-planes kernel until each query's running bound has tightened.  Most queries
    // of real inputs have their k nearest subjects within a few mismatches, so first ask the cheap questions: a LADDER of
    // scans whose bound starts at a value the prefilters still handle well (5 of the first 32 columns: level 1 prunes,
    // zone kernel; then 12, and 30 (two planes: 16) at two words per plane: the folded bounds of the all-planes kernel still reject
    // nearly every pair there), each over the queries the step before left open.  A query with at least k rows within a step's bound is
    // finished: its k-th smallest distance is <= that bound, so every row it may print is among them.  Whoever is
    // left takes the tightening path, as a compacted batch.  Exact at every step.  A step that finishes fewer than an
    // eighth of its queries is the last one (data without near neighbours pays for one cheap step only).
    // Measured, 10 000 queries x 10M aa subjects, best hit, per step (tools/bound_probe.py, profiles/
    // r02_bound_probe.txt): bound 5 2.1 ms, 8 8.1 ms, 12 9.2 ms, 14 14.3 ms, no bound 28 ms; nucleotides 1.8, 6.6, 7.7,
    // 10.4, 16.7.
    const uint32_t k_mode = max_num_hits == SMAFA_NONE ? 0u : max_num_hits;
    const uint32_t cols = std::min<uint32_t>(32u, db->L);
    // first step: level 1 looks at `cols` columns of one plane; at a bound of a sixth of them it still rejects all but
    // a few percent of the (wave, query) steps
    std::vector<uint32_t> ladder = {(cols - 1u) / 6u, 3u * cols / 8u};
    if (db->W == 2 && db->L >= 33) {  // two words: scan_kernel's FOLD 1 / FOLD 2 forms still reject at 13..17 / 18..32
        // (a step at 16 in front of the one at 30 costs nearly as much and is redundant: queries 0..30 substitutions away
        // from their subject, 10 000 x 10M aa: 35 ms with both, profiles/r02_besthit_ladder.txt)
        // (three planes and more: a step at 17, the top of FOLD 1's range — 12 ms where the step at 30 costs 15 — is there for
        // the planner below to choose when most open queries lie within it)
        // — only where the planner will run (plan_later_steps needs 2048 open queries): a small batch would otherwise run the
        // steps at 17 AND 30 back to back, which the measurement above found redundant
        if (db->P >= 3 && n_queries >= 2048) ladder.push_back(17u);
        ladder.push_back(db->P >= 3 ? 30u : 16u);
    }
    const uint32_t limit = std::min<uint32_t>(max_div, db->L);
    // A store with a current block index answers "every pair within d" for the largest bound d its blocks serve in a few
    // microseconds per thousand queries: that fixed-bound scan goes first, in place of every tightening step at or below d
    // (a query with k rows within d is finished exactly as after any other step).
    // Rent or buy (mode 3 — what `smafa query` sets) for calls WITHOUT a usable bound too: the ladder and the loose path cost the
    // scan kernels ~1.6e-11 ms per pair and stored vector (10 000 queries x 10M aa: 16-19 ms, profiles/r04_bench_full.json); once
    // such calls have cost what an index for the ladder's first steps would, it is built — blocks as narrow as the store's size
    // leaves selective (~log2(n) - 2 bits of letters per block: 4 aa columns, 10 nucleotides at 10M subjects), so that it serves
    // the widest bounds it can (aa: up to 14).  Only where that index would take at least the ladder's first step.
    if (db->index_mode == 3 && k_mode >= 1 && n_queries > 64 && db->use_filter && db->lazy && db->two_phase && !ladder.empty() &&
        limit > ladder[0] && db->W <= (uint32_t)kIndexMaxWords && db->n >= db->index_min_rows && db->n < (1ull << 31) &&
        db->index_failed_generation != db->generation + 1u) {
        const double letter_bits = db->alphabet == SMAFA_ALPHABET_AA ? 4.3 : db->P == 2 ? 2.0 : 2.3;
        const uint32_t width = (uint32_t)std::max(2.0, std::floor((std::log2((double)db->n) - 2.0) / letter_bits));
        const uint32_t want = std::min<uint32_t>((uint32_t)kIndexMaxBlocks, db->L / width);
        if (want >= ladder[0] + 1u && (!index_current(db) || db->index.B < want)) {
            if (db->index_debt_generation != db->generation) db->index_debt_ms = 0.0, db->index_debt_generation = db->generation;
            db->index_debt_ms += (double)n_queries * (double)db->n * (double)(db->P * db->W) * 1.6e-11;
            if (db->index_debt_ms >= 0.3 + (double)want * (double)db->n * 1.0e-7) {
                if (index_build(db, want) != SMAFA_OK) {
                    log_line(1, "block index not built (%s): scanning as before", smafa_last_error());
                    index_drop(db);
                    db->index_failed_generation = db->generation + 1u;
                }
            }
        }
    }
    uint32_t index_step = UINT32_MAX;  // position in the ladder of the step the index answers
    if (k_mode >= 1 && n_queries > 64) {
        uint8_t blocks[kIndexMaxBlocks];
        uint32_t d = index_current(db) ? db->index.B : 0u;
        while (d > 0 && !index_plan(db, d - 1u, (uint32_t)n_queries, blocks)) d--;
        // (only where it replaces the ladder's first step: a lower bound than that finishes too few queries to pay for the
        // extra round of compaction — nucleotides, 10M x 100 000 queries, index served up to 2: 27 -> 45 ms; profiles/r04_index.txt)
        if (d > 0 && d - 1u < limit && !ladder.empty() && d - 1u >= ladder[0]) {
            const uint32_t served = d - 1u;
            std::vector<uint32_t> kept = {served};
            for (uint32_t b : ladder)
                if (b > served) kept.push_back(b);
            ladder.swap(kept);
            index_step = 0;
        }
    }
    std::vector<smafa_hit> done;      // rows of the finished queries (the caller's query numbers), ordered
    std::vector<uint32_t> ids;        // open queries: position in the current batch -> the caller's number (empty: same)
    std::vector<uint8_t> open_codes;  // ... and their code rows
    const uint8_t *cur = query_codes;
    uint32_t cur_n = (uint32_t)n_queries;
    smafa_qset *qs = &db->scratch_q;
    auto merge_done = [&](std::vector<smafa_hit> &more) {  // both ordered, disjoint queries
        if (done.empty()) {
            done.swap(more);
            return;
        }
        std::vector<smafa_hit> all(done.size() + more.size());
        std::merge(done.begin(), done.end(), more.begin(), more.end(), all.begin(), hit_less);
        done.swap(all);
    };
    const bool laddered = k_mode >= 1 && db->use_filter && db->lazy && db->two_phase && n_queries >= 16;
    // Which LATER steps pay is estimated once, on a sample of the queries the first step left open: every (cur_n / 256)-th
    // open query is scanned in the k-th mode at the ladder's last bound, and the distribution of their k-th distances says
    // what share of the open queries each later step would finish.  A step costs about 0.37 (bounds the OR-fold still
    // rejects at), 0.5 (FOLD 1: the filter plane's per-word sums) or 0.53 (FOLD 2: two planes) of what the loose path costs
    // per query (10M x 60 aa, 10 000 queries, best-hit launches: 9.1 / 12.5 / 13.2 ms against 25 ms for queries nothing is near to,
    // profiles/r04_bound_probe.txt; round 3's 0.3 / 0.4 / 0.5 were taken against a 29 ms loose path: with half the queries
    // unrelated the step at 12 then cost 7.4 ms to finish what the loose path does in 5.5, profiles/r04_kth.txt); the cheapest
    // sequence of steps + loose path for the rest wins.
    // (Round 2 stopped after any step that finished less than an eighth: queries 9-14 columns away from their nearest
    // subject — novel members of a family — then paid the loose path in full: 33 ms per 10 000 instead of ~17.)
    std::vector<char> run_step(ladder.size(), 1);
    bool planned = false;
    auto plan_later_steps = [&](size_t from) -> int {
        planned = true;
        size_t last = from;
        while (last + 1 < ladder.size() && ladder[last + 1] < limit) last++;
        if (from >= ladder.size() || ladder[from] >= limit) return SMAFA_OK;
        const uint32_t ns = std::min<uint32_t>(256u, cur_n / 8u);  // (a share estimated to +-3 %; ~0.7 ms at 10M subjects)
        const uint32_t stride = cur_n / ns;
        std::vector<uint8_t> sample((size_t)ns * db->L);
        for (uint32_t i = 0; i < ns; i++) memcpy(&sample[(size_t)i * db->L], cur + (size_t)i * stride * db->L, db->L);
        int prc = qset_fill(&db->scratch_q3, db, sample.data(), ns);
        if (prc) return prc;
        unsigned long long count = 0;
        // (one fixed-bound launch: every pair of the sample within the last bound; the tightening form's seed and growing
        // segments — up to a dozen small launches — cost a 256-query sample 0.9 ms where this costs 0.4)
        prc = scan_range(db, &db->scratch_q3, 0, ns, ladder[last], 0, db->hits.as<smafa_hit>(), db->hits_cap(),
                         db->count.as<unsigned long long>());
        if (prc) return prc;
        HIP_TRY(hipMemcpyAsync(&count, db->count.p, sizeof count, hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        note_call_scan(db);
        std::vector<uint32_t> kth(ns, UINT32_MAX);  // k-th smallest distance of each sample query within the last bound
        if (count > db->hits_cap()) {
            std::fill(kth.begin(), kth.end(), 0u);  // too dense to look at: every step will finish plenty
        } else if (count) {
            std::vector<smafa_hit> rows;
            prc = fetch_rows(db, count, 0, ns, rows);
            if (prc) return prc;
            for (size_t i = 0; i < rows.size();) {
                size_t j = i;
                while (j < rows.size() && rows[j].query == rows[i].query) j++;
                if (j - i >= k_mode) kth[rows[i].query] = rows[i + k_mode - 1].dist;
                i = j;
            }
        }
        const size_t n_later = last - from + 1;  // at most two steps today: every subset is tried
        double best_cost = 1.0;                   // no further step: the loose path for everybody
        uint32_t best_mask = 0;
        for (uint32_t mask = 1; mask < (1u << n_later); mask++) {
            double cost = 0.0, open_share = 1.0;
            for (size_t t = 0; t < n_later; t++) {
                if (!((mask >> t) & 1u)) continue;
                const uint32_t b = ladder[from + t];
                size_t fin = 0;
                for (uint32_t v : kth) fin += v <= b;
                cost += open_share * (b <= 3u * cols / 8u ? 0.37 : b <= 17u ? 0.5 : 0.53);
                open_share = 1.0 - (double)fin / (double)ns;
            }
            cost += open_share;
            if (cost < best_cost) {
                best_cost = cost;
                best_mask = mask;
            }
        }
        for (size_t t = from; t < ladder.size(); t++) run_step[t] = t <= last && ((best_mask >> (t - from)) & 1u);
        std::string chosen;
        for (size_t t = 0; t < n_later; t++)
            if ((best_mask >> t) & 1u) chosen += " " + std::to_string(ladder[from + t]);
        log_line(2, "near-hit plan from a sample of %u open queries: further steps at bounds [%s ] -> %.2f of the loose path's cost",
                 ns, chosen.c_str(), best_cost);
        return SMAFA_OK;
    };
    // Does the FIRST step finish anybody?  On queries unrelated to the store (or a k that wants more neighbours than a family has)
    // it costs a full zone-kernel pass and finishes nobody (2.2 ms per 10 000 queries at 10M subjects: 14 % of a batch of unrelated
    // queries, 7 % of a k = 5 call).  Asked of every (n / 256)-th query first (0.1-0.2 ms): below a sixteenth of the sample finished,
    // the step is skipped and the later steps are planned right away.  Exact either way (a skipped step only moves queries to a
    // later, looser scan).
    if (laddered && db->ladder_probe && n_queries >= 2048 && !ladder.empty() && limit > ladder[0] && index_step != 0) {
        const uint32_t ns = 256, stride = (uint32_t)(n_queries / ns);
        std::vector<uint8_t> sample((size_t)ns * db->L);
        for (uint32_t i = 0; i < ns; i++) memcpy(&sample[(size_t)i * db->L], query_codes + (size_t)i * stride * db->L, db->L);
        rc = qset_fill(&db->scratch_q3, db, sample.data(), ns);
        if (rc) return rc;
        unsigned long long count = 0;
        // (a plain fixed-bound launch — every pair within the bound — not the step's tightening form with its seed and growing
        // segments: seven small launches cost the sample more than the answer is worth)
        rc = scan_range(db, &db->scratch_q3, 0, ns, ladder[0], 0, db->hits.as<smafa_hit>(), db->hits_cap(),
                        db->count.as<unsigned long long>());
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(&count, db->count.p, sizeof count, hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        note_call_scan(db);
        uint32_t finished = ns;  // (too dense to look at: the step will finish plenty)
        if (count <= db->hits_cap()) {
            std::vector<smafa_hit> rows;
            rc = fetch_rows(db, count, 0, ns, rows);
            if (rc) return rc;
            std::vector<uint32_t> have(ns, 0);
            for (const smafa_hit &h : rows) have[h.query]++;
            finished = 0;
            for (uint32_t v : have) finished += v >= k_mode;
        }
        if (finished * 16u < ns) {
            run_step[0] = 0;
            log_line(2, "near-hit probe: %u of %u sampled queries finish at bound %u: the step is skipped", finished, ns, ladder[0]);
            rc = plan_later_steps(1);
            if (rc) return rc;
        }
    }
    for (size_t step = 0; laddered && step < ladder.size() && cur_n >= 16; step++) {
        const uint32_t bound = ladder[step];
        if (limit <= bound || (step > 0 && bound <= ladder[step - 1])) break;
        if (!run_step[step]) continue;
        unsigned long long count = 0;
        // the step itself runs in the tightening mode (bound lowered to each query's k-th distance as the scan
        // proceeds), so on dense stores only the rows within the final bound come back, not every pair within it
        // (the index's step is a fixed-bound scan: every pair within the bound, from bound + 1 probes per query)
        rc = scan_range(db, qs, 0, cur_n, bound, step == index_step ? 0u : k_mode, db->hits.as<smafa_hit>(), db->hits_cap(),
                        db->count.as<unsigned long long>());
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(&count, db->count.p, sizeof count, hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        note_call_scan(db);
        if (count > db->hits_cap()) {
            if (step == index_step) continue;  // (more rows than the buffer holds: the tightening steps take over)
            break;  // too dense to look at: the full path decides
        }
        if (count == 0) {  // nobody within this bound: do the later steps pay?
            if (!planned && cur_n >= 2048 && step + 1 < ladder.size()) {
                rc = plan_later_steps(step + 1);
                if (rc) return rc;
                continue;
            }
            break;
        }
        std::vector<smafa_hit> near;
        rc = fetch_rows(db, count, 0, cur_n, near);
        if (rc) return rc;
        std::vector<uint32_t> have(cur_n, 0);
        for (const smafa_hit &h : near) have[h.query]++;
        std::vector<uint32_t> open;  // positions in the current batch that still need an answer, ascending
        for (uint32_t q = 0; q < cur_n; q++)
            if (have[q] < k_mode) open.push_back(q);
        if (open.size() == cur_n) {
            if (!planned && cur_n >= 2048 && step + 1 < ladder.size()) {
                rc = plan_later_steps(step + 1);
                if (rc) return rc;
                continue;
            }
            break;
        }
        std::vector<smafa_hit> fin;
        fin.reserve(near.size());
        for (smafa_hit h : near)
            if (have[h.query] >= k_mode) {
                if (!ids.empty()) h.query = ids[h.query];  // ascending map: the order is kept
                fin.push_back(h);
            }
        merge_done(fin);
        log_line(2, "near-hit step at bound %u finished %u of %u queries", bound, cur_n - (uint32_t)open.size(), cur_n);
        // (the step at 30 costs half of what the loose path costs: it is taken only after a step that finished a quarter)
        const bool paid = (uint64_t)(cur_n - open.size()) * (step + 1 < ladder.size() && ladder[step + 1] >= 30u ? 4u : 8u) >= cur_n;
        std::vector<uint32_t> next_ids(open.size());
        std::vector<uint8_t> next_codes(open.size() * (size_t)db->L);
        for (size_t i = 0; i < open.size(); i++) {
            next_ids[i] = ids.empty() ? open[i] : ids[open[i]];
            memcpy(&next_codes[i * db->L], cur + (size_t)open[i] * db->L, db->L);
        }
        ids.swap(next_ids);
        open_codes.swap(next_codes);
        cur = open_codes.data();
        cur_n = (uint32_t)open.size();
        if (cur_n == 0) break;
        rc = qset_fill(&db->scratch_q2, db, cur, cur_n);
        if (rc) return rc;
        qs = &db->scratch_q2;
        if (!planned && cur_n >= 2048 && step + 1 < ladder.size()) {
            rc = plan_later_steps(step + 1);  // (the sample is drawn from the compacted batch of open queries)
            if (rc) return rc;
        } else if (!planned && !paid) {
            break;  // a small batch: round 2's rule of thumb
        }
    }
    if (cur_n > 0) {
        std::vector<smafa_hit> far;
        rc = collect_range(db, qs, 0, cur_n, max_div, max_num_hits, far);
        if (rc) return rc;
        if (!ids.empty())
            for (smafa_hit &h : far) h.query = ids[h.query];
        merge_done(far);
    }
    out.swap(done);
    if (max_num_hits != SMAFA_NONE && max_num_hits >= 1) {
        // drop rows above the k-th smallest distance of their query (the device bound only tightens)
        size_t w = 0, i = 0;
        while (i < out.size()) {
            size_t j = i;
            while (j < out.size() && out[j].query == out[i].query) j++;
            const size_t cnt = j - i;
            const uint32_t kth = cnt >= max_num_hits ? out[i + max_num_hits - 1].dist : UINT32_MAX;
            for (size_t t = i; t < j && out[t].dist <= kth; t++) out[w++] = out[t];
            i = j;
        }
        out.resize(w);
    }
    // scratch of an unusually large answer is not worth keeping (the buffers regrow on demand)
    if (db->scratch.cap > (512ull << 20)) db->scratch.release();
    if (db->keys_a.cap > (512ull << 20)) {
        db->keys_a.release();
        db->keys_b.release();
        db->sort_tmp.release();
    }
    return SMAFA_OK;
}

// A handle whose HBM image is a mapped packed store file: no decode, no pack kernel — three copies.
int db_load_packed(smafa_db **out, int device, const PackedStore &pk) {
    if (!out) return set_error(SMAFA_ERR_INVALID, "db_load_packed: out is NULL");
    int rc = smafa_db_create(out, device, (int)pk.h.alphabet, pk.h.seq_len);
    if (rc) return rc;
    smafa_db *db = *out;
    auto fail = [&](int code) {
        smafa_db_destroy(db);
        *out = nullptr;
        return code;
    };
    db->P = pk.h.planes;
    db->perm.assign(pk.perm, pk.perm + (size_t)db->W * 32);
    db->tab.assign(pk.tab, pk.tab + (size_t)db->L * 32);
    rc = db->d_perm.ensure(db->perm.size() * sizeof(uint32_t));
    if (!rc) rc = db->d_tab.ensure(db->tab.size());
    if (rc) return fail(rc);
    hipError_t e = hipMemcpyAsync(db->d_perm.p, db->perm.data(), db->perm.size() * sizeof(uint32_t), hipMemcpyHostToDevice, db->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db->d_tab.p, db->tab.data(), db->tab.size(), hipMemcpyHostToDevice, db->stream);
    db->layout_set = true;
    if (e == hipSuccess && pk.h.n > 0) {
        rc = reserve_tiles(db, pk.h.n_tiles);
        if (rc) return fail(rc);
        e = hipMemcpyAsync(db->d_planes, pk.planes, pk.h.n_tiles * db->tile_words() * sizeof(uint32_t), hipMemcpyHostToDevice, db->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(db->d_order, pk.order, pk.h.n_tiles * kWaveTile * sizeof(uint32_t), hipMemcpyHostToDevice, db->stream);
        // The zone words are recomputed from the planes, not taken from the file: a scan skips tiles on their say-so, and
        // a damaged copy would lose rows silently (the file's copy serves host-side readers).  One pass over the filter
        // plane: 20 us at 10M rows.
        std::vector<uint4> z(pk.h.n_tiles);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(zone_kernel, dim3((uint32_t)((pk.h.n_tiles + kWgWaves - 1) / kWgWaves)), dim3(256), 0, db->stream,
                               reinterpret_cast<const uint4 *>(db->d_planes), db->P, db->W, db->L, 0u, (uint32_t)pk.h.n_tiles,
                               (uint32_t)pk.h.n, db->d_zone);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(z.data(), db->d_zone, z.size() * sizeof(uint4), hipMemcpyDeviceToHost, db->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(db->stream);
        if (e == hipSuccess) note_zone_words(db, 0, z.data(), z.size());
    }
    if (e == hipSuccess) e = hipStreamSynchronize(db->stream);
    if (e != hipSuccess) return fail(set_error(SMAFA_ERR_DEVICE, "loading the packed store failed: %s", hipGetErrorString(e)));
    db->n = pk.h.n;
    for (uint64_t r = 0; r < pk.h.n_runs; r++) db->runs.push_back({pk.runs[2 * r], pk.runs[2 * r + 1] != 0});
    db->rows_since_sort = pk.h.n_runs > 1 ? pk.h.n : 0;  // a file saved from a patchwork store: sorted at the first scan
    db->generation++;
    return SMAFA_OK;
}

}  // namespace smafa

// ------------------------------------------------------------------------------------- C ABI
extern "C" {

int smafa_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} catch (...) {
    return smafa::exception_code("smafa_device_count");
}

int smafa_db_create(smafa_db **out, int device, int alphabet, uint32_t seq_len) try {
    if (!out) return set_error(SMAFA_ERR_INVALID, "smafa_db_create: out is NULL");
    *out = nullptr;
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    if (seq_len == 0) return set_error(SMAFA_ERR_PANIC, "Cannot add empty sequence to WindowSet");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return set_error(SMAFA_ERR_DEVICE, "no HIP device visible: the smafa scan engine has no CPU fallback");
    if (device < 0 || device >= ndev) return set_error(SMAFA_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    smafa_db *db = new smafa_db();
    db->device = device;
    db->alphabet = alphabet;
    db->L = seq_len;
    db->PQ = (uint32_t)query_planes_for(alphabet);
    // nucleotide stores start in the 2-bit form (A C G T only) and gain the N plane when an N is appended
    db->P = alphabet == SMAFA_ALPHABET_NT ? 2u : db->PQ;
    if (const char *pv = getenv("SMAFA_NT_PLANES")) {  // testing: force the 3-plane store
        if (alphabet == SMAFA_ALPHABET_NT && atoi(pv) == 3) db->P = 3;
    }
    db->W = (seq_len + 31) / 32;
    db->QS = (uint32_t)qrec_stride((int)db->PQ, (int)db->W);
    if (const char *fv = getenv("SMAFA_FILTER")) db->use_filter = atoi(fv) != 0;
    if (const char *tv = getenv("SMAFA_TILES")) {
        const int t = atoi(tv);
        db->tiles_override = (t == 1 || t == 2 || t == 4) ? (uint32_t)t : 0u;
    }
    if (const char *lv = getenv("SMAFA_LAZY")) db->lazy = atoi(lv) != 0;
    if (const char *pv2 = getenv("SMAFA_TWO_PHASE")) db->two_phase = atoi(pv2) != 0;
    if (const char *f3 = getenv("SMAFA_FOLD3")) db->fold3 = atoi(f3) != 0;
    if (const char *sn = getenv("SMAFA_STREAM_NT")) db->stream_nt = atoi(sn) != 0;
    if (const char *cv = getenv("SMAFA_COUNT_FIRST_K")) db->count_first_k = (uint32_t)std::max(2, atoi(cv));
    if (const char *lp = getenv("SMAFA_LADDER_PROBE")) db->ladder_probe = atoi(lp) != 0;
    if (const char *zd = getenv("SMAFA_ZONE_DIRECT")) db->zone_direct = atoi(zd) != 0;
    if (const char *lf = getenv("SMAFA_LAZY_FOLD")) db->lazy_fold = atoi(lf) != 0;
    if (const char *ks = getenv("SMAFA_KTH_HIST_SEED")) db->kth_hist_seed = atoi(ks) != 0;
    if (const char *ks = getenv("SMAFA_KTH_SAMPLE")) db->kth_sample_div = (uint32_t)std::max(0, atoi(ks));
    if (const char *ks = getenv("SMAFA_KTH_SAMPLE_MIN_TILES")) db->kth_sample_min_tiles = (uint32_t)std::max(1, atoi(ks));
    if (const char *ov = getenv("SMAFA_WIDE_ONE")) db->wide_one = atoi(ov) != 0;
    if (const char *wv = getenv("SMAFA_WIDE_FROM")) db->wide_from = (uint32_t)std::max(3, atoi(wv));
    if (const char *zv = getenv("SMAFA_ZONE")) db->zone = std::min(2, std::max(0, atoi(zv)));
    if (const char *sv = getenv("SMAFA_SORT")) db->sort_rows = atoi(sv) != 0;
    if (const char *sv = getenv("SMAFA_RESORT")) db->resort = atoi(sv) != 0;
    if (const char *sv = getenv("SMAFA_PRUNE_P")) db->prune_p = atof(sv);
    if (const char *sv = getenv("SMAFA_RESORT_MIN")) db->resort_min = std::max<uint64_t>(2, strtoull(sv, nullptr, 10));
    if (const char *zl = getenv("SMAFA_ZONE_LOOSE")) db->zone_loose = atof(zl);
    if (const char *iv = getenv("SMAFA_INDEX")) db->index_mode = std::min(3, std::max(0, atoi(iv)));
    if (const char *iv = getenv("SMAFA_INDEX_MAX_RUN")) db->index_max_run = std::max<uint64_t>(1, strtoull(iv, nullptr, 10));
    if (const char *iv = getenv("SMAFA_INDEX_CAND")) db->index_cand_per_subject = atof(iv);
    if (const char *iv = getenv("SMAFA_INDEX_MIN_ROWS")) db->index_min_rows = std::max<uint64_t>(1, strtoull(iv, nullptr, 10));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) db->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&db->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&db->ev0);
    if (e == hipSuccess) e = hipEventCreate(&db->ev1);
    if (e == hipSuccess && db->ctrs.ensure(kCtrBytes) != SMAFA_OK) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemset(db->ctrs.p, 0, kCtrBytes);  // the scan kernels leave their counters at zero
    if (e != hipSuccess) {
        smafa_db_destroy(db);
        return set_error(SMAFA_ERR_DEVICE, "stream/event/counter creation failed: %s", hipGetErrorString(e));
    }
    db->stream = db->own_stream;
    *out = db;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_db_create");
}

int smafa_db_append(smafa_db *db, const uint8_t *codes, uint64_t n) try {
    if (!db || (!codes && n)) return set_error(SMAFA_ERR_INVALID, "smafa_db_append: NULL argument");
    if (n == 0) return SMAFA_OK;
    if (db->n + n > 0xffffff00ull) return set_error(SMAFA_ERR_INVALID, "subject store limited to 2^32 rows");
    int rc = use_device(db);
    if (rc) return rc;
    uint8_t max_code = 0;
    rc = validate_codes(db, codes, n, &max_code);
    if (rc) return rc;
    if (db->alphabet == SMAFA_ALPHABET_NT && max_code >= 4 && db->P < 3) {  // first N: leave the 2-bit form
        rc = upgrade_planes(db, 3);
        if (rc) return rc;
    }
    rc = reserve_tiles(db, (db->n + n + kWaveTile - 1) / kWaveTile);
    if (rc) return rc;
    rc = pack_rows(db, codes, db->n, n, db->d_planes, 0);
    if (rc) return rc;
    db->n += n;
    db->rows_since_sort += n;
    // a store that is still ONE sorted run (the bulk load itself) has nothing to gain from a re-sort: the quarter rule
    // counts growth since the store was last in one sorted run
    if (db->runs.size() == 1 && db->runs[0].sorted) db->rows_since_sort = 0;
    db->generation++;
    if (n > (1u << 20)) {  // a bulk load's staging and sort buffers are not worth keeping
        for (DevBuf *b : {&db->upload, &db->keys_a, &db->keys_b, &db->idx_a, &db->idx_b, &db->sort_tmp}) b->release();
    }
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_db_append");
}

int smafa_db_save(smafa_db *db, const char *path) try {
    if (!db || !path) return set_error(SMAFA_ERR_INVALID, "smafa_db_save: NULL argument");
    int rc = use_device(db);
    if (rc) return rc;
    if (!db->layout_set) {  // an empty store has no layout yet: give it the default one
        rc = choose_layout(db, nullptr, 0);
        if (rc) return rc;
    }
    rc = maybe_resort(db);  // a patchwork of appends is saved as one sorted run
    if (rc) return rc;
    PackedHeader h{};
    h.alphabet = (uint32_t)db->alphabet;
    h.seq_len = db->L;
    h.planes = db->P;
    h.words = db->W;
    h.n = db->n;
    h.n_tiles = (db->n + kWaveTile - 1) / kWaveTile;
    h.n_runs = db->runs.size();
    std::vector<uint32_t> planes(h.n_tiles * db->tile_words()), order(h.n_tiles * kWaveTile);
    std::vector<uint4> zone(h.n_tiles);
    if (h.n_tiles) {
        HIP_TRY(hipMemcpyAsync(planes.data(), db->d_planes, planes.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipMemcpyAsync(order.data(), db->d_order, order.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipMemcpyAsync(zone.data(), db->d_zone, zone.size() * sizeof(uint4), hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
    }
    std::vector<uint64_t> runs;
    for (const smafa_db::Run &r : db->runs) {
        runs.push_back(r.rows);
        runs.push_back(r.sorted ? 1u : 0u);
    }
    return write_packed_file(path, h, db->perm.data(), db->tab.data(), runs.data(), order.data(), zone.data(), planes.data());
} catch (...) {
    return smafa::exception_code("smafa_db_save");
}

int smafa_db_load(smafa_db **out, int device, const char *path) try {
    if (!out || !path) return set_error(SMAFA_ERR_INVALID, "smafa_db_load: NULL argument");
    *out = nullptr;
    PackedStore pk;
    int rc = pk.open(path);
    if (rc) return rc;
    return db_load_packed(out, device, pk);
} catch (...) {
    return smafa::exception_code("smafa_db_load");
}

int smafa_db_info(const smafa_db *db, smafa_db_info_t *info) try {
    if (!db || !info) return set_error(SMAFA_ERR_INVALID, "smafa_db_info: NULL argument");
    info->n_subjects = db->n;
    info->seq_len = db->L;
    info->alphabet = db->alphabet;
    info->device = db->device;
    info->planes = db->P;
    info->words_per_plane = db->W;
    info->bytes_per_subject = (uint64_t)db->P * db->W * 4;
    info->hbm_bytes = (db->n + kWaveTile - 1) / kWaveTile * db->tile_words() * sizeof(uint32_t);
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_db_info");
}

int smafa_db_set_stream(smafa_db *db, void *hip_stream) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_db_set_stream: NULL handle");
    db->stream = hip_stream ? (hipStream_t)hip_stream : db->own_stream;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_db_set_stream");
}

void smafa_db_destroy(smafa_db *db) {
    if (!db) return;
    (void)hipSetDevice(db->device);
    if (db->d_planes) (void)hipFree(db->d_planes);
    if (db->d_order) (void)hipFree(db->d_order);
    if (db->d_zone) (void)hipFree(db->d_zone);
    for (DevBuf *b : {&db->upload, &db->hits, &db->count, &db->scratch, &db->ctrs, &db->keys_a, &db->keys_b, &db->sort_tmp,
                      &db->idx_a, &db->idx_b, &db->d_perm, &db->d_tab, &db->scratch_q.qrec,
                      &db->scratch_q.thr, &db->scratch_q.cnt, &db->scratch_q2.qrec, &db->scratch_q2.thr, &db->scratch_q2.cnt,
                      &db->scratch_q3.qrec, &db->scratch_q3.thr, &db->scratch_q3.cnt, &db->index.kp, &db->index.dir,
                      &db->index.stats, &db->index.rows})
        b->release();
    if (db->each_graph) (void)hipGraphExecDestroy(db->each_graph);
    if (db->ev0) (void)hipEventDestroy(db->ev0);
    if (db->ev1) (void)hipEventDestroy(db->ev1);
    if (db->own_stream) (void)hipStreamDestroy(db->own_stream);
    delete db;
}

int smafa_set_query_block(smafa_db *db, uint32_t queries_per_block) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_set_query_block: NULL handle");
    db->qb_override = queries_per_block;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_set_query_block");
}

int smafa_last_scan_plan(smafa_db *db, uint32_t *filter_plane_resident, uint32_t *tiles_per_wave, uint32_t *query_blocks) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_last_scan_plan: NULL handle");
    if (filter_plane_resident) *filter_plane_resident = db->plan_lazy;
    if (tiles_per_wave) *tiles_per_wave = db->plan_tiles;
    if (query_blocks) *query_blocks = db->plan_qblocks;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_last_scan_plan");
}

int smafa_last_scan_kernel(smafa_db *db, char *name, uint64_t cap) try {
    if (!db || !name || cap == 0) return set_error(SMAFA_ERR_INVALID, "smafa_last_scan_kernel: NULL argument");
    snprintf(name, (size_t)cap, "%s", db->plan_kernel);
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_last_scan_kernel");
}

int smafa_hbm_read_probe(int device, uint64_t bytes, double *gb_per_s) try {
    if (!gb_per_s) return set_error(SMAFA_ERR_INVALID, "smafa_hbm_read_probe: NULL argument");
    *gb_per_s = 0.0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return set_error(SMAFA_ERR_DEVICE, "no HIP device visible: the smafa scan engine has no CPU fallback");
    if (device < 0 || device >= ndev) return set_error(SMAFA_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    bytes = std::max<uint64_t>(bytes, 1ull << 20) / 16 * 16;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    uint4 *d = nullptr;
    uint32_t *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&d, bytes);
    if (e == hipSuccess) e = hipMalloc(&out, 4);
    if (e == hipSuccess) e = hipMemset(d, 1, bytes);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best = 0.f;
    for (int shape = 0; shape < 4; shape++) {  // grid-stride at 8 and 32 workgroups per CU, then contiguous spans + nt at 8 and 16
        const int grid = prop.multiProcessorCount * (shape == 0 ? 8 : shape == 1 ? 32 : shape == 2 ? 8 : 16);
        for (int rep = 0; rep < 4 && e == hipSuccess; rep++) {  // the first repetition of each shape warms up
            e = hipEventRecord(e0, nullptr);
            if (shape < 2)
                hipLaunchKernelGGL(hbm_read_probe_kernel, dim3(grid), dim3(256), 0, nullptr, d, (size_t)(bytes / 16), out);
            else
                hipLaunchKernelGGL(hbm_read_probe_span_kernel, dim3(grid), dim3(256), 0, nullptr, d, (size_t)(bytes / 16), out);
            if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e == hipSuccess && rep > 0 && ms > 0.f && (best == 0.f || ms < best)) best = ms;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d) (void)hipFree(d);
    if (out) (void)hipFree(out);
    if (e != hipSuccess) return set_error(SMAFA_ERR_DEVICE, "HBM read probe failed: %s", hipGetErrorString(e));
    if (best > 0.f) *gb_per_s = (double)bytes / ((double)best * 1e-3) / 1e9;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_hbm_read_probe");
}

const char *smafa_build_id(void) {
#ifdef SMAFA_BUILD_ID
    return SMAFA_BUILD_ID;
#else
    return "unknown";
#endif
}

int smafa_set_zone_level(smafa_db *db, int mode) try {
    if (!db || mode < 0 || mode > 2) return set_error(SMAFA_ERR_INVALID, "smafa_set_zone_level: bad argument");
    db->zone = mode;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_set_zone_level");
}

int smafa_db_build_index(smafa_db *db, uint32_t max_div) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_db_build_index: NULL handle");
    if (max_div >= (uint32_t)kIndexMaxBlocks) return set_error(SMAFA_ERR_INVALID, "smafa_db_build_index: bounds up to %d", kIndexMaxBlocks - 1);
    return index_build(db, max_div + 1u);
} catch (...) {
    return smafa::exception_code("smafa_db_build_index");
}

int smafa_db_drop_index(smafa_db *db) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_db_drop_index: NULL handle");
    int rc = use_device(db);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(db->stream));
    index_drop(db);
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_db_drop_index");
}

int smafa_set_index(smafa_db *db, int mode) try {
    if (!db || mode < 0 || mode > 3) return set_error(SMAFA_ERR_INVALID, "smafa_set_index: bad argument");
    db->index_mode = mode;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_set_index");
}

int smafa_index_info(const smafa_db *db, smafa_index_info_t *info) try {
    if (!db || !info) return set_error(SMAFA_ERR_INVALID, "smafa_index_info: NULL argument");
    memset(info, 0, sizeof *info);
    const auto &ix = db->index;
    info->mode = db->index_mode;
    info->probe_launches = db->index_probes;
    if (!index_current(db)) return SMAFA_OK;
    info->current = 1;
    info->blocks = ix.B;
    info->bytes = ix.kp.cap + ix.dir.cap + ix.rows.cap;
    info->build_ms = ix.build_ms;
    std::vector<double> runs;
    for (uint32_t b = 0; b < ix.B; b++) {
        if (ix.max_run[b] > info->longest_run) info->longest_run = ix.max_run[b];
        if (ix.max_run[b] <= db->index_max_run) runs.push_back(ix.mean_run[b]);
    }
    std::sort(runs.begin(), runs.end());
    info->usable_blocks = (uint32_t)runs.size();
    // the largest bound the index would answer for a big batch, and the candidates per query expected at it
    // (not every bound below it need be: the limit on candidates is wider where the scan kernels are in a slower form)
    double sum = 0.0;
    info->max_div_served = SMAFA_NONE;
    for (size_t j = 0; j < runs.size(); j++) {
        sum += runs[j];
        if (sum > index_cand_limit(db, (uint32_t)j)) continue;
        info->max_div_served = (uint32_t)j;
        info->candidates_per_query = sum;
    }
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_index_info");
}

int smafa_set_prefilter(smafa_db *db, int enabled) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_set_prefilter: NULL handle");
    db->use_filter = enabled != 0;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_set_prefilter");
}

int smafa_qset_create(smafa_qset **out, smafa_db *db, const uint8_t *query_codes, uint64_t n_queries) try {
    if (!out || !db || (!query_codes && n_queries)) return set_error(SMAFA_ERR_INVALID, "smafa_qset_create: NULL argument");
    *out = nullptr;
    if (n_queries > 0xfffffff0ull) return set_error(SMAFA_ERR_INVALID, "too many queries in one set");
    int rc = use_device(db);
    if (rc) return rc;
    rc = validate_codes(db, query_codes, n_queries);
    if (rc) return rc;
    smafa_qset *qs = new smafa_qset();
    rc = qset_fill(qs, db, query_codes, n_queries);
    if (rc) {
        smafa_qset_destroy(qs);
        return rc;
    }
    *out = qs;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_qset_create");
}

void smafa_qset_destroy(smafa_qset *qs) {
    if (!qs) return;
    if (qs->db) (void)hipSetDevice(qs->db->device);
    if (qs->db && qs->db->each_graph && qs->db->each_key.qs == qs) {  // the captured passes read this set's records
        (void)hipGraphExecDestroy(qs->db->each_graph);
        qs->db->each_graph = nullptr;
    }
    qs->qrec.release();
    qs->thr.release();
    qs->cnt.release();
    delete qs;
}

int smafa_scan_launch(smafa_db *db, smafa_qset *qs, uint32_t max_div, uint32_t max_num_hits, void *d_hits,
                      uint64_t cap, void *d_count) try {
    if (!db || !qs || !d_count || (!d_hits && cap)) return set_error(SMAFA_ERR_INVALID, "smafa_scan_launch: NULL argument");
    if (qs->db != db) return set_error(SMAFA_ERR_INVALID, "query set was packed for a different store");
    if (max_num_hits == 0) max_num_hits = SMAFA_NONE;
    int rc = use_device(db);
    if (rc) return rc;
    // cap = 0 with no buffer: the caller only wants the count — the kernels still need a non-NULL list to know that rows
    // are wanted (nothing is ever written to it at capacity 0)
    if (!d_hits) d_hits = db->ctrs.p;
    return scan_range(db, qs, 0, (uint32_t)qs->nq, max_div, max_num_hits == SMAFA_NONE ? 0u : max_num_hits,
                      (smafa_hit *)d_hits, cap, (unsigned long long *)d_count);
} catch (...) {
    return smafa::exception_code("smafa_scan_launch");
}

// One pass over the store PER QUERY (north_star's literal "each query is broadcast against all subjects"), the passes
// enqueued back to back by this one call: query i's rows go to d_hits + i * cap_per_query, its exact count to d_counts[i].
// Each pass is the complete one-launch fixed-bound scan of smafa_scan_launch for a one-query set (same kernel choice: a
// sorted store takes scan_zone_few_kernel unless the zone level is off, then the pass streams the prefilter's plane).
// use_graph: the passes are captured once as a HIP graph and replayed while the arguments stay the same — no launch
// gap on the host side at all.
int smafa_scan_each(smafa_db *db, smafa_qset *qs, uint32_t max_div, void *d_hits, uint64_t cap_per_query, void *d_counts,
                    int use_graph) try {
    if (!db || !qs || !d_counts || (!d_hits && cap_per_query)) return set_error(SMAFA_ERR_INVALID, "smafa_scan_each: NULL argument");
    if (qs->db != db) return set_error(SMAFA_ERR_INVALID, "query set was packed for a different store");
    int rc = use_device(db);
    if (rc) return rc;
    db->last_launches = 0;
    db->timed = false;
    const uint32_t nq = (uint32_t)qs->nq;
    if (nq == 0) return SMAFA_OK;
    if (db->n == 0) {
        HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nq * sizeof(unsigned long long), db->stream));
        return SMAFA_OK;
    }
    rc = maybe_resort(db);
    if (rc) return rc;
    if (!d_hits) d_hits = db->ctrs.p;  // count-only: see smafa_scan_launch
    const uint32_t thr0 = std::min<uint32_t>(max_div, db->L);
    const uint32_t n_tiles = (uint32_t)((db->n + kWaveTile - 1) / kWaveTile);
    auto enqueue = [&]() -> int {
        // every pass reserves its rows from its own counter, d_counts[q], zeroed here once for all of them: the count a pass
        // leaves there is its result, and no workgroup has to find out whether it was the last one
        // (a kernel, not hipMemsetAsync: the memset node of a captured graph left garbage in the counters when the graph was
        // replayed — ROCm 7.2; kernel nodes replay as captured)
        hipLaunchKernelGGL(fill_u32_kernel, dim3((2 * nq + 255) / 256), dim3(256), 0, db->stream, (uint32_t *)d_counts, 0u,
                           (uint64_t)2 * nq);
        for (uint32_t q = 0; q < nq; q++) {
            int r = launch_tiles(db, qs, q, q + 1, 0, n_tiles, 0, thr0, (smafa_hit *)d_hits + (size_t)q * cap_per_query,
                                 cap_per_query, nullptr, false, (unsigned long long *)d_counts + q);
            if (r) return r;
        }
        return SMAFA_OK;
    };
    if (!use_graph) {
        HIP_TRY(hipEventRecord(db->ev0, db->stream));
        rc = enqueue();
        if (rc) return rc;
        HIP_TRY(hipEventRecord(db->ev1, db->stream));
        db->timed = true;
        return SMAFA_OK;
    }
    smafa_db::EachKey key;
    memset(&key, 0, sizeof key);  // (padding bytes take part in the comparison below)
    key.qs = qs, key.qrec = qs->qrec.p, key.qs_serial = qs->serial;
    key.hits = d_hits, key.counts = d_counts, key.cap = cap_per_query, key.nq = qs->nq, key.generation = db->generation;
    key.max_div = max_div, key.qb = db->qb_override, key.zone = db->zone, key.filter = db->use_filter;
    if (!db->each_graph || memcmp(&key, &db->each_key, sizeof key) != 0) {
        if (db->each_graph) (void)hipGraphExecDestroy(db->each_graph);
        db->each_graph = nullptr;
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(db->stream, hipStreamCaptureModeThreadLocal));
        rc = enqueue();
        hipError_t e = hipStreamEndCapture(db->stream, &graph);
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e != hipSuccess) return set_error(SMAFA_ERR_DEVICE, "graph capture failed: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&db->each_graph, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
            db->each_graph = nullptr;
            return set_error(SMAFA_ERR_DEVICE, "graph instantiation failed: %s", hipGetErrorString(e));
        }
        memcpy(&db->each_key, &key, sizeof key);
    }
    db->last_launches = nq;
    HIP_TRY(hipEventRecord(db->ev0, db->stream));
    HIP_TRY(hipGraphLaunch(db->each_graph, db->stream));
    HIP_TRY(hipEventRecord(db->ev1, db->stream));
    db->timed = true;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_scan_each");
}

int smafa_last_call_stats(smafa_db *db, float *kernel_ms, uint32_t *n_launches, uint32_t *n_scans) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_last_call_stats: NULL handle");
    if (kernel_ms) *kernel_ms = db->call_ms;
    if (n_launches) *n_launches = db->call_launches;
    if (n_scans) *n_scans = db->call_scans;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_last_call_stats");
}

int smafa_launch_device(smafa_db *db, int *device_at_last_launch, uint64_t *launches_off_device) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_launch_device: NULL handle");
    if (device_at_last_launch) *device_at_last_launch = db->launch_device;
    if (launches_off_device) *launches_off_device = db->launches_off_device;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_launch_device");
}

int smafa_sync(smafa_db *db) try {
    if (!db) return set_error(SMAFA_ERR_INVALID, "smafa_sync: NULL handle");
    HIP_TRY(hipStreamSynchronize(db->stream));
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_sync");
}

int smafa_last_scan_ms(smafa_db *db, float *ms, uint32_t *n_launches) try {
    if (!db || !ms) return set_error(SMAFA_ERR_INVALID, "smafa_last_scan_ms: NULL argument");
    *ms = 0.f;
    if (n_launches) *n_launches = db->last_launches;
    if (!db->timed) return SMAFA_OK;
    HIP_TRY(hipEventSynchronize(db->ev1));
    HIP_TRY(hipEventElapsedTime(ms, db->ev0, db->ev1));
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_last_scan_ms");
}

int smafa_scan_hits(smafa_db *db, const uint8_t *query_codes, uint64_t n_queries, uint32_t max_div,
                    uint32_t max_num_hits, smafa_hit *out, uint64_t cap, uint64_t *n_out) try {
    if (!db || !n_out || (!query_codes && n_queries) || (!out && cap))
        return set_error(SMAFA_ERR_INVALID, "smafa_scan_hits: NULL argument");
    if (max_num_hits == 0) max_num_hits = SMAFA_NONE;
    // fingerprint of the request: the query bytes, their count, the bounds and the state of the store
    uint64_t key = 0x9e3779b97f4a7c15ull ^ db->generation;
    {
        const size_t bytes = (size_t)n_queries * db->L;
        size_t i = 0;
        for (; i + 8 <= bytes; i += 8) {
            uint64_t v;
            memcpy(&v, query_codes + i, 8);
            key = (key ^ v) * 0xff51afd7ed558ccdull;
            key ^= key >> 32;
        }
        for (; i < bytes; i++) key = (key ^ query_codes[i]) * 0x100000001b3ull;
    }
    std::vector<smafa_hit> rows;
    if (db->retry_valid && db->retry_key == key && db->retry_nq == n_queries && db->retry_div == max_div &&
        db->retry_k == max_num_hits && db->retry_codes.size() == (size_t)n_queries * db->L &&
        memcmp(db->retry_codes.data(), query_codes, db->retry_codes.size()) == 0) {
        rows.swap(db->retry_rows);  // the retry after SMAFA_ERR_CAPACITY: same request, same store
    } else {
        int rc = scan_to_host(db, query_codes, n_queries, max_div, max_num_hits, rows);
        if (rc) return rc;
    }
    db->retry_valid = false;
    std::vector<smafa_hit>().swap(db->retry_rows);
    *n_out = rows.size();
    if (rows.size() <= cap) std::vector<uint8_t>().swap(db->retry_codes);
    if (rows.size() > cap) {
        const size_t need = rows.size();
        db->retry_rows.swap(rows);
        db->retry_codes.assign(query_codes, query_codes + (size_t)n_queries * db->L);
        db->retry_key = key;
        db->retry_nq = n_queries;
        db->retry_div = max_div;
        db->retry_k = max_num_hits;
        db->retry_valid = true;
        return set_error(SMAFA_ERR_CAPACITY, "hit buffer too small: %zu rows needed, capacity %llu", need,
                         (unsigned long long)cap);
    }
    if (!rows.empty()) memcpy(out, rows.data(), rows.size() * sizeof(smafa_hit));
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_scan_hits");
}

int smafa_distances(smafa_db *db, const uint8_t *query_codes, uint32_t *distances) try {
    if (!db || !query_codes || (!distances && db->n)) return set_error(SMAFA_ERR_INVALID, "smafa_distances: NULL argument");
    if (db->n == 0) return SMAFA_OK;
    int rc = use_device(db);
    if (rc) return rc;
    rc = validate_codes(db, query_codes, 1);
    if (rc) return rc;
    rc = qset_fill(&db->scratch_q, db, query_codes, 1);
    if (rc) return rc;
    const uint32_t n_tiles = (uint32_t)((db->n + kWaveTile - 1) / kWaveTile);
    rc = db->keys_a.ensure((size_t)n_tiles * kWaveTile * sizeof(uint32_t));  // any scratch buffer will do
    if (rc) return rc;
    hipLaunchKernelGGL(distances_kernel, dim3((n_tiles + kWgWaves - 1) / kWgWaves), dim3(256), 0, db->stream,
                       reinterpret_cast<const uint4 *>(db->d_planes), n_tiles, (uint32_t)db->n, db->P, db->PQ, db->W,
                       db->scratch_q.qrec.as<uint32_t>(), db->d_order, db->keys_a.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(distances, db->keys_a.p, db->n * sizeof(uint32_t), hipMemcpyDeviceToHost, db->stream));
    HIP_TRY(hipStreamSynchronize(db->stream));
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_distances");
}

}  // extern "C"
