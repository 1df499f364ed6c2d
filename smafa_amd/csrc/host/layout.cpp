// layout.cpp — the layout of a packed store, and the store packed on the HOST.
//
// compute_layout: which source column sits in which packed column and how each column's codes are re-coded (see the
// comment above choose_layout in engine.hip and DESIGN.md §2).  Shared by the device path (engine.hip uploads the tables
// and pack_rows_kernel applies them) and by pack_store_on_host below.
//
// pack_store_on_host: everything smafa_db_append does on the GPU for a store's first, single append — layout, sort by
// filter words (stable, same key), bit-plane tiles, order, zone words — restated with plain loops, so that
// `smafa makedb --packed` also works on a machine without a GPU, and so that the device kernels have an independent
// check: for the same rows the file written here is byte-identical to smafa_db_save's (tests/test_gpu_layout.py).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "packed.h"

namespace smafa {

void compute_layout(int alphabet, uint32_t L, const uint8_t *codes, uint64_t n, std::vector<uint32_t> &perm,
                    std::vector<uint8_t> &tab) {
    const uint32_t W = (L + 31) / 32;
    const bool aa = alphabet == SMAFA_ALPHABET_AA;
    const uint32_t n_sym = aa ? 28u : 4u, side_cap = aa ? 16u : 2u;
    std::vector<uint32_t> cnt((size_t)L * 32, 0);
    // SMAFA_LAYOUT=0 (A/B runs, tools/layout_check.py): no statistics — columns in file order, the default code split
    const char *lv = getenv("SMAFA_LAYOUT");
    const uint64_t S = (lv && atoi(lv) == 0) ? 0 : std::min<uint64_t>(n, 4096);
    for (uint64_t k = 0; k < S; k++) {
        const uint8_t *row = codes + (size_t)(k * n / S) * L;
        for (uint32_t c = 0; c < L; c++) cnt[(size_t)c * 32 + (row[c] & 31u)]++;
    }
    tab.assign((size_t)L * 32, 0);
    std::vector<double> score(L, 0.0);
    for (uint32_t c = 0; c < L; c++) {
        const uint32_t *cc = &cnt[(size_t)c * 32];
        uint8_t *tc = &tab[(size_t)c * 32];
        for (uint32_t v = 0; v < 32; v++) tc[v] = (uint8_t)v;  // codes outside the movable set keep their value
        uint32_t side_of[32];
        uint64_t tot[2] = {0, 0};
        auto balance = [&]() {
            const double all = (double)(tot[0] + tot[1]);
            return all > 0 ? 2.0 * (double)tot[0] * (double)tot[1] / (all * all) : 0.0;
        };
        if (!aa) {
            // three pairings of A C G T (codes 0..3); side 1 listed
            static const uint8_t pair[3][2] = {{2, 3}, {1, 3}, {1, 2}};  // {A,C}|{G,T}  {A,G}|{C,T}  {A,T}|{C,G}
            double sc[3];
            for (int k = 0; k < 3; k++) {
                tot[1] = (uint64_t)cc[pair[k][0]] + cc[pair[k][1]];
                tot[0] = (uint64_t)cc[0] + cc[1] + cc[2] + cc[3] - tot[1];
                sc[k] = balance();
            }
            int best = 0;
            const double top = std::max(sc[0], std::max(sc[1], sc[2]));
            if (sc[0] < 0.95 * top) best = sc[1] >= sc[2] ? 1 : 2;
            for (uint32_t v = 0; v < 4; v++) side_of[v] = (v == pair[best][0] || v == pair[best][1]) ? 1u : 0u;
            score[c] = sc[best];
        } else {
            uint32_t idx[32], num[2] = {0, 0};
            for (uint32_t v = 0; v < n_sym; v++) idx[v] = v;
            std::stable_sort(idx, idx + n_sym, [&](uint32_t x, uint32_t y) { return cc[x] > cc[y]; });
            for (uint32_t k = 0; k < n_sym; k++) {
                uint32_t side = tot[0] <= tot[1] ? 0u : 1u;
                if (num[side] == side_cap) side ^= 1u;
                side_of[idx[k]] = side;
                tot[side] += cc[idx[k]];
                num[side]++;
            }
            score[c] = balance();
        }
        // even codes for side 0, odd for side 1.  Amino acids: within a side the most frequent letters get the lowest
        // codes, so the HIGHEST plane is set only for the rarest letters (20 letters: four of them) — the bound that leaves
        // the last plane out (scan_kernel FOLD 3) then misses a mismatch only between those and the four commonest.
        // Nucleotides: in code order (A C G T among 0..3).
        uint32_t next[2] = {0, 1};
        bool used[32] = {false};
        uint32_t order_v[32];
        for (uint32_t v = 0; v < n_sym; v++) order_v[v] = v;
        if (aa) std::stable_sort(order_v, order_v + n_sym, [&](uint32_t x, uint32_t y) { return cc[x] > cc[y]; });
        for (uint32_t k = 0; k < n_sym; k++) {
            const uint32_t v = order_v[k];
            tc[v] = (uint8_t)next[side_of[v]];
            used[next[side_of[v]]] = true;
            next[side_of[v]] += 2;
        }
        if (aa) {  // codes 28..31 never occur; keep the map a permutation anyway
            uint32_t free_v = 0;
            for (uint32_t v = n_sym; v < 32; v++) {
                while (used[free_v]) free_v++;
                tc[v] = (uint8_t)free_v;
                used[free_v] = true;
            }
        }
    }
    std::vector<uint32_t> cols(L);
    for (uint32_t c = 0; c < L; c++) cols[c] = c;
    std::stable_sort(cols.begin(), cols.end(), [&](uint32_t x, uint32_t y) { return score[x] > score[y]; });
    perm.assign((size_t)W * 32, 0);
    for (uint32_t j = 0; j < L; j++) perm[j] = cols[j];
}

static inline uint32_t brev32(uint32_t x) {
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    return x;
}

static inline uint32_t gray_rank_host(uint32_t x) {  // row_keys_kernel's gray_rank
    x = brev32(x);
    x ^= x >> 1;
    x ^= x >> 2;
    x ^= x >> 4;
    x ^= x >> 8;
    x ^= x >> 16;
    return x;
}

int pack_store_on_host(int alphabet, uint32_t L, const uint8_t *codes, uint64_t n, const char *path) {
    if (!codes || !path || n == 0 || L == 0) return set_error(SMAFA_ERR_INVALID, "pack_store_on_host: bad argument");
    if (n > 0xffffff00ull) return set_error(SMAFA_ERR_INVALID, "subject store limited to 2^32 rows");
    const uint32_t W = (L + 31) / 32;
    const uint32_t lim = alphabet == SMAFA_ALPHABET_AA ? 28u : 5u;
    const unsigned T = n >= (1u << 16) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    auto parallel = [&](auto &&fn) {
        if (T == 1) return fn(0u);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; t++) pool.emplace_back(fn, t);
        for (auto &th : pool) th.join();
    };
    std::vector<uint8_t> worst(T, 0);
    parallel([&](unsigned t) {
        uint8_t w = 0;
        for (size_t i = (size_t)n * L * t / T, e = (size_t)n * L * (t + 1) / T; i < e; i++) w = codes[i] > w ? codes[i] : w;
        worst[t] = w;
    });
    const uint8_t max_code = *std::max_element(worst.begin(), worst.end());
    if (max_code >= lim) return set_error(SMAFA_ERR_INVALID, "code byte %u outside the alphabet (max %u)", max_code, lim - 1);
    const uint32_t P = alphabet == SMAFA_ALPHABET_AA ? 5u : (max_code >= 4 ? 3u : 2u);  // nucleotides: 2 planes while there is no N
    std::vector<uint32_t> perm;
    std::vector<uint8_t> tab;
    compute_layout(alphabet, L, codes, n, perm, tab);
    // ---- sort key per row (row_keys_kernel), stable sort (the device's radix sort is stable)
    const bool sorted = n >= 4096 && n < (1ull << 31);  // kSortMin and the sort limit of engine.hip
    std::vector<uint32_t> src(n);
    for (uint64_t i = 0; i < n; i++) src[i] = (uint32_t)i;
    if (sorted) {
        std::vector<uint64_t> keys(n);
        parallel([&](unsigned t) {
            const uint32_t c0 = std::min(L, 32u), c1 = std::min(L, 64u);
            for (uint64_t i = n * t / T, e = n * (t + 1) / T; i < e; i++) {
                const uint8_t *row = codes + (size_t)i * L;
                uint32_t w0 = 0, w1 = 0;
                for (uint32_t j = 0; j < c0; j++) w0 |= (uint32_t)(tab[(size_t)perm[j] * 32 + (row[perm[j]] & 31u)] & 1u) << j;
                for (uint32_t j = 32; j < c1; j++) w1 |= (uint32_t)(tab[(size_t)perm[j] * 32 + (row[perm[j]] & 31u)] & 1u) << (j - 32);
                keys[i] = ((uint64_t)gray_rank_host(w0) << 32) | gray_rank_host(w1);
            }
        });
        // LSD radix sort of (key, row) pairs, 8 bits a pass, stable like the device's (DeviceRadixSort): every thread
        // counts its slice, the slices' counts are laid out digit-major, every thread scatters its slice in order.
        // (std::stable_sort through an indirect comparison took 2-3 s for 10M rows.)
        std::vector<uint64_t> keys2(n);
        std::vector<uint32_t> src2(n);
        std::vector<size_t> counts((size_t)T * 256);
        uint64_t *ka = keys.data(), *kb = keys2.data();
        uint32_t *ia = src.data(), *ib = src2.data();
        for (int pass = 0; pass < 8; pass++) {
            const int shift = 8 * pass;
            parallel([&](unsigned t) {
                size_t *c = &counts[(size_t)t * 256];
                std::fill(c, c + 256, (size_t)0);
                for (uint64_t i = n * t / T, e = n * (t + 1) / T; i < e; i++) c[(ka[i] >> shift) & 255u]++;
            });
            size_t run = 0;
            for (unsigned d = 0; d < 256; d++)
                for (unsigned t = 0; t < T; t++) {
                    const size_t c = counts[(size_t)t * 256 + d];
                    counts[(size_t)t * 256 + d] = run;
                    run += c;
                }
            parallel([&](unsigned t) {
                size_t *c = &counts[(size_t)t * 256];
                for (uint64_t i = n * t / T, e = n * (t + 1) / T; i < e; i++) {
                    const size_t at = c[(ka[i] >> shift) & 255u]++;
                    kb[at] = ka[i];
                    ib[at] = ia[i];
                }
            });
            std::swap(ka, kb);
            std::swap(ia, ib);
        }
        // (8 passes: the result is back in keys / src)
    }
    // ---- tiles: planes[tile][p][w][256], order[position] = source row, zone[tile]
    const uint64_t n_tiles = (n + 255) / 256;
    std::vector<uint32_t> planes((size_t)n_tiles * P * W * 256, 0), order((size_t)n_tiles * 256, 0);
    std::vector<uint32_t> zone((size_t)n_tiles * 4, 0);
    parallel([&](unsigned t) {
        for (uint64_t tile = n_tiles * t / T, te = n_tiles * (t + 1) / T; tile < te; tile++) {
            uint32_t *tp = planes.data() + (size_t)tile * P * W * 256;
            uint32_t land[2] = {0xffffffffu, 0xffffffffu}, lor[2] = {0u, 0u};
            for (uint32_t r = 0; r < 256; r++) {
                const uint64_t pos = tile * 256 + r;
                if (pos >= n) break;
                const uint8_t *row = codes + (size_t)src[pos] * L;
                order[pos] = src[pos];
                for (uint32_t j = 0; j < L; j++) {
                    const uint32_t code = tab[(size_t)perm[j] * 32 + (row[perm[j]] & 31u)];
                    for (uint32_t p = 0; p < P; p++)
                        if ((code >> p) & 1u) tp[((size_t)p * W + j / 32) * 256 + r] |= 1u << (j % 32);
                }
                for (uint32_t w = 0; w < 2 && w < W; w++) {  // plane 0 is the filter plane
                    const uint32_t x = tp[(size_t)w * 256 + r];
                    land[w] &= x;
                    lor[w] |= x;
                }
            }
            uint32_t *z = &zone[(size_t)tile * 4];  // zone_kernel
            // (bits past the last column are zero in every subject and every query: not shared information)
            const uint32_t cols0 = L >= 32 ? 0xffffffffu : (1u << L) - 1u;
            const uint32_t cols1 = L >= 64 ? 0xffffffffu : L > 32 ? (1u << (L - 32)) - 1u : 0u;
            z[1] = ~(land[0] ^ lor[0]) & cols0;
            z[0] = land[0] & z[1];
            z[3] = W > 1 ? (~(land[1] ^ lor[1]) & cols1) : 0u;
            z[2] = W > 1 ? (land[1] & z[3]) : 0u;
        }
    });
    PackedHeader h{};
    h.alphabet = (uint32_t)alphabet;
    h.seq_len = L;
    h.planes = P;
    h.words = W;
    h.n = n;
    h.n_tiles = n_tiles;
    h.n_runs = 1;
    const uint64_t runs[2] = {n, sorted ? 1u : 0u};
    return write_packed_file(path, h, perm.data(), tab.data(), runs, order.data(), zone.data(), planes.data());
}

}  // namespace smafa
