// packed.cpp — the packed store file: a subject store saved exactly as it lies in HBM (SURVEY 8f1 "raw packed DB,
// new version tag"), so that start-up is mmap -> hipMemcpy instead of a varint decode of <= 46 B per subject
// (/root/reference/src/lib.rs:161-162 writes, :208-218 reads the version-2 form) plus a re-pack on the device.
//
//   byte 0      varint(3)        the reference reads its version from bytes 0..4 (src/lib.rs:214) and rejects this file with
//                                its own "Unsupported db file version: 3." panic, as it should
//   byte 1      varint(2)        kind: 2 = packed tiles (kind 1 is the raw amino-acid code container of dbfile.cpp)
//   bytes 2..6  "SMAFA"
//   byte 7      layout revision  1 = 32-bit column table (perm u32[W*32]); 0 = the 16-bit table of earlier builds: refused
//                                by name ("written by an older build"), not by whatever check its bytes happen to fail
//   byte 8      PackedHeader     little-endian, fixed width; every section starts on a 4096-byte boundary
//   sections    perm  u32[W*32]  packed column j holds source column perm[j]          (layout, engine.hip choose_layout)
//               tab   u8[L*32]   [source column][code] -> stored code
//               runs  {u64 rows, u64 sorted}[n_runs]   the appends the store was built from (zone-level eligibility)
//               inv   u32[n]     subject index -> position (for decoding a subject on the host)
//               order u32[n_tiles*256]  position -> subject index (what a scan reports)
//               zone  uint4[n_tiles]    shared filter bits per wave tile
//               planes u32[n_tiles*P*W*256]   the bit-plane tiles
// A subject's symbols are recovered from the planes on the host (PackedStore::row): only hit rows need it.
#include "packed.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>

namespace smafa {

constexpr char kLayoutRevision = 1;
static const char kMagic[8] = {0x03, 0x02, 'S', 'M', 'A', 'F', 'A', kLayoutRevision};

// any revision of the packed store file (the revision itself is checked, with its own message, by PackedStore::open)
bool is_packed_file(const uint8_t *p, size_t len) { return len >= sizeof kMagic && memcmp(p, kMagic, sizeof kMagic - 1) == 0; }

PackedStore::~PackedStore() {
    if (map_) munmap(map_, map_len_);
}

int PackedStore::open(const char *path) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return set_error(SMAFA_ERR_IO, "%s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < (off_t)(sizeof kMagic + sizeof(PackedHeader))) {
        close(fd);
        return set_error(SMAFA_ERR_FORMAT, "%s: not a packed store file", path);
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return set_error(SMAFA_ERR_IO, "%s: mmap failed: %s", path, strerror(errno));
    map_ = m;
    map_len_ = (size_t)st.st_size;
    const uint8_t *p = (const uint8_t *)m;
    if (!is_packed_file(p, map_len_)) return set_error(SMAFA_ERR_FORMAT, "%s: not a packed store file", path);
    if ((char)p[sizeof kMagic - 1] != kLayoutRevision)
        return set_error(SMAFA_ERR_FORMAT, "%s: packed store written by %s build (layout revision %u, this build reads %u): re-run makedb --packed",
                         path, (char)p[sizeof kMagic - 1] < kLayoutRevision ? "an older" : "a newer", (unsigned)p[sizeof kMagic - 1],
                         (unsigned)kLayoutRevision);
    memcpy(&h, p + sizeof kMagic, sizeof h);
    // ---- everything a reader (host decode, device kernels) indexes with is checked here, once
    const uint64_t L = h.seq_len, W = h.words, P = h.planes;
    const bool aa = h.alphabet == SMAFA_ALPHABET_AA;
    bool ok = (h.alphabet == SMAFA_ALPHABET_NT || aa) && L >= 1 && W == (L + 31) / 32 && (aa ? P == 5 : (P == 2 || P == 3)) &&
              h.n <= 0xffffff00ull && h.n_tiles == (h.n + 255) / 256 && h.file_bytes == map_len_ && h.n_runs <= (1u << 20);
    auto inside = [&](uint64_t off, uint64_t bytes) { return off % 64 == 0 && off <= map_len_ && bytes <= map_len_ - off; };
    ok = ok && inside(h.off_perm, W * 32 * 4) && inside(h.off_tab, L * 32) && inside(h.off_runs, h.n_runs * 16) &&
         inside(h.off_inv, h.n * 4) && inside(h.off_order, h.n_tiles * 256 * 4) && inside(h.off_zone, h.n_tiles * 16) &&
         inside(h.off_planes, h.n_tiles * P * W * 256 * 4);
    if (!ok) return set_error(SMAFA_ERR_FORMAT, "%s: malformed packed store header", path);
    perm = (const uint32_t *)(p + h.off_perm);
    tab = p + h.off_tab;
    runs = (const uint64_t *)(p + h.off_runs);
    inv = (const uint32_t *)(p + h.off_inv);
    order = (const uint32_t *)(p + h.off_order);
    zone = p + h.off_zone;
    planes = (const uint32_t *)(p + h.off_planes);
    // perm: a permutation of the columns; tab: injective per column on the alphabet's codes, values below 2^P (5 planes: 32)
    std::vector<uint8_t> seen(L, 0);
    for (uint64_t j = 0; j < L; j++) {
        if (perm[j] >= L || seen[perm[j]]) return set_error(SMAFA_ERR_FORMAT, "%s: column order is not a permutation", path);
        seen[perm[j]] = 1;
    }
    const uint32_t n_sym = aa ? 28u : 5u, lim = aa ? 32u : 8u;
    untab.assign(L * 32, 255);
    for (uint64_t c = 0; c < L; c++)
        for (uint32_t v = 0; v < n_sym; v++) {
            const uint8_t s = tab[c * 32 + v];
            if (s >= lim || untab[c * 32 + s] != 255) return set_error(SMAFA_ERR_FORMAT, "%s: code table is not injective", path);
            untab[c * 32 + s] = (uint8_t)v;
        }
    uint64_t run_rows = 0;
    for (uint64_t r = 0; r < h.n_runs; r++) run_rows += runs[2 * r];
    if (run_rows != h.n) return set_error(SMAFA_ERR_FORMAT, "%s: run list does not add up to the row count", path);
    // order: a permutation of the subjects (the device writes out[order[pos]], a scan reports order[pos]); inv: its inverse.
    // inv[order[pos]] == pos for every pos < n makes order injective into [0, n), hence a permutation, and inv its inverse.
    for (uint64_t pos = 0; pos < h.n; pos++) {
        const uint32_t s = order[pos];
        if (s >= h.n || inv[s] != pos)
            return set_error(SMAFA_ERR_FORMAT, "%s: subject order table is not a permutation (position %llu)", path,
                             (unsigned long long)pos);
    }
    return SMAFA_OK;
}

int PackedStore::row(uint64_t subject, uint8_t *out) const {
    if (subject >= h.n) return set_error(SMAFA_ERR_INVALID, "subject %llu of %llu", (unsigned long long)subject, (unsigned long long)h.n);
    const uint64_t pos = inv[subject];
    if (pos >= h.n || order[pos] != subject)
        return set_error(SMAFA_ERR_FORMAT, "packed store: position table does not match the order table at subject %llu",
                         (unsigned long long)subject);
    const uint32_t L = h.seq_len, W = h.words, P = h.planes;
    const uint32_t *tile = planes + (pos >> 8) * (uint64_t)P * W * 256 + (pos & 255);
    for (uint32_t j = 0; j < L; j++) {
        uint32_t stored = 0;
        for (uint32_t p = 0; p < P; p++) stored |= ((tile[((uint64_t)p * W + j / 32) * 256] >> (j % 32)) & 1u) << p;
        const uint32_t col = perm[j];
        const uint8_t code = untab[(uint64_t)col * 32 + stored];
        if (code == 255) return set_error(SMAFA_ERR_PANIC, "Invalid character in query sequence: %u", stored);  // src/lib.rs:127
        out[col] = code;
    }
    return SMAFA_OK;
}

int write_packed_file(const char *path, const PackedHeader &hdr_in, const uint32_t *perm, const uint8_t *tab,
                      const uint64_t *runs, const uint32_t *order, const void *zone, const uint32_t *planes) {
    PackedHeader h = hdr_in;
    auto align = [](uint64_t x) { return (x + 4095) / 4096 * 4096; };
    const uint64_t L = h.seq_len, W = h.words, P = h.planes;
    uint64_t off = align(sizeof kMagic + sizeof(PackedHeader));
    h.off_perm = off, off = align(off + W * 32 * 4);
    h.off_tab = off, off = align(off + L * 32);
    h.off_runs = off, off = align(off + h.n_runs * 16);
    h.off_inv = off, off = align(off + h.n * 4);
    h.off_order = off, off = align(off + h.n_tiles * 256 * 4);
    h.off_zone = off, off = align(off + h.n_tiles * 16);
    h.off_planes = off, off = off + h.n_tiles * P * W * 256 * 4;
    h.file_bytes = off;
    std::vector<uint32_t> inv(h.n);
    for (uint64_t pos = 0; pos < h.n; pos++) {
        if (order[pos] >= h.n) return set_error(SMAFA_ERR_INVALID, "order table entry out of range");
        inv[order[pos]] = (uint32_t)pos;
    }
    // written beside the target and renamed over it once complete and on disk: an interrupted `makedb --packed` never
    // leaves a file with a valid header over a truncated body
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return set_error(SMAFA_ERR_IO, "%s: %s", path, strerror(errno));
    bool ok = true;
    auto put = [&](uint64_t at, const void *p, uint64_t bytes) {
        ok = ok && fseeko(f, (off_t)at, SEEK_SET) == 0 && (bytes == 0 || fwrite(p, 1, bytes, f) == bytes);
    };
    put(0, kMagic, sizeof kMagic);
    put(sizeof kMagic, &h, sizeof h);
    put(h.off_perm, perm, W * 32 * 4);
    put(h.off_tab, tab, L * 32);
    put(h.off_runs, runs, h.n_runs * 16);
    put(h.off_inv, inv.data(), h.n * 4);
    put(h.off_order, order, h.n_tiles * 256 * 4);
    put(h.off_zone, zone, h.n_tiles * 16);
    put(h.off_planes, planes, h.n_tiles * P * W * 256 * 4);
    if (ftruncate(fileno(f), (off_t)h.file_bytes) != 0) ok = false;
    if (ok && (fflush(f) != 0 || fsync(fileno(f)) != 0)) ok = false;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path) != 0) {
        const int err = errno;
        (void)unlink(tmp.c_str());
        return set_error(SMAFA_ERR_IO, "%s: write error: %s", path, strerror(err));
    }
    return SMAFA_OK;
}

int SubjectRows::get(uint64_t j, uint8_t *out) const {
    if (codes) {
        memcpy(out, codes + (size_t)j * L, L);
        return SMAFA_OK;
    }
    return packed->row(j, out);  // a damaged plane block fails the query instead of printing placeholder columns
}

}  // namespace smafa
