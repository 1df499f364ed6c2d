// packed.h — the packed store file (see packed.cpp) and the "subject rows" view the drivers read subjects through.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../engine.h"

namespace smafa {

struct PackedHeader {
    uint32_t alphabet, seq_len, planes, words;
    uint64_t n, n_tiles, n_runs;
    uint64_t off_perm, off_tab, off_runs, off_inv, off_order, off_zone, off_planes, file_bytes;
};

bool is_packed_file(const uint8_t *p, size_t len);

// A packed store file, mapped.  open() validates everything later readers index with.
class PackedStore {
  public:
    PackedHeader h{};
    const uint32_t *perm = nullptr;
    const uint8_t *tab = nullptr;
    const uint64_t *runs = nullptr;  // n_runs x {rows, sorted}
    const uint32_t *inv = nullptr, *order = nullptr, *planes = nullptr;
    const void *zone = nullptr;
    int open(const char *path);
    int row(uint64_t subject, uint8_t *codes_out) const;  // the subject's code bytes (seq_len of them)
    PackedStore() = default;
    PackedStore(const PackedStore &) = delete;
    PackedStore &operator=(const PackedStore &) = delete;
    ~PackedStore();

  private:
    void *map_ = nullptr;
    size_t map_len_ = 0;
    std::vector<uint8_t> untab;  // [source column][stored code] -> code, 255 = never produced
};

// column order + per-column code table of a store, from a sample of its rows (host/layout.cpp)
void compute_layout(int alphabet, uint32_t L, const uint8_t *codes, uint64_t n, std::vector<uint32_t> &perm,
                    std::vector<uint8_t> &tab);
// the packed store file of n code rows, packed on the host (no GPU): the same bytes smafa_db_save writes for them
int pack_store_on_host(int alphabet, uint32_t L, const uint8_t *codes, uint64_t n, const char *path);

int write_packed_file(const char *path, const PackedHeader &hdr, const uint32_t *perm, const uint8_t *tab,
                      const uint64_t *runs, const uint32_t *order, const void *zone, const uint32_t *planes);

}  // namespace smafa
