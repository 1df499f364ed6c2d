// fastx.h — FASTA/FASTQ(+gzip) record reader; the role needletail 0.5 plays for the reference
// (call sites /root/reference/src/lib.rs:144,221,381 and src/cluster.rs:28).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace smafa {

struct FastxRecord {
    const uint8_t *id = nullptr;  // header line without the marker (needletail's id())
    size_t id_len = 0;
    const uint8_t *seq = nullptr;  // sequence with line breaks removed (needletail's seq())
    size_t seq_len = 0;
};

class FastxReader {
  public:
    // SMAFA_OK or an error code with smafa_last_error() set
    int open(const char *path);
    // 1 = record, 0 = end of input, negative = error code
    int next(FastxRecord &rec);

  private:
    std::vector<uint8_t> data_;
    std::vector<uint8_t> seq_;
    size_t pos_ = 0;
    bool fastq_ = false;
    std::string path_;
};

}  // namespace smafa
