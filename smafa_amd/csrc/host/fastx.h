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
    // the whole (decompressed) input and its format, for the bulk loader
    const uint8_t *data() const { return base_; }
    size_t size() const { return size_; }
    bool is_fastq() const { return fastq_; }
    FastxReader() = default;
    FastxReader(const FastxReader &) = delete;
    FastxReader &operator=(const FastxReader &) = delete;
    ~FastxReader();

  private:
    // plain regular files are mapped (no copy; the bulk loader's threads fault the pages in side by side);
    // gzip input and pipes are read into data_
    const uint8_t *base_ = nullptr;
    size_t size_ = 0;
    void *map_ = nullptr;
    std::vector<uint8_t> data_;
    std::vector<uint8_t> seq_;
    size_t pos_ = 0;
    bool fastq_ = false;
    std::string path_;
};

}  // namespace smafa

namespace smafa {

// Bulk load of a FASTA/FASTQ(+gzip) file of equal-length records into code rows (and optionally the raw sequence
// bytes, for cluster's first output column).  Inputs of 32 MB and more are parsed and encoded by several threads: the
// file is cut at record starts (FASTA: a '>' that begins a line; 4-line FASTQ: an '@' line whose third line is a '+'
// line and whose fourth is as long as its second, verified against the neighbouring chunk afterwards); a gzip file is
// inflated by one thread (a gzip stream cannot be split) while the others parse what has already come out.  Small
// inputs, multi-member gzip files and anything the chunking cannot vouch for go through FastxReader on one thread.
// Stops at the FIRST offending record in file order, exactly where the sequential reader would:
//   err_kind 0 none | 1 byte outside the alphabet (err_msg = the reference's panic text) | 2 length differs from
//   the first record's (err_len = its length) | 3 the first record is empty | 4 parse error (err_msg)
// Records before it are complete in codes/raw (n of them).
struct BulkRecords {
    std::vector<uint8_t> codes, raw;
    uint64_t n = 0;
    size_t L = 0;
    int err_kind = 0;
    size_t err_len = 0;
    std::string err_msg;
};
int load_records_bulk(const char *path, int alphabet, bool want_raw, BulkRecords &out);
// One PART of a plain (not gzip) FASTA/FASTQ file, for hosts that shard a query file over several processes (one per
// GPU): the records that START in [cut(part), cut(part + 1)), cut(i) = the first record start at or after byte
// size * i / parts — every process computes the same cuts, so the parts partition the records in file order and each
// process reads only its own bytes.  `out` as load_records_bulk fills it (L = the length of the PART's first record; the
// caller compares it with the store's).  *usable = false: this file cannot be taken in parts (gzip, not FASTX, a cut that
// did not hold, a malformed record) — every process must then load the whole file (load_records_bulk) instead.
int load_records_part(const char *path, int alphabet, unsigned part, unsigned parts, BulkRecords &out, bool *usable);
// Size of the file's contents once decompressed (gzip: the ISIZE trailer; plain: the file size; 0: cannot tell) — how the
// drivers decide between streaming a query file and bulk-loading it with all threads.
uint64_t fastx_expanded_size(const char *path);

}  // namespace smafa
