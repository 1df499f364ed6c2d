// drivers.cpp — the reference crate's public functions over the HIP scan engine:
//   makedb  (/root/reference/src/lib.rs:137-165)   host only
//   query   (/root/reference/src/lib.rs:198-325)   scan on the GPU, selection + TSV on the host
//   cluster (/root/reference/src/cluster.rs:13-94) batched, exact restatement of the greedy loop
//   count   (/root/reference/src/lib.rs:378-398)   host only
// Output bytes are the reference's: `query` prints "{query}\t{subject}\t{distance}\t{subject string}\n"
// (src/lib.rs:292,310), `cluster` prints "{raw record}\t{centroid string}\n" (src/cluster.rs:79-84).
// Inputs the reference panics on fail with SMAFA_ERR_PANIC and the same message, after the rows the
// reference would already have printed.
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../engine.h"
#include "fastx.h"
#include "packed.h"

namespace smafa {

namespace {

int write_all(int fd, const char *p, size_t n) {
    while (n) {
        const ssize_t w = ::write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return set_error(SMAFA_ERR_IO, "write failed: %s", strerror(errno));
        }
        p += w;
        n -= (size_t)w;
    }
    return SMAFA_OK;
}

// encode one record; on a byte outside the alphabet produce the panic text of src/lib.rs:38-41
int encode_record(int alphabet, const FastxRecord &rec, std::vector<uint8_t> &codes) {
    const size_t base = codes.size();
    codes.resize(base + rec.seq_len);
    for (size_t i = 0; i < rec.seq_len; i++) {
        const uint8_t c = code_of(alphabet, rec.seq[i]);
        if (c == 255) {
            codes.resize(base);
            return set_error(SMAFA_ERR_PANIC, "Byte %u cannot be interpreted as %s, in sequence \"%.*s\" at position %zu",
                             rec.seq[i], alphabet_noun(alphabet), (int)rec.id_len, (const char *)rec.id, i);
        }
        codes[base + i] = c;
    }
    return SMAFA_OK;
}

void append_u32(std::string &s, uint32_t v) {
    char tmp[12];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) s.push_back(tmp[--n]);
}

void append_decoded(std::string &s, int alphabet, const uint8_t *codes, uint32_t L) {
    const size_t base = s.size();
    s.resize(base + L);
    for (uint32_t i = 0; i < L; i++) s[base + i] = letter_of(alphabet, codes[i]);
}

}  // namespace

// "{query}\t{subject}\t{distance}\t{subject string}\n" per row (src/lib.rs:292,310), written to fd in order.  Big row
// lists are formatted by several threads, each into its own buffer over a contiguous slice of the rows.
int write_rows_text(const smafa_hit *rows, size_t n, const SubjectRows &subjects, int alphabet, uint32_t q_base, int fd) {
    const uint32_t L = subjects.L;
    // (a subject that cannot be read back — a damaged packed store — fails the query; the error text is per thread, so the
    // first failing slice's code and message are carried to the caller)
    auto format = [&](size_t lo, size_t hi, std::string &text, int &frc, std::string &fmsg) {
        frc = SMAFA_OK;
        text.clear();
        text.reserve((hi - lo) * ((size_t)L + 24));
        std::vector<uint8_t> row(L);
        for (size_t i = lo; i < hi; i++) {
            const smafa_hit &h = rows[i];
            append_u32(text, q_base + h.query);
            text.push_back('\t');
            append_u32(text, h.subject);
            text.push_back('\t');
            append_u32(text, h.dist);
            text.push_back('\t');
            frc = subjects.get(h.subject, row.data());
            if (frc) {
                fmsg = smafa_last_error();
                return;
            }
            append_decoded(text, alphabet, row.data(), L);
            text.push_back('\n');
        }
    };
    const size_t block = 1u << 18;  // rows per write
    const unsigned T = n >= (1u << 16) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    std::vector<std::string> parts(T), msgs(T);
    std::vector<int> rcs(T, SMAFA_OK);
    for (size_t b0 = 0; b0 < n; b0 += block * T) {
        const size_t b1 = std::min(n, b0 + block * T), span = b1 - b0;
        if (T == 1) {
            format(b0, b1, parts[0], rcs[0], msgs[0]);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < T; t++)
                pool.emplace_back([&, t] { format(b0 + span * t / T, b0 + span * (t + 1) / T, parts[t], rcs[t], msgs[t]); });
            for (auto &th : pool) th.join();
        }
        for (unsigned t = 0; t < T; t++) {
            // the rows in front of the damaged one are written, as a sequential writer would have
            int rc = write_all(fd, parts[t].data(), parts[t].size());
            if (rc) return rc;
            if (rcs[t]) return set_error(rcs[t], "%s", msgs[t].c_str());
        }
    }
    return SMAFA_OK;
}

// The reference `.expect()`s its FASTX inputs: parse_fastx_file(..).expect(what) on the path and record.expect(what) on
// every record (src/lib.rs:144,149,221,234; src/cluster.rs:28,39), so an unreadable, empty, non-FASTX or truncated input
// is a panic — exit 101 — not an Err.  (The DB file is opened with `?`: src/lib.rs:208-210,218 stay SMAFA_ERR_IO /
// SMAFA_ERR_FORMAT, exit 1; so does everything in `count`, src/lib.rs:381,385.)
int expect_fastx(int rc, const char *what) {
    if (rc != SMAFA_ERR_IO && rc != SMAFA_ERR_FORMAT) return rc;
    const std::string msg = smafa_last_error();
    return set_error(SMAFA_ERR_PANIC, "%s: %s", what, msg.c_str());
}

namespace {

struct DbGuard {
    smafa_db *db = nullptr;
    ~DbGuard() { smafa_db_destroy(db); }
};

struct FreeGuard {
    void *p = nullptr;
    ~FreeGuard() { smafa_free(p); }
};

// First occurrences of every distinct row, in input order — the HashSet<Vec<u64>> test of src/cluster.rs:24,46-48.
// Rows are hashed by all threads, then thread t owns the rows whose hash falls in partition t and runs them, in input
// order, through its own open-addressing table; "first" is decided inside one partition, so the result does not
// depend on the number of threads.
inline uint64_t row_hash(const uint8_t *p, uint32_t L) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    uint32_t i = 0;
    for (; i + 8 <= L; i += 8) {
        uint64_t v;
        memcpy(&v, p + i, 8);
        h = (h ^ v) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    for (; i < L; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h ^ (h >> 29);
}

void first_occurrences(const uint8_t *rows, uint64_t n, uint32_t L, std::vector<uint32_t> &uniq) {
    const unsigned T = n >= (1u << 18) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    std::vector<uint64_t> hash(n);
    std::vector<uint8_t> first(n, 0);
    auto run = [&](auto &&fn) {
        if (T == 1) return fn(0u);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; t++) pool.emplace_back(fn, t);
        for (auto &th : pool) th.join();
    };
    run([&](unsigned t) {
        for (uint64_t i = n * t / T, e = n * (t + 1) / T; i < e; i++) hash[i] = row_hash(rows + i * L, L);
    });
    run([&](unsigned t) {
        auto mine = [&](uint64_t h) { return (unsigned)((h >> 40) % T) == t; };  // slot bits come from the low end
        size_t count = 0;
        for (uint64_t i = 0; i < n; i++) count += mine(hash[i]);
        size_t cap = 16;
        while (cap < count * 2) cap *= 2;
        std::vector<uint32_t> slots(cap, UINT32_MAX);
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t h = hash[i];
            if (!mine(h)) continue;
            size_t s = h & (cap - 1);
            bool seen = false;
            while (slots[s] != UINT32_MAX) {
                const uint32_t j = slots[s];
                if (hash[j] == h && memcmp(rows + (size_t)j * L, rows + i * L, L) == 0) {
                    seen = true;
                    break;
                }
                s = (s + 1) & (cap - 1);
            }
            if (!seen) {
                slots[s] = (uint32_t)i;
                first[i] = 1;
            }
        }
    });
    uniq.clear();
    for (uint64_t i = 0; i < n; i++)
        if (first[i]) uniq.push_back((uint32_t)i);
}

}  // namespace

}  // namespace smafa

using namespace smafa;

extern "C" {

// ---------------------------------------------------------------------------------- fastx_load
// The records BEFORE the first offending one, and what is wrong with that one (*pending = 0 or the error code, its text in
// smafa_last_error()): what a driver needs to print the rows the reference prints before it panics (src/lib.rs:232-318).
int smafa_fastx_load_partial(const char *path, int alphabet, uint8_t **codes_out, uint64_t *n_out, uint32_t *seq_len,
                             int *pending) try {
    if (!path || !codes_out || !n_out || !seq_len || !pending)
        return set_error(SMAFA_ERR_INVALID, "smafa_fastx_load_partial: NULL argument");
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    *codes_out = nullptr;
    *n_out = 0;
    *seq_len = 0;
    *pending = SMAFA_OK;
    BulkRecords recs;
    int rc = load_records_bulk(path, alphabet, false, recs);
    if (rc) return rc;
    uint8_t *out = (uint8_t *)malloc(recs.codes.empty() ? 1 : recs.codes.size());
    if (!out) return set_error(SMAFA_ERR_NOMEM, "out of host memory");
    if (!recs.codes.empty()) memcpy(out, recs.codes.data(), recs.codes.size());
    *codes_out = out;
    *n_out = recs.n;
    *seq_len = (uint32_t)recs.L;
    if (recs.err_kind == 1) *pending = set_error(SMAFA_ERR_PANIC, "%s", recs.err_msg.c_str());
    else if (recs.err_kind == 2)
        *pending = set_error(SMAFA_ERR_PANIC, "Cannot compute distances between seq of length %zu and windows of lengths %zu",
                             recs.err_len, recs.L);
    else if (recs.err_kind == 3) *pending = set_error(SMAFA_ERR_PANIC, "Cannot add empty sequence to WindowSet");
    else if (recs.err_kind == 4) *pending = set_error(SMAFA_ERR_FORMAT, "%s", recs.err_msg.c_str());
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_fastx_load_partial");
}

int smafa_fastx_load(const char *path, int alphabet, uint8_t **codes_out, uint64_t *n_out, uint32_t *seq_len) try {
    if (!path || !codes_out || !n_out || !seq_len) return set_error(SMAFA_ERR_INVALID, "smafa_fastx_load: NULL argument");
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    *codes_out = nullptr;
    *n_out = 0;
    *seq_len = 0;
    BulkRecords recs;
    int rc = load_records_bulk(path, alphabet, false, recs);
    if (rc) return rc;
    if (recs.err_kind == 1) return set_error(SMAFA_ERR_PANIC, "%s", recs.err_msg.c_str());
    if (recs.err_kind == 2)
        return set_error(SMAFA_ERR_PANIC, "Cannot compute distances between seq of length %zu and windows of lengths %zu",
                         recs.err_len, recs.L);
    if (recs.err_kind == 3) return set_error(SMAFA_ERR_PANIC, "Cannot add empty sequence to WindowSet");
    if (recs.err_kind == 4) return set_error(SMAFA_ERR_FORMAT, "%s", recs.err_msg.c_str());
    std::vector<uint8_t> &codes = recs.codes;
    const uint64_t n = recs.n;
    const size_t L = recs.L;
    uint8_t *out = (uint8_t *)malloc(codes.empty() ? 1 : codes.size());
    if (!out) return set_error(SMAFA_ERR_NOMEM, "out of host memory");
    if (!codes.empty()) memcpy(out, codes.data(), codes.size());
    *codes_out = out;
    *n_out = n;
    *seq_len = (uint32_t)L;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_fastx_load");
}

// ------------------------------------------------------------------------------------- makedb
int smafa_makedb(const char *subject_fasta, const char *db_path, int alphabet) try {
    if (!subject_fasta || !db_path) return set_error(SMAFA_ERR_INVALID, "smafa_makedb: NULL path");
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    BulkRecords recs;
    const double t_start = now_seconds();
    int rc = load_records_bulk(subject_fasta, alphabet, false, recs);  // src/lib.rs:143-152
    if (rc) return expect_fastx(rc, "valid path/file of subject fasta");  // src/lib.rs:144
    const double t_parsed = now_seconds();
    if (recs.err_kind == 3) return set_error(SMAFA_ERR_PANIC, "Cannot add empty sequence to WindowSet");  // src/lib.rs:103-108
    if (recs.err_kind == 1) return set_error(SMAFA_ERR_PANIC, "%s", recs.err_msg.c_str());                  // src/lib.rs:38-41
    if (recs.err_kind == 2)  // src/lib.rs:92-101
        return set_error(SMAFA_ERR_PANIC, "WindowSet seq length is %zu, got a new sequence of length %zu", recs.L, recs.err_len);
    if (recs.err_kind == 4) return set_error(SMAFA_ERR_PANIC, "valid record: %s", recs.err_msg.c_str());  // src/lib.rs:149
    std::vector<uint8_t> &codes = recs.codes;
    const uint64_t n = recs.n;
    const size_t L = recs.L;
    if (L > 0xffffffffull) return set_error(SMAFA_ERR_INVALID, "sequence too long");
    log_line(1, "Encoding of %llu sequences complete, writing db file %s", (unsigned long long)n, db_path);  // src/lib.rs:154-158
    rc = smafa_dbfile_write(db_path, alphabet, codes.data(), n, (uint32_t)L);  // src/lib.rs:161-162
    if (rc == SMAFA_OK) log_line(1, "DB file written");
    log_line(2, "makedb: parse + encode %.2f s, serialise + write %.2f s", t_parsed - t_start, now_seconds() - t_parsed);
    return rc;
} catch (...) {
    return smafa::exception_code("smafa_makedb");
}

// makedb with the packed store file as output (host/packed.cpp): the same parse + encode, then the subjects are packed on
// the device exactly as `query` would pack them, and the resident store is saved.
int smafa_makedb_packed(const char *subject_fasta, const char *db_path, int alphabet, int device) try {
    if (!subject_fasta || !db_path) return set_error(SMAFA_ERR_INVALID, "smafa_makedb_packed: NULL path");
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    std::thread warm(warm_device, device < 0 ? 1 << 30 : device);  // (an out-of-range device: nothing to warm)
    BulkRecords recs;
    const double t_start = now_seconds();
    int rc = load_records_bulk(subject_fasta, alphabet, false, recs);  // src/lib.rs:143-152
    const double t_parsed = now_seconds();
    warm.join();
    if (rc) return expect_fastx(rc, "valid path/file of subject fasta");
    if (recs.err_kind == 3) return set_error(SMAFA_ERR_PANIC, "Cannot add empty sequence to WindowSet");
    if (recs.err_kind == 1) return set_error(SMAFA_ERR_PANIC, "%s", recs.err_msg.c_str());
    if (recs.err_kind == 2)
        return set_error(SMAFA_ERR_PANIC, "WindowSet seq length is %zu, got a new sequence of length %zu", recs.L, recs.err_len);
    if (recs.err_kind == 4) return set_error(SMAFA_ERR_PANIC, "valid record: %s", recs.err_msg.c_str());
    if (recs.n == 0) return set_error(SMAFA_ERR_INVALID, "a packed store needs at least one sequence (its length fixes the layout)");
    if (recs.L > 0xffffffffull) return set_error(SMAFA_ERR_INVALID, "sequence too long");
    if (device < 0 || smafa_device_count() == 0) {  // no GPU (or none wanted): the same file, packed by host threads
        log_line(1, "Encoding of %llu sequences complete, packing on the host and writing %s", (unsigned long long)recs.n, db_path);
        rc = pack_store_on_host(alphabet, (uint32_t)recs.L, recs.codes.data(), recs.n, db_path);
        if (rc == SMAFA_OK) log_line(1, "DB file written");
        log_line(2, "makedb --packed: parse + encode %.2f s, pack on the host + write %.2f s", t_parsed - t_start, now_seconds() - t_parsed);
        return rc;
    }
    log_line(1, "Encoding of %llu sequences complete, packing on device %d and writing %s", (unsigned long long)recs.n, device, db_path);
    DbGuard guard;
    const double t_ready = now_seconds();
    rc = smafa_db_create(&guard.db, device, alphabet, (uint32_t)recs.L);
    if (!rc) rc = smafa_db_append(guard.db, recs.codes.data(), recs.n);
    const double t_packed = now_seconds();
    if (!rc) rc = smafa_db_save(guard.db, db_path);
    if (rc == SMAFA_OK) log_line(1, "DB file written");
    log_line(2, "makedb --packed: parse + encode %.2f s (device ready %.2f s after start), pack on the device %.2f s, copy back + write %.2f s",
             t_parsed - t_start, t_ready - t_start, t_packed - t_ready, now_seconds() - t_packed);
    return rc;
} catch (...) {
    return smafa::exception_code("smafa_makedb_packed");
}

// -------------------------------------------------------------------------------------- query
// One process, one handle per entry of `devices` (entries may repeat: several handles on one GPU), the store replicated
// on each.  The query loop of the reference carries no state between records except the running query number
// (src/lib.rs:232-318), so every chunk of queries is cut into ndev contiguous blocks, block g is scanned and selected by
// host thread g on handle g, and the blocks' rows are printed in block order: the output does not depend on ndev.
int smafa_query_multi(const char *db_path, const char *query_fasta, uint32_t max_divergence, uint32_t max_num_hits,
                      uint32_t limit_per_sequence, int out_fd, const int *devices, int ndev) try {
    if (!db_path || !query_fasta) return set_error(SMAFA_ERR_INVALID, "smafa_query: NULL path");
    if (!devices || ndev < 1 || ndev > 64) return set_error(SMAFA_ERR_INVALID, "smafa_query_multi: 1 to 64 devices expected");
    int alphabet = 0;
    uint64_t n = 0;
    uint32_t L = 0;
    FreeGuard codes_guard;
    uint8_t *codes = nullptr;
    const double t_start = now_seconds();
    struct Warm {  // device bring-up overlaps the file read and decode; joined before the first device call or return
        std::vector<std::thread> t;
        Warm(const int *devices, int ndev) {
            for (int g = 0; g < ndev; g++) {
                bool seen = false;
                for (int h = 0; h < g; h++) seen = seen || devices[h] == devices[g];
                if (!seen) t.emplace_back(warm_device, devices[g]);
            }
        }
        void join() {
            for (auto &th : t)
                if (th.joinable()) th.join();
        }
        ~Warm() { join(); }
    } warm(devices, ndev);
    log_line(1, "Decoding db file \"%s\"", db_path);  // src/lib.rs:206
    PackedStore pk;  // a packed store file is mapped, not decoded (host/packed.cpp)
    bool packed = false;
    {
        uint8_t head[8] = {0};
        FILE *f = fopen(db_path, "rb");
        if (f) {
            const size_t got = fread(head, 1, sizeof head, f);
            fclose(f);
            packed = is_packed_file(head, got);
        }
    }
    int rc;
    if (packed) {
        rc = pk.open(db_path);
        if (rc) return rc;
        alphabet = (int)pk.h.alphabet;
        n = pk.h.n;
        L = pk.h.seq_len;
    } else {
        rc = smafa_dbfile_read(db_path, &alphabet, &codes, &n, &L);  // src/lib.rs:208-218
        if (rc) return rc;
        codes_guard.p = codes;
    }
    SubjectRows subjects;
    subjects.codes = codes;
    subjects.packed = packed ? &pk : nullptr;
    subjects.L = L;
    log_line(2, "db %s: %llu sequences of length %u in %.2f s", packed ? "mapped" : "decoded", (unsigned long long)n, L,
             now_seconds() - t_start);

    // Query files of 32 MB and more (after decompression) are parsed and encoded up front by all threads
    // (load_records_bulk: FASTA or FASTQ, plain or gzip); smaller ones are streamed record by record.
    const uint64_t q_bytes = fastx_expanded_size(query_fasta);
    const bool bulk_queries = n > 0 && q_bytes >= (32u << 20) && q_bytes <= (4ull << 30);
    FastxReader reader;
    if (!bulk_queries) {
        rc = reader.open(query_fasta);  // src/lib.rs:221
        if (rc) return expect_fastx(rc, "valid path/file of query fasta");
    }

    // the store on every entry of `devices`: a group (host/group.cpp; public: smafa_group_*)
    struct GroupGuard {
        smafa_group *g = nullptr;
        ~GroupGuard() { smafa_group_destroy(g); }
    } group;
    if (n > 0) {
        warm.join();
        const double t0 = now_seconds();
        if (packed) {
            rc = group_load_packed(&group.g, devices, ndev, pk);
        } else {
            rc = smafa_group_create(&group.g, devices, ndev, alphabet, L);
            if (!rc) rc = smafa_group_append(group.g, codes, n);
        }
        if (rc) return rc;
        // A query file is scanned chunk by chunk against the same resident store: let every replica decide by itself when its block
        // index has paid for itself (rent or buy, smafa_set_index 3 — a run of a few thousand queries never builds one; a million
        // queries against 50M subjects build it during the first chunk).  An explicit SMAFA_INDEX wins.
        if (!getenv("SMAFA_INDEX"))
            for (int g = 0; g < ndev; g++) (void)smafa_set_index(group_member(group.g, g), 3);
        log_line(2, "subject store packed into HBM on %d handle(s) in %.2f s", ndev, now_seconds() - t0);
    }
    // fn(g) for every block of a chunk on its own host thread (without a store: nothing to scan, one thread)
    auto on_every_handle = [&](const std::function<int(int)> &fn) -> int {
        if (group.g) return group_on_every_handle(group.g, fn);
        for (int g = 0; g < ndev; g++) {
            const int r = fn(g);
            if (r) return r;
        }
        return SMAFA_OK;
    };
    log_line(1, "Querying ..");  // src/lib.rs:230
    double t_scan = 0;

    const bool kmode = max_num_hits != SMAFA_NONE && max_num_hits != 1;  // src/lib.rs:224
    // the device bound: k-th smallest distance (k = 1: the minimum); no k bound when k exceeds the store
    const uint32_t dev_k = !kmode ? 1u : (max_num_hits == 0 || max_num_hits > (uint32_t)n) ? SMAFA_NONE : max_num_hits;
    // rows per query are bounded by the store size when nothing else bounds them
    const bool unbounded = max_divergence == SMAFA_NONE && dev_k == SMAFA_NONE;
    const uint64_t chunk_queries =
        (unbounded ? std::max<uint64_t>(1, (16ull << 20) / std::max<uint64_t>(n, 1)) : 65536) * (uint64_t)ndev;

    std::vector<uint8_t> qcodes;
    uint32_t query_number = 0;  // src/lib.rs:231
    uint64_t in_chunk = 0;
    int pending = SMAFA_OK;  // error to report after the rows already due have been printed
    std::string pending_msg;

    // scan + select + print `count` queries whose code rows start at qptr; query_number already counts them
    std::vector<std::vector<smafa_hit>> block_hits((size_t)ndev), block_rows((size_t)ndev);
    auto run_chunk = [&](const uint8_t *qptr, uint64_t count) -> int {
        if (count == 0) return SMAFA_OK;
        const double t0 = now_seconds();
        // block g = queries [count*g/ndev, count*(g+1)/ndev) of the chunk, on handle g
        int r = on_every_handle([&](int g) -> int {
            const uint64_t lo = count * (uint64_t)g / (uint64_t)ndev, hi = count * (uint64_t)(g + 1) / (uint64_t)ndev;
            block_hits[g].clear();
            block_rows[g].clear();
            if (hi == lo) return SMAFA_OK;
            if (n > 0) {
                int rr = scan_to_host(group_member(group.g, g), qptr + (size_t)lo * L, hi - lo, max_divergence, dev_k, block_hits[g]);
                if (rr) return rr;
            }
            return select_rows(block_hits[g].data(), block_hits[g].size(), hi - lo, n, subjects, max_divergence, max_num_hits,
                               limit_per_sequence, block_rows[g]);
        });
        t_scan += now_seconds() - t0;
        if (r) return r;
        for (int g = 0; g < ndev; g++) {
            const uint64_t lo = count * (uint64_t)g / (uint64_t)ndev;
            r = write_rows_text(block_rows[g].data(), block_rows[g].size(), subjects, alphabet,
                                query_number - (uint32_t)count + (uint32_t)lo, out_fd);
            if (r) return r;
        }
        return SMAFA_OK;
    };
    auto flush = [&]() -> int {
        int r = run_chunk(qcodes.data(), in_chunk);
        in_chunk = 0;
        qcodes.clear();
        return r;
    };
    auto length_panic = [&](size_t len) {  // src/lib.rs:72-79
        char msg[160];
        snprintf(msg, sizeof msg, "Cannot compute distances between seq of length %zu and windows of lengths %u", len, L);
        pending = SMAFA_ERR_PANIC;
        pending_msg = msg;
    };

    if (bulk_queries) {
        // big query files: every record parsed and encoded up front by several threads (the loader stops at the
        // first offending record in file order, like the loop below), then scanned chunk by chunk
        BulkRecords recs;
        rc = load_records_bulk(query_fasta, alphabet, false, recs);
        if (rc) return expect_fastx(rc, "valid path/file of query fasta");
        uint64_t usable = recs.n;
        if (recs.n > 0 && recs.L != L) {  // the first query already fails the length check: nothing is printed
            usable = 0;
            length_panic(recs.L);
        } else if (recs.err_kind == 1) {  // src/lib.rs:38-41
            pending = SMAFA_ERR_PANIC;
            pending_msg = recs.err_msg;
        } else if (recs.err_kind == 2) {
            length_panic(recs.err_len);
        } else if (recs.err_kind == 3) {
            length_panic(0);
        } else if (recs.err_kind == 4) {  // record.expect(..), src/lib.rs:234
            pending = SMAFA_ERR_PANIC;
            pending_msg = "Failed to parse query sequence: " + recs.err_msg;
        }
        if (usable > 0xfffffff0ull) return set_error(SMAFA_ERR_INVALID, "too many queries");
        for (uint64_t off = 0; off < usable; off += chunk_queries) {
            const uint64_t count = std::min<uint64_t>(chunk_queries, usable - off);
            query_number += (uint32_t)count;
            int frc = run_chunk(recs.codes.data() + (size_t)off * L, count);
            if (frc) return frc;
        }
        rc = 0;
    } else {
        FastxRecord rec;
        while ((rc = reader.next(rec)) == 1) {
            int erc = encode_record(alphabet, rec, qcodes);  // src/lib.rs:235
            if (erc) {
                pending = erc;
                pending_msg = smafa_last_error();
                break;
            }
            if (n > 0 && rec.seq_len != L) {  // only checked when the store has a length
                qcodes.resize(qcodes.size() - rec.seq_len);
                length_panic(rec.seq_len);
                break;
            }
            if (n == 0) qcodes.resize(qcodes.size() - rec.seq_len);  // nothing to compare with; selection will panic
            in_chunk++;
            query_number++;
            if (in_chunk >= chunk_queries) {
                int frc = flush();
                if (frc) return frc;
            }
        }
    }
    if (rc < 0 && pending == SMAFA_OK) {  // record.expect("Failed to parse query sequence"), src/lib.rs:234
        pending = expect_fastx(rc, "Failed to parse query sequence");
        pending_msg = smafa_last_error();
    }
    int frc = flush();
    if (frc) return frc;
    if (pending != SMAFA_OK) return set_error(pending, "%s", pending_msg.c_str());
    log_line(2, "%u queries: scans + selection %.2f s on %d handle(s)", query_number, t_scan, ndev);
    log_line(1, "Querying complete, took %llu seconds", (unsigned long long)(now_seconds() - t_start));  // src/lib.rs:320-323
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_query_multi");
}

int smafa_query(const char *db_path, const char *query_fasta, uint32_t max_divergence, uint32_t max_num_hits,
                uint32_t limit_per_sequence, int out_fd, int device) try {
    return smafa_query_multi(db_path, query_fasta, max_divergence, max_num_hits, limit_per_sequence, out_fd, &device, 1);
} catch (...) {
    return smafa::exception_code("smafa_query");
}

// ------------------------------------------------------------------------------------ cluster
//
// The reference handles one record at a time: skip exact duplicates, scan the record against the
// centroids found so far, join the nearest one if it is within max_divergence (ties: lowest centroid
// index), else become a new centroid (src/cluster.rs:35-85).  The same assignment is computed here in
// batches of B unseen records:
//   1. one GPU scan of the batch against the centroids known BEFORE the batch (bound: max_divergence,
//      tightened to each record's minimum) -> per record its nearest old centroid, lowest index on ties;
//   2. records with no old centroid in range are the only ones that can become centroids in this batch
//      ("candidates"); one GPU scan of the batch against the candidates gives every in-range
//      (record, candidate) pair;
//   3. a sequential pass in input order decides, for each record, between its old centroid and the
//      candidates EARLIER in the batch that did become centroids — old centroids have smaller indices
//      than new ones, and new ones are numbered in input order, so "smallest distance, then lowest
//      centroid index" is evaluated exactly as the reference's first-minimum scan (src/cluster.rs:62-68).
// Nothing is approximated: the result is the reference's, record for record.
//
// Sharded over `world` processes (one GPU each, smafa_cluster_sharded): steps 1 and 2 scan only this rank's
// contiguous slice of the batch, the slices' results are exchanged through the caller's allgather, and step 3
// runs identically on every rank, so the replicas of the centroid store stay equal without a broadcast.
// the input of a clustering run, parsed and de-duplicated ONCE (every rank of a sharded run works from the same records)
struct ClusterInput {
    BulkRecords recs;
    std::vector<uint32_t> uniq;  // first occurrences, input order (src/cluster.rs:46-48)
    int pending = SMAFA_OK;      // the record that stopped the load fails AFTER the lines of the records before it are written
    std::string pending_msg;
    double t_start = 0;
};

static int cluster_load(const char *input_fasta, int alphabet, int device, ClusterInput &in) {
    if (!input_fasta) return set_error(SMAFA_ERR_INVALID, "smafa_cluster: NULL path");
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    in.t_start = now_seconds();
    log_line(1, "Clustering ..");  // src/cluster.rs:33
    std::thread warm(warm_device, device);  // device bring-up overlaps the parse
    BulkRecords &recs = in.recs;
    int rc = load_records_bulk(input_fasta, alphabet, true, recs);  // src/cluster.rs:28,35-43 for every record
    struct Joiner {  // the bring-up is waited for after the duplicates have been found too (they need no device)
        std::thread &t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } joiner{warm};
    if (rc) return expect_fastx(rc, "valid path/file of input fasta");  // src/cluster.rs:28
    if (recs.n >= 0xfffffff0ull) return set_error(SMAFA_ERR_INVALID, "too many records");
    if (recs.err_kind == 1) {  // src/lib.rs:38-41
        in.pending = SMAFA_ERR_PANIC;
        in.pending_msg = recs.err_msg;
    } else if (recs.err_kind == 3) {  // first record becomes a centroid: push_encoding, src/lib.rs:103-108
        in.pending = SMAFA_ERR_PANIC;
        in.pending_msg = "Cannot add empty sequence to WindowSet";
    } else if (recs.err_kind == 2) {  // get_distances, src/lib.rs:72-79
        char msg[160];
        snprintf(msg, sizeof msg, "Cannot compute distances between seq of length %zu and windows of lengths %zu",
                 recs.err_len, recs.L);
        in.pending = SMAFA_ERR_PANIC;
        in.pending_msg = msg;
    } else if (recs.err_kind == 4) {  // record.expect(..), src/cluster.rs:39
        in.pending = SMAFA_ERR_PANIC;
        in.pending_msg = "Failed to parse input sequence: " + recs.err_msg;
    }
    const double t_loaded = now_seconds();
    if (recs.n > 0) {
        in.uniq.reserve(recs.n);
        first_occurrences(recs.codes.data(), recs.n, (uint32_t)recs.L, in.uniq);  // src/cluster.rs:46-48
        log_line(2, "parsed %llu records in %.2f s, %zu distinct found in %.2f s", (unsigned long long)recs.n,
                 t_loaded - in.t_start, in.uniq.size(), now_seconds() - t_loaded);
    }
    const double t_wait = now_seconds();
    warm.join();
    log_line(2, "device bring-up: waited %.2f s for it after the load (%.2f s since the start)", now_seconds() - t_wait,
             now_seconds() - in.t_start);
    return SMAFA_OK;
}

static int cluster_run(const ClusterInput &in, uint32_t max_divergence, int out_fd, int device, int alphabet,
                       uint32_t rank, uint32_t world, smafa_allgather_fn allgather, void *ctx) {
    if (world == 0 || rank >= world) return set_error(SMAFA_ERR_INVALID, "rank %u outside world of %u", rank, world);
    if (world > 1 && !allgather) return set_error(SMAFA_ERR_INVALID, "smafa_cluster_sharded: NULL allgather");
    const double t_start = in.t_start;
    int rc = SMAFA_OK;
    const std::vector<uint8_t> &raw = in.recs.raw, &codes = in.recs.codes;
    const std::vector<uint32_t> &uniq = in.uniq;
    const uint64_t n = in.recs.n;
    const size_t L = in.recs.L;
    const int pending = in.pending;
    const std::string &pending_msg = in.pending_msg;
    if (n > 0) {
        const uint32_t Lw = (uint32_t)L;
        std::vector<uint32_t> centroid_of(n, UINT32_MAX);  // record -> centroid ordinal
        std::vector<uint32_t> centroid_rec;                // centroid ordinal -> record

        DbGuard centroids, cand;
        rc = smafa_db_create(&centroids.db, device, alphabet, Lw);
        if (rc) return rc;

        // ---- output stage, overlapped with the scans (src/cluster.rs:79-84).  A record's line is final as soon as the batch
        // that holds it is resolved (records are resolved in input order; duplicates print nothing), so rank 0 formats and
        // writes the lines of finished batches on a writer thread while the next batches scan.  Every line is
        // "raw record \t centroid string \n" = 2L + 2 bytes; blocks of lines are formatted by all threads and written in order.
        centroid_rec.reserve(uniq.size());  // the writer reads entries while the main loop appends: never reallocated
        struct Progress {
            std::mutex m;
            std::condition_variable cv;
            uint64_t final_upto = 0;  // records [0, final_upto) have their final centroid_of / centroid_rec entries
            bool stop = false;        // the main loop failed: write nothing more
        } prog;
        int write_rc = SMAFA_OK;
        std::string write_msg;
        size_t lines_written = 0;
        double t_write_busy = 0.0;
        auto writer_body = [&]() noexcept {
            try {
                char letters[32];
                for (int c = 0; c < 32; c++) letters[c] = letter_of(alphabet, (uint8_t)c);
                const size_t line_bytes = 2 * L + 2, block_records = 1u << 20;
                const unsigned T = n >= (1u << 16) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
                std::vector<char> text;
                std::vector<uint32_t> lines;
                uint64_t done = 0;
                while (done < n) {
                    uint64_t upto;
                    {
                        std::unique_lock<std::mutex> lock(prog.m);
                        prog.cv.wait(lock, [&] { return prog.stop || prog.final_upto > done; });
                        if (prog.stop) return;
                        upto = std::min<uint64_t>(prog.final_upto, done + block_records);
                    }
                    const double t0 = now_seconds();
                    lines.clear();
                    for (uint64_t i = done; i < upto; i++)
                        if (centroid_of[i] != UINT32_MAX) lines.push_back((uint32_t)i);
                    const size_t nk = lines.size();
                    if (text.size() < nk * line_bytes) text.resize(nk * line_bytes);
                    auto fill = [&](unsigned t, unsigned of) {
                        for (size_t k = nk * t / of, e = nk * (t + 1) / of; k < e; k++) {
                            const uint32_t i = lines[k];
                            char *o = &text[k * line_bytes];
                            memcpy(o, &raw[(size_t)i * L], L);
                            o[L] = '\t';
                            const uint8_t *c = &codes[(size_t)centroid_rec[centroid_of[i]] * L];
                            for (size_t j = 0; j < L; j++) o[L + 1 + j] = letters[c[j] & 31];
                            o[2 * L + 1] = '\n';
                        }
                    };
                    const unsigned use = nk >= (1u << 14) ? T : 1u;
                    if (use == 1) {
                        fill(0, 1);
                    } else {
                        std::vector<std::thread> pool;
                        pool.reserve(use);
                        for (unsigned t = 0; t < use; t++) pool.emplace_back(fill, t, use);
                        for (auto &th : pool) th.join();
                    }
                    if (nk) {
                        write_rc = write_all(out_fd, text.data(), nk * line_bytes);
                        if (write_rc) {
                            write_msg = smafa_last_error();
                            return;
                        }
                    }
                    lines_written += nk;
                    done = upto;
                    t_write_busy += now_seconds() - t0;
                }
            } catch (...) {
                write_rc = smafa::exception_code("the cluster output thread");
                try {
                    write_msg = smafa_last_error();
                } catch (...) {
                }
            }
        };
        std::thread writer;
        struct WriterGuard {  // whatever way the function is left: the writer is told to stop and joined
            Progress &p;
            std::thread &th;
            ~WriterGuard() {
                if (!th.joinable()) return;
                {
                    std::lock_guard<std::mutex> lock(p.m);
                    p.stop = true;
                }
                p.cv.notify_all();
                th.join();
            }
        } writer_guard{prog, writer};
        if (rank == 0) writer = std::thread(writer_body);
        auto publish = [&](uint64_t upto) {
            {
                std::lock_guard<std::mutex> lock(prog.m);
                prog.final_upto = upto;
            }
            prog.cv.notify_all();
        };

        std::vector<uint8_t> batch_codes, cand_codes, new_codes;
        std::vector<smafa_hit> old_hits, cand_hits;
        std::vector<uint32_t> cand_pos;        // candidate ordinal -> position in the batch
        std::vector<uint32_t> cand_centroid;   // candidate ordinal -> centroid ordinal it became (or NONE)
        size_t pos = 0, B = 1024, n_batches = 0;
        // largest batch: a batch costs two scans + an append whatever its size (~2 ms of host round trips), the scans themselves
        // grow with it; SMAFA_CLUSTER_BATCH overrides (every rank reads the same environment)
        size_t batch_cap = 65536;
        if (const char *bc = getenv("SMAFA_CLUSTER_BATCH")) batch_cap = std::max<size_t>(256, strtoull(bc, nullptr, 10));
        double t_old = 0, t_cand = 0, t_seq = 0, t_append = 0;
        while (pos < uniq.size()) {
            n_batches++;
            double t0 = now_seconds();
            const size_t nb = std::min(B, uniq.size() - pos);
            batch_codes.resize(nb * L);
            for (size_t b = 0; b < nb; b++) memcpy(&batch_codes[b * L], &codes[(size_t)uniq[pos + b] * L], L);
            // this rank's slice of the batch; slices are contiguous and in rank order
            const size_t s_lo = nb * rank / world, s_hi = nb * (rank + 1) / world, ns = s_hi - s_lo;

            // 1. nearest old centroid per record: rows ordered (record, distance, centroid), minimum only
            old_hits.clear();
            if (!centroid_rec.empty() && ns) {
                rc = scan_to_host(centroids.db, batch_codes.data() + s_lo * L, ns, max_divergence, 1, old_hits);
                if (rc) return rc;
            }
            std::vector<uint32_t> old_d(nb, UINT32_MAX), old_c(nb, UINT32_MAX);
            for (size_t t = old_hits.size(); t-- > 0;) {  // backwards: the first row of each record wins
                old_d[s_lo + old_hits[t].query] = old_hits[t].dist;
                old_c[s_lo + old_hits[t].query] = old_hits[t].subject;
            }
            if (world > 1) {  // exchange (distance, centroid) of every record: nb x 8 bytes
                std::vector<uint32_t> mine(2 * ns);
                for (size_t b = 0; b < ns; b++) {
                    mine[2 * b] = old_d[s_lo + b];
                    mine[2 * b + 1] = old_c[s_lo + b];
                }
                const void *all = nullptr;
                uint64_t all_bytes = 0;
                if (allgather(ctx, mine.data(), mine.size() * 4, &all, &all_bytes) != 0)
                    return set_error(SMAFA_ERR_IO, "allgather failed (nearest old centroids)");
                if (all_bytes != nb * 8 || (nb && !all))
                    return set_error(SMAFA_ERR_INVALID, "allgather returned %llu bytes, expected %zu",
                                     (unsigned long long)all_bytes, nb * 8);
                const uint32_t *w = (const uint32_t *)all;
                for (size_t b = 0; b < nb; b++) {
                    old_d[b] = w[2 * b];
                    old_c[b] = w[2 * b + 1];
                }
            }
            t_old += now_seconds() - t0;
            t0 = now_seconds();

            // 2. candidates = records with no old centroid in range
            cand_pos.clear();
            cand_codes.clear();
            for (size_t b = 0; b < nb; b++)
                if (old_c[b] == UINT32_MAX) {
                    cand_pos.push_back((uint32_t)b);
                    cand_codes.insert(cand_codes.end(), &batch_codes[b * L], &batch_codes[b * L] + L);
                }
            cand_hits.clear();
            if (!cand_pos.empty()) {
                if (!cand.db) {  // one handle for every batch: its device buffers are reused
                    rc = smafa_db_create(&cand.db, device, alphabet, Lw);
                    if (rc) return rc;
                }
                rc = db_clear(cand.db);
                if (rc) return rc;
                rc = smafa_db_append(cand.db, cand_codes.data(), cand_pos.size());
                if (rc) return rc;
                if (ns) {
                    rc = scan_to_host(cand.db, batch_codes.data() + s_lo * L, ns, max_divergence, SMAFA_NONE, cand_hits);
                    if (rc) return rc;
                }
                for (smafa_hit &hrow : cand_hits) hrow.query += (uint32_t)s_lo;  // position in the batch
                if (world > 1) {  // rows of slice r precede rows of slice r+1: the concatenation stays ordered
                    const void *all = nullptr;
                    uint64_t all_bytes = 0;
                    if (allgather(ctx, cand_hits.data(), cand_hits.size() * sizeof(smafa_hit), &all, &all_bytes) != 0)
                        return set_error(SMAFA_ERR_IO, "allgather failed (candidate rows)");
                    if (all_bytes % sizeof(smafa_hit) || (all_bytes && !all))
                        return set_error(SMAFA_ERR_INVALID, "allgather returned %llu bytes, not whole rows",
                                         (unsigned long long)all_bytes);
                    const smafa_hit *rows = (const smafa_hit *)all;
                    cand_hits.assign(rows, rows + all_bytes / sizeof(smafa_hit));
                    for (const smafa_hit &hrow : cand_hits)
                        if (hrow.query >= nb || hrow.subject >= cand_pos.size())
                            return set_error(SMAFA_ERR_INVALID, "allgather returned a row outside the batch");
                }
            }

            t_cand += now_seconds() - t0;
            t0 = now_seconds();
            // 3. sequential pass in input order
            cand_centroid.assign(cand_pos.size(), UINT32_MAX);
            new_codes.clear();
            size_t h = 0, next_cand = 0;
            for (size_t b = 0; b < nb; b++) {
                uint32_t best_d = old_d[b], best_c = old_c[b];
                // rows of this record are ordered (distance, candidate); candidates that became centroids
                // are numbered in batch order, so the first admissible row is the best new centroid
                while (h < cand_hits.size() && cand_hits[h].query < b) h++;
                for (size_t t = h; t < cand_hits.size() && cand_hits[t].query == b; t++) {
                    const uint32_t c = cand_hits[t].subject;
                    if (cand_pos[c] >= b || cand_centroid[c] == UINT32_MAX) continue;  // later record / not a centroid
                    if (cand_hits[t].dist < best_d) {  // strict: an old centroid wins ties (lower index)
                        best_d = cand_hits[t].dist;
                        best_c = cand_centroid[c];
                    }
                    break;
                }
                const uint32_t record = uniq[pos + b];
                if (best_c != UINT32_MAX && best_d <= max_divergence) {  // src/cluster.rs:62-68
                    centroid_of[record] = best_c;
                } else {  // src/cluster.rs:69-74
                    while (cand_pos[next_cand] != b) next_cand++;
                    cand_centroid[next_cand] = (uint32_t)centroid_rec.size();
                    centroid_of[record] = (uint32_t)centroid_rec.size();
                    centroid_rec.push_back(record);
                    new_codes.insert(new_codes.end(), &batch_codes[b * L], &batch_codes[b * L] + L);
                }
            }
            t_seq += now_seconds() - t0;
            t0 = now_seconds();
            if (!new_codes.empty()) {
                rc = smafa_db_append(centroids.db, new_codes.data(), new_codes.size() / L);
                if (rc) return rc;
            }
            t_append += now_seconds() - t0;
            pos += nb;
            // every record in front of the next unseen one is final now (the records between two first occurrences are duplicates)
            if (rank == 0) publish(pos < uniq.size() ? uniq[pos] : n);
            // batch size follows the row volume: grow while the scans stay cheap, shrink on dense input
            // (from exchanged quantities only, so every rank takes the same decision)
            const size_t volume = (nb - cand_pos.size()) + cand_hits.size();
            if (volume < (4u << 20) && B < batch_cap) B *= 2;
            else if (volume > (16u << 20) && B > 256) B /= 2;
        }

        log_line(2, "%zu batches: scans vs old centroids %.2f s, candidate scans %.2f s, sequential pass %.2f s, "
                    "centroid appends %.2f s", n_batches, t_old, t_cand, t_seq, t_append);
        {
            double ms_a = 0, ms_b = 0;
            uint64_t la = 0, lb = 0;
            db_life_stats(centroids.db, &ms_a, &la);
            db_life_stats(cand.db, &ms_b, &lb);
            log_line(2, "scan kernels %.1f ms over %llu launches (vs old centroids %.1f ms, vs candidates %.1f ms)", ms_a + ms_b,
                     (unsigned long long)(la + lb), ms_a, ms_b);
            // where this rank's launches really went (a multi-device run must not end up with every replica on device 0)
            int at = -1, at_c = -1;
            uint64_t off = 0, off_c = 0;
            smafa_launch_device(centroids.db, &at, &off);
            if (cand.db) smafa_launch_device(cand.db, &at_c, &off_c);
            log_line(2, "rank %u of %u: handle on device %d, scan launches issued with device %d current, %llu off the handle's device",
                     rank, world, device, at, (unsigned long long)(off + off_c));
        }
        if (rank == 0) {  // the writer finishes the lines of the last batches
            const double t_out = now_seconds();
            publish(n);
            writer.join();
            if (write_rc) return set_error(write_rc, "%s", write_msg.c_str());
            log_line(2, "%zu lines written in %.2f s (%.2f s of formatting + writing overlapped with the scans, %.2f s after the last batch)",
                     lines_written, t_write_busy, t_write_busy - std::min(t_write_busy, now_seconds() - t_out), now_seconds() - t_out);
        }
        log_line(2, "cluster: %.3f s from the start of the load to the last line", now_seconds() - t_start);
        log_line(1, "Clustering complete, took %llu seconds. Clustered %llu sequences into %zu clusters.",  // src/cluster.rs:87-92
                 (unsigned long long)(now_seconds() - t_start), (unsigned long long)n, centroid_rec.size());
    }
    if (pending != SMAFA_OK) return set_error(pending, "%s", pending_msg.c_str());
    return SMAFA_OK;
}

int smafa_cluster(const char *input_fasta, uint32_t max_divergence, int out_fd, int device, int alphabet) try {
    ClusterInput in;
    int rc = cluster_load(input_fasta, alphabet, device, in);
    if (rc) return rc;
    return cluster_run(in, max_divergence, out_fd, device, alphabet, 0, 1, nullptr, nullptr);
} catch (...) {
    return smafa::exception_code("smafa_cluster");
}

int smafa_cluster_sharded(const char *input_fasta, uint32_t max_divergence, int out_fd, int device, int alphabet,
                          uint32_t rank, uint32_t world, smafa_allgather_fn allgather, void *ctx) try {
    if (world == 0 || rank >= world) return set_error(SMAFA_ERR_INVALID, "rank %u outside world of %u", rank, world);
    if (world > 1 && !allgather) return set_error(SMAFA_ERR_INVALID, "smafa_cluster_sharded: NULL allgather");
    ClusterInput in;
    int rc = cluster_load(input_fasta, alphabet, device, in);
    if (rc) return rc;
    return cluster_run(in, max_divergence, out_fd, device, alphabet, rank, world, allgather, ctx);
} catch (...) {
    return smafa::exception_code("smafa_cluster_sharded");
}

// ---- the same clustering over several GPUs of one node by ONE process: one host thread per entry of `devices` plays a
// rank of smafa_cluster_sharded (its own replica of the centroid store on its own device, its slice of every batch), the
// input is parsed and de-duplicated once for all of them, and the two exchanges per batch go through memory.
namespace {
struct ThreadExchange {
    std::mutex m;
    std::condition_variable cv;
    uint32_t world = 0, arrived = 0;
    uint64_t round = 0;
    bool failed = false;
    std::vector<std::vector<uint8_t>> parts;
    std::vector<uint8_t> all[2];  // the result of round r lives in all[r & 1] until round r + 2 overwrites it
};
struct ThreadRank {
    ThreadExchange *ex;
    uint32_t rank;
};
int thread_allgather(void *ctx, const void *send, uint64_t send_bytes, const void **recv, uint64_t *recv_bytes) {
    ThreadRank *me = (ThreadRank *)ctx;
    ThreadExchange &ex = *me->ex;
    std::unique_lock<std::mutex> lock(ex.m);
    if (ex.failed) return 1;
    ex.parts[me->rank].assign((const uint8_t *)send, (const uint8_t *)send + send_bytes);
    const uint64_t my_round = ex.round;
    if (++ex.arrived == ex.world) {
        std::vector<uint8_t> &out = ex.all[my_round & 1];
        out.clear();
        for (const auto &p : ex.parts) out.insert(out.end(), p.begin(), p.end());
        ex.arrived = 0;
        ex.round++;
        ex.cv.notify_all();
    } else {
        ex.cv.wait(lock, [&] { return ex.round != my_round || ex.failed; });
        if (ex.round == my_round) return 1;  // somebody failed before the round completed
    }
    *recv = ex.all[my_round & 1].empty() ? nullptr : ex.all[my_round & 1].data();
    *recv_bytes = ex.all[my_round & 1].size();
    return 0;
}
}  // namespace

int smafa_cluster_multi(const char *input_fasta, uint32_t max_divergence, int out_fd, const int *devices, int ndev, int alphabet) try {
    if (!devices || ndev < 1 || ndev > 64) return set_error(SMAFA_ERR_INVALID, "smafa_cluster_multi: 1 to 64 devices expected");
    ClusterInput in;
    int rc = cluster_load(input_fasta, alphabet, devices[0], in);
    if (rc) return rc;
    if (ndev == 1) return cluster_run(in, max_divergence, out_fd, devices[0], alphabet, 0, 1, nullptr, nullptr);
    ThreadExchange ex;
    ex.world = (uint32_t)ndev;
    ex.parts.resize((size_t)ndev);
    std::vector<int> rcs((size_t)ndev, SMAFA_OK);
    std::vector<std::string> msgs((size_t)ndev);
    std::vector<ThreadRank> ranks((size_t)ndev);
    std::vector<std::thread> pool;
    auto leave = [&] {  // nobody waits for a rank that has left
        std::lock_guard<std::mutex> lock(ex.m);
        ex.failed = true;
        ex.cv.notify_all();
    };
    int start_rc = SMAFA_OK;
    try {
        pool.reserve((size_t)ndev);
        for (int r = 0; r < ndev; r++) {
            ranks[r] = {&ex, (uint32_t)r};
            pool.emplace_back([&, r]() noexcept {
                try {
                    rcs[r] = cluster_run(in, max_divergence, out_fd, devices[r], alphabet, (uint32_t)r, (uint32_t)ndev, thread_allgather, &ranks[r]);
                } catch (...) {  // no exception ends a rank's thread (std::terminate): it becomes the rank's error code
                    rcs[r] = smafa::exception_code("a cluster rank's thread");
                }
                if (rcs[r]) {
                    try {
                        msgs[r] = smafa_last_error();  // the text is per thread
                    } catch (...) {
                    }
                    leave();
                }
            });
        }
    } catch (...) {  // a thread could not be started: the ranks already running must not wait for it, and are joined
        start_rc = smafa::exception_code("starting a cluster rank's thread");
        leave();
    }
    for (auto &th : pool) th.join();
    if (start_rc) return start_rc;
    // every rank reports the same input-borne failure; a rank-local one (a device) is the first one's to tell
    for (int r = 0; r < ndev; r++)
        if (rcs[r] && msgs[r].find("allgather failed") == std::string::npos) return set_error(rcs[r], "%s", msgs[r].c_str());
    for (int r = 0; r < ndev; r++)
        if (rcs[r]) return set_error(rcs[r], "%s", msgs[r].c_str());
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_cluster_multi");
}

// ---------------------------------------------------------------------------------- write_rows
int smafa_write_rows(const smafa_hit *rows, uint64_t n_rows, const uint8_t *subject_codes, uint64_t n_subjects,
                     uint32_t seq_len, int alphabet, uint32_t query_offset, int out_fd) try {
    if ((!rows && n_rows) || (!subject_codes && n_rows)) return set_error(SMAFA_ERR_INVALID, "smafa_write_rows: NULL argument");
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    for (uint64_t i = 0; i < n_rows; i++)
        if (rows[i].subject >= n_subjects)
            return set_error(SMAFA_ERR_INVALID, "row %llu names subject %u of %llu", (unsigned long long)i, rows[i].subject,
                             (unsigned long long)n_subjects);
    SubjectRows subjects;
    subjects.codes = subject_codes;
    subjects.L = seq_len;
    return write_rows_text(rows, n_rows, subjects, alphabet, query_offset, out_fd);
} catch (...) {
    return smafa::exception_code("smafa_write_rows");
}

// -------------------------------------------------------------------------------------- count
int smafa_count(const char *const *paths, uint64_t n_paths, int out_fd) try {
    std::string text = "[";
    for (uint64_t i = 0; i < n_paths; i++) {
        FastxReader reader;
        int rc = reader.open(paths[i]);
        if (rc) return rc;
        uint64_t reads = 0, bases = 0;
        FastxRecord rec;
        while ((rc = reader.next(rec)) == 1) {
            reads++;
            bases += rec.seq_len;
        }
        if (rc < 0) return rc;
        if (i) text.push_back(',');
        text += "{\"path\":\"";
        for (const char *c = paths[i]; *c; c++) {
            if (*c == '"' || *c == '\\') text.push_back('\\');
            text.push_back(*c);
        }
        text += "\",\"num_reads\":" + std::to_string(reads) + ",\"num_bases\":" + std::to_string(bases) + "}";
    }
    text += "]\n";
    return write_all(out_fd, text.data(), text.size());
} catch (...) {
    return smafa::exception_code("smafa_count");
}

}  // extern "C"
