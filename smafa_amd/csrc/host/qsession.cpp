// qsession.cpp — `smafa query` for hosts that run ONE PROCESS PER GPU (smafa_amd/dist.py over torch.distributed / RCCL; an
// MPI host would call the same three functions).  The reference's query loop carries no state between records but the
// running query number (/root/reference/src/lib.rs:232-318), so the query FILE is cut into `parts` contiguous shares in
// rank order; a process
//   * opens the DB once (smafa_qsession_open): a packed store file is mapped and copied to HBM — no decode, no host code
//     rows; a version-2 file is decoded by all threads and packed on the device;
//   * answers its share (smafa_qsession_scan_part): it reads ONLY its byte range of the query file (host/fastx.cpp
//     load_records_part), scans, applies the selection rules of src/lib.rs:241-315 and returns rows numbered from 0 within
//     the share, plus the share's record count — the caller adds the counts of the ranks in front (one tiny all-gather) and
//     gathers the rows on rank 0;
//   * rank 0 prints (smafa_qsession_write): subject strings are decoded from the mapped planes for the hit rows only.
// Errors keep the reference's order: the records in front of a bad one are answered, the shares behind it are not printed
// (the caller drops the rows of the ranks after the first one that reports *pending != 0).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../engine.h"
#include "fastx.h"
#include "packed.h"

struct smafa_qsession {
    smafa_db *db = nullptr;
    smafa::PackedStore pk;
    bool packed = false;
    uint8_t *codes = nullptr;  // version-2 file: the decoded rows (strings for the output, limit-per-sequence)
    uint64_t n = 0;
    uint32_t L = 0;
    int alphabet = 0;
    smafa::SubjectRows subjects;
};

using namespace smafa;

extern "C" {

int smafa_fastx_load_part(const char *path, int alphabet, uint32_t part, uint32_t parts, uint8_t **codes_out, uint64_t *n_out,
                          uint32_t *seq_len, int *pending, int *usable) try {
    if (!path || !codes_out || !n_out || !seq_len || !pending || !usable)
        return set_error(SMAFA_ERR_INVALID, "smafa_fastx_load_part: NULL argument");
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA) return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    *codes_out = nullptr;
    *n_out = 0;
    *seq_len = 0;
    *pending = SMAFA_OK;
    *usable = 0;
    BulkRecords recs;
    bool ok = false;
    int rc = load_records_part(path, alphabet, part, parts, recs, &ok);
    if (rc) return rc;
    if (!ok) return SMAFA_OK;
    *usable = 1;
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
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_fastx_load_part");
}

int smafa_qsession_open(smafa_qsession **out, const char *db_path, int device) try {
    if (!out || !db_path) return set_error(SMAFA_ERR_INVALID, "smafa_qsession_open: NULL argument");
    *out = nullptr;
    struct Closer {
        void operator()(smafa_qsession *q) const { smafa_qsession_close(q); }
    };
    std::unique_ptr<smafa_qsession, Closer> owner(new smafa_qsession());  // closed on every early return and on an exception
    smafa_qsession *s = owner.get();
    auto fail = [&](int rc) { return rc; };
    uint8_t head[8] = {0};
    if (FILE *f = fopen(db_path, "rb")) {
        const size_t got = fread(head, 1, sizeof head, f);
        fclose(f);
        s->packed = is_packed_file(head, got);
    }
    int rc;
    if (s->packed) {
        rc = s->pk.open(db_path);
        if (rc) return fail(rc);
        s->alphabet = (int)s->pk.h.alphabet;
        s->n = s->pk.h.n;
        s->L = s->pk.h.seq_len;
        if (s->n && device >= 0) {
            rc = db_load_packed(&s->db, device, s->pk);
            if (rc) return fail(rc);
        }
    } else {
        rc = smafa_dbfile_read(db_path, &s->alphabet, &s->codes, &s->n, &s->L);  // src/lib.rs:208-218
        if (rc) return fail(rc);
        if (s->n && device >= 0) {
            rc = smafa_db_create(&s->db, device, s->alphabet, s->L);
            if (!rc) rc = smafa_db_append(s->db, s->codes, s->n);
            if (rc) return fail(rc);
        }
    }
    if (s->db && !getenv("SMAFA_INDEX")) (void)smafa_set_index(s->db, 3);  // (as smafa_query_multi: rent or buy)
    s->subjects.codes = s->codes;
    s->subjects.packed = s->packed ? &s->pk : nullptr;
    s->subjects.L = s->L;
    *out = owner.release();
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_qsession_open");
}

int smafa_qsession_info(const smafa_qsession *s, uint64_t *n_subjects, uint32_t *seq_len, int *alphabet) try {
    if (!s) return set_error(SMAFA_ERR_INVALID, "smafa_qsession_info: NULL session");
    if (n_subjects) *n_subjects = s->n;
    if (seq_len) *seq_len = s->L;
    if (alphabet) *alphabet = s->alphabet;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_qsession_info");
}

int smafa_qsession_scan_part(smafa_qsession *s, const char *query_fasta, uint32_t max_divergence, uint32_t max_num_hits,
                             uint32_t limit_per_sequence, uint32_t part, uint32_t parts, int whole_file, smafa_hit **rows_out,
                             uint64_t *n_rows, uint64_t *n_queries, uint64_t *n_before, int *pending, int *retry_whole) try {
    if (!s || !query_fasta || !rows_out || !n_rows || !n_queries || !n_before || !pending || !retry_whole)
        return set_error(SMAFA_ERR_INVALID, "smafa_qsession_scan_part: NULL argument");
    if (parts == 0 || part >= parts) return set_error(SMAFA_ERR_INVALID, "part %u of %u", part, parts);
    *rows_out = nullptr;
    *n_rows = *n_queries = *n_before = 0;
    *pending = SMAFA_OK;
    *retry_whole = 0;
    const uint32_t L = s->L;
    BulkRecords recs;
    uint64_t lo = 0, count = 0;  // this share = records [lo, lo + count) of `recs`
    std::string pending_msg;
    auto length_panic = [&](size_t len) {  // src/lib.rs:72-79
        char msg[160];
        snprintf(msg, sizeof msg, "Cannot compute distances between seq of length %zu and windows of lengths %u", len, L);
        *pending = SMAFA_ERR_PANIC;
        pending_msg = msg;
    };
    int rc;
    if (!whole_file) {
        bool usable = false;
        rc = load_records_part(query_fasta, s->alphabet, part, parts, recs, &usable);
        if (rc) return rc;
        if (!usable) {
            *retry_whole = 1;  // every rank must take the whole file instead (the caller agrees on that collectively)
            return SMAFA_OK;
        }
        count = recs.n;
        *n_before = UINT64_MAX;  // known only once the counts of the ranks in front are
    } else {
        rc = load_records_bulk(query_fasta, s->alphabet, false, recs);
        if (rc) return expect_fastx(rc, "valid path/file of query fasta");  // src/lib.rs:221
    }
    // what stops the share: the first record the reference's loop fails on (src/lib.rs:234-238)
    uint64_t usable_n = recs.n;
    if (s->n > 0 && recs.n > 0 && recs.L != L) {  // the share's first record already fails the length check
        usable_n = 0;
        length_panic(recs.L);
    } else if (recs.err_kind == 1) {
        *pending = SMAFA_ERR_PANIC;
        pending_msg = recs.err_msg;
    } else if (recs.err_kind == 2) {
        // the offender differs from the share's first record; if THAT one differed from the store, the branch above took it
        length_panic(recs.err_len);
    } else if (recs.err_kind == 3) {
        length_panic(0);
    } else if (recs.err_kind == 4) {
        *pending = SMAFA_ERR_PANIC;  // record.expect(..), src/lib.rs:234
        pending_msg = "Failed to parse query sequence: " + recs.err_msg;
    }
    if (!whole_file) {
        lo = 0;
        count = usable_n;
    } else {  // the whole file was parsed: this rank's block of the usable records, by count
        lo = usable_n * (uint64_t)part / (uint64_t)parts;
        count = usable_n * (uint64_t)(part + 1) / (uint64_t)parts - lo;
        *n_before = lo;
    }
    if (count > 0xfffffff0ull) return set_error(SMAFA_ERR_INVALID, "too many queries");
    *n_queries = count;

    const bool kmode = max_num_hits != SMAFA_NONE && max_num_hits != 1;  // src/lib.rs:224
    const uint32_t dev_k = !kmode ? 1u : (max_num_hits == 0 || max_num_hits > (uint32_t)s->n) ? SMAFA_NONE : max_num_hits;
    std::vector<smafa_hit> hits, rows;
    if (count > 0) {
        if (s->n > 0) {
            if (!s->db) return set_error(SMAFA_ERR_DEVICE, "this session was opened without a device");
            // chunks bound the device's row list when nothing else does (no bound and no k: every pair is a row)
            const bool unbounded = max_divergence == SMAFA_NONE && dev_k == SMAFA_NONE;
            const uint64_t chunk = unbounded ? std::max<uint64_t>(1, (16ull << 20) / std::max<uint64_t>(s->n, 1)) : (1ull << 20);
            std::vector<smafa_hit> part_hits;
            for (uint64_t off = 0; off < count; off += chunk) {
                const uint64_t c = std::min<uint64_t>(chunk, count - off);
                rc = scan_to_host(s->db, recs.codes.data() + (size_t)(lo + off) * L, c, max_divergence, dev_k, part_hits);
                if (rc) return rc;
                for (smafa_hit &h : part_hits) h.query += (uint32_t)off;
                hits.insert(hits.end(), part_hits.begin(), part_hits.end());
            }
        }
        // (an empty store, k = 0, --limit-per-sequence without k > 1: the reference's own panics, worded by select_rows)
        rc = select_rows(hits.data(), hits.size(), count, s->n, s->subjects, max_divergence, max_num_hits, limit_per_sequence, rows);
        if (rc) return rc;
    }
    if (!rows.empty()) {
        smafa_hit *p = (smafa_hit *)malloc(rows.size() * sizeof(smafa_hit));
        if (!p) return set_error(SMAFA_ERR_NOMEM, "out of host memory");
        memcpy(p, rows.data(), rows.size() * sizeof(smafa_hit));
        *rows_out = p;
        *n_rows = rows.size();
    }
    if (*pending != SMAFA_OK) set_error(*pending, "%s", pending_msg.c_str());  // the text for the caller to relay
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_qsession_scan_part");
}

int smafa_qsession_write(smafa_qsession *s, const smafa_hit *rows, uint64_t n_rows, int out_fd) try {
    if (!s || (!rows && n_rows)) return set_error(SMAFA_ERR_INVALID, "smafa_qsession_write: NULL argument");
    for (uint64_t i = 0; i < n_rows; i++)
        if (rows[i].subject >= s->n)
            return set_error(SMAFA_ERR_INVALID, "row %llu names subject %u of %llu", (unsigned long long)i, rows[i].subject,
                             (unsigned long long)s->n);
    return write_rows_text(rows, n_rows, s->subjects, s->alphabet, 0, out_fd);  // src/lib.rs:292,310
} catch (...) {
    return smafa::exception_code("smafa_qsession_write");
}

void smafa_qsession_close(smafa_qsession *s) {
    if (!s) return;
    smafa_db_destroy(s->db);
    smafa_free(s->codes);
    delete s;
}

}  // extern "C"
