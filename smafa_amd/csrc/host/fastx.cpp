// fastx.cpp — see fastx.h.  Behaviour the reference relies on and its fixtures pin:
//  * format by first byte: '>' FASTA, '@' FASTQ; anything else (or an empty file) is an error —
//    the reference .expect()s a valid file (src/lib.rs:144,221);
//  * gzip input is decoded (tests/data/random_30_4.fq.gz: 4 reads, 120 bases, test_cmdline.rs:194-201);
//    bzip2/xz are refused here (no headers in this image);
//  * multi-line FASTA is joined, CR/LF dropped; the last record needs no trailing newline
//    (tests/data/subjects.fa); FASTQ is the 4-line form.
#include "fastx.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>
#include <cstdio>
#include <cstring>

#include "../engine.h"

namespace smafa {

int FastxReader::open(const char *path) {
    path_ = path;
    FILE *f = fopen(path, "rb");
    if (!f) return set_error(SMAFA_ERR_IO, "valid path/file expected: %s: %s", path, strerror(errno));
    unsigned char magic[6] = {0};
    const size_t got = fread(magic, 1, sizeof magic, f);
    fclose(f);
    if (got >= 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h')
        return set_error(SMAFA_ERR_FORMAT, "%s: bzip2 input is not supported by this build", path);
    if (got >= 6 && magic[0] == 0xfd && magic[1] == '7' && magic[2] == 'z' && magic[3] == 'X' && magic[4] == 'Z')
        return set_error(SMAFA_ERR_FORMAT, "%s: xz input is not supported by this build", path);
    data_.clear();
    if (map_) munmap(map_, size_);
    map_ = nullptr;
    base_ = nullptr;
    size_ = 0;
    const bool gz = got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (!gz) {  // plain file: mapped when it is a regular file, otherwise read to the end
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return set_error(SMAFA_ERR_IO, "%s: cannot open", path);
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                map_ = m;
                base_ = (const uint8_t *)m;
                size_ = (size_t)st.st_size;
            }
        }
        if (!map_) {
            std::vector<uint8_t> more(1u << 20);
            for (;;) {
                const ssize_t r = read(fd, more.data(), more.size());
                if (r < 0) {
                    if (errno == EINTR) continue;
                    close(fd);
                    return set_error(SMAFA_ERR_IO, "%s: read error", path);
                }
                if (r == 0) break;
                data_.insert(data_.end(), more.begin(), more.begin() + r);
            }
        }
        close(fd);
    } else {
        gzFile g = gzopen(path, "rb");
        if (!g) return set_error(SMAFA_ERR_IO, "%s: cannot open", path);
        gzbuffer(g, 1u << 20);
        std::vector<uint8_t> chunk(8u << 20);
        for (;;) {
            const int r = gzread(g, chunk.data(), (unsigned)chunk.size());
            if (r < 0) {
                gzclose(g);
                return set_error(SMAFA_ERR_IO, "%s: read error", path);
            }
            if (r == 0) break;
            data_.insert(data_.end(), chunk.begin(), chunk.begin() + r);
        }
        gzclose(g);
    }
    if (!map_) {
        base_ = data_.data();
        size_ = data_.size();
    }
    if (size_ == 0) return set_error(SMAFA_ERR_FORMAT, "valid path/file expected: %s is empty", path);
    if (base_[0] != '>' && base_[0] != '@')
        return set_error(SMAFA_ERR_FORMAT, "valid path/file expected: %s does not start with '>' or '@'", path);
    fastq_ = base_[0] == '@';
    pos_ = 0;
    return SMAFA_OK;
}

FastxReader::~FastxReader() {
    if (map_) munmap(map_, size_);
}

int FastxReader::next(FastxRecord &rec) {
    const uint8_t *d = base_;
    const size_t n = size_;
    auto eol = [&](size_t from) {
        const void *p = from < n ? memchr(d + from, '\n', n - from) : nullptr;
        return p ? (size_t)((const uint8_t *)p - d) : n;
    };
    auto after = [&](size_t line_end) { return line_end < n ? line_end + 1 : n; };
    while (pos_ < n && (d[pos_] == '\n' || d[pos_] == '\r')) pos_++;
    if (pos_ >= n) return 0;
    if (d[pos_] != (fastq_ ? '@' : '>'))
        return set_error(SMAFA_ERR_FORMAT, "Failed to parse sequence: %s: record does not start with '%c'", path_.c_str(),
                         fastq_ ? '@' : '>');
    const size_t hs = pos_ + 1, he = eol(hs);
    rec.id = d + hs;
    rec.id_len = he - hs;
    if (rec.id_len && rec.id[rec.id_len - 1] == '\r') rec.id_len--;
    size_t p = after(he);
    seq_.clear();
    // a line goes in with one bulk copy; carriage returns (rare) are squeezed out afterwards
    auto take_line = [&](size_t from, size_t to) {
        const size_t base = seq_.size();
        seq_.insert(seq_.end(), d + from, d + to);
        if (to > from && memchr(d + from, '\r', to - from)) {
            size_t w = base;
            for (size_t i = base; i < seq_.size(); i++)
                if (seq_[i] != '\r') seq_[w++] = seq_[i];
            seq_.resize(w);
        }
    };
    if (!fastq_) {
        const size_t le0 = eol(p);
        const size_t p2 = after(le0);
        if ((p2 >= n || d[p2] == '>') && !(le0 > p && memchr(d + p, '\r', le0 - p))) {
            // the common case — one sequence line, no CR: hand out the bytes in place, no copy
            rec.seq = d + p;
            rec.seq_len = le0 - p;
            pos_ = p2;
            return 1;
        }
        while (p < n && d[p] != '>') {
            const size_t le = eol(p);
            take_line(p, le);
            p = after(le);
        }
    } else {
        size_t le = eol(p);
        take_line(p, le);
        p = after(le);
        if (p >= n || d[p] != '+')
            return set_error(SMAFA_ERR_FORMAT, "Failed to parse sequence: %s: FASTQ record without '+' line", path_.c_str());
        p = after(eol(p));
        le = eol(p);
        size_t qlen = le - p;
        if (qlen && d[p + qlen - 1] == '\r') qlen--;
        if (qlen != seq_.size())
            return set_error(SMAFA_ERR_FORMAT, "Failed to parse sequence: %s: sequence and quality lengths differ", path_.c_str());
        p = after(le);
    }
    pos_ = p;
    seq_.push_back(0);  // keeps data() non-null for empty sequences; not counted
    rec.seq = seq_.data();
    rec.seq_len = seq_.size() - 1;
    return 1;
}

}  // namespace smafa

// ---------------------------------------------------------------------------------------------- bulk loading
#include <algorithm>
#include <thread>

namespace smafa {

namespace {

struct ChunkResult {
    uint64_t n_records = 0;   // records that start in the chunk
    uint64_t good = 0;        // parsed without error, in order
    int err_kind = 0;
    size_t err_len = 0;
    std::string err_msg;
};

// parse the records starting in [lo, hi) of a FASTA buffer into rows first_row.. of codes/raw
void parse_fasta_chunk(const uint8_t *d, size_t n, size_t lo, size_t hi, int alphabet, size_t L, uint64_t first_row,
                       uint8_t *codes, uint8_t *raw, ChunkResult &res) {
    std::vector<uint8_t> seq;
    size_t p = lo;
    uint64_t row = first_row;
    auto eol = [&](size_t from) {
        const void *q = from < n ? memchr(d + from, '\n', n - from) : nullptr;
        return q ? (size_t)((const uint8_t *)q - d) : n;
    };
    while (p < hi) {
        // p is at a '>' that begins a line
        const size_t hs = p + 1, he = eol(hs);
        size_t id_len = he - hs;
        if (id_len && d[hs + id_len - 1] == '\r') id_len--;
        size_t q = he < n ? he + 1 : n;
        const uint8_t *s;
        size_t slen;
        const size_t le0 = eol(q);
        const size_t q2 = le0 < n ? le0 + 1 : n;
        if ((q2 >= n || d[q2] == '>') && !(le0 > q && memchr(d + q, '\r', le0 - q))) {
            s = d + q;  // one sequence line, no CR: in place
            slen = le0 - q;
            q = q2;
        } else {
            seq.clear();
            while (q < n && d[q] != '>') {
                const size_t le = eol(q);
                for (size_t i = q; i < le; i++)
                    if (d[i] != '\r') seq.push_back(d[i]);
                q = le < n ? le + 1 : n;
            }
            s = seq.data();
            slen = seq.size();
        }
        // encode first (src/lib.rs:150,235: from_bytes runs before any length check), into a scratch row if the
        // length is wrong so that a bad byte in an over-long record is still reported
        uint8_t *crow = codes + (size_t)row * L;
        for (size_t i = 0; i < slen; i++) {
            const uint8_t c = code_of(alphabet, s[i]);
            if (c == 255) {
                char msg[512];
                snprintf(msg, sizeof msg, "Byte %u cannot be interpreted as %s, in sequence \"%.*s\" at position %zu", s[i],
                         alphabet_noun(alphabet), (int)std::min<size_t>(id_len, 300), (const char *)(d + hs), i);
                res.err_kind = 1;
                res.err_msg = msg;
                return;
            }
            if (i < L) crow[i] = c;
        }
        if (slen != L) {
            res.err_kind = 2;
            res.err_len = slen;
            return;
        }
        if (raw) memcpy(raw + (size_t)row * L, s, L);
        row++;
        res.good++;
        p = q;
        while (p < hi && (d[p] == '\n' || d[p] == '\r')) p++;  // blank lines between records
        if (p < hi && d[p] != '>') {
            res.err_kind = 4;
            res.err_msg = "record does not start with '>'";
            return;
        }
    }
}

}  // namespace

int load_records_bulk(const char *path, int alphabet, bool want_raw, BulkRecords &out) {
    out = BulkRecords();
    FastxReader reader;
    int rc = reader.open(path);
    if (rc) return rc;
    const uint8_t *d = reader.data();
    const size_t n = reader.size();
    unsigned n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (reader.is_fastq() || n < (32u << 20) || n_threads < 2) {
        // sequential path: FastxReader record by record
        FastxRecord rec;
        while ((rc = reader.next(rec)) == 1) {
            if (out.n == 0) {
                out.L = rec.seq_len;
            }
            const size_t base = out.codes.size();
            out.codes.resize(base + rec.seq_len);
            for (size_t i = 0; i < rec.seq_len; i++) {
                const uint8_t c = code_of(alphabet, rec.seq[i]);
                if (c == 255) {
                    out.codes.resize(base);
                    char msg[512];
                    snprintf(msg, sizeof msg, "Byte %u cannot be interpreted as %s, in sequence \"%.*s\" at position %zu",
                             rec.seq[i], alphabet_noun(alphabet), (int)std::min<size_t>(rec.id_len, 300), (const char *)rec.id, i);
                    out.err_kind = 1;
                    out.err_msg = msg;
                    return SMAFA_OK;
                }
                out.codes[base + i] = c;
            }
            if (out.n == 0 && rec.seq_len == 0) {
                out.codes.resize(base);
                out.err_kind = 3;
                return SMAFA_OK;
            }
            if (rec.seq_len != out.L) {
                out.codes.resize(base);
                out.err_kind = 2;
                out.err_len = rec.seq_len;
                return SMAFA_OK;
            }
            if (want_raw) out.raw.insert(out.raw.end(), rec.seq, rec.seq + rec.seq_len);
            out.n++;
        }
        if (rc < 0) {
            out.err_kind = 4;
            out.err_msg = smafa_last_error();
        }
        return SMAFA_OK;
    }
    // parallel path (plain FASTA): the first record fixes L
    {
        FastxRecord rec;
        rc = reader.next(rec);
        if (rc < 0) {
            out.err_kind = 4;
            out.err_msg = smafa_last_error();
            return SMAFA_OK;
        }
        if (rc == 0) return SMAFA_OK;
        out.L = rec.seq_len;
        if (rec.seq_len == 0) {
            // still report a bad byte first?  an empty sequence has none
            out.err_kind = 3;
            return SMAFA_OK;
        }
    }
    const size_t L = out.L;
    // chunk boundaries snapped forward to the next record start ('>' after '\n')
    std::vector<size_t> bounds(n_threads + 1, n);
    bounds[0] = 0;
    for (unsigned t = 1; t < n_threads; t++) {
        size_t p = (size_t)((unsigned __int128)n * t / n_threads);
        while (true) {
            const void *q = p < n ? memchr(d + p, '>', n - p) : nullptr;
            if (!q) {
                p = n;
                break;
            }
            p = (size_t)((const uint8_t *)q - d);
            if (p == 0 || d[p - 1] == '\n') break;
            p++;
        }
        bounds[t] = p;
    }
    for (unsigned t = 1; t <= n_threads; t++) bounds[t] = std::max(bounds[t], bounds[t - 1]);
    // pass 1: count record starts per chunk
    std::vector<ChunkResult> res(n_threads);
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; t++)
            pool.emplace_back([&, t] {
                uint64_t c = 0;
                size_t p = bounds[t];
                const size_t hi = bounds[t + 1];
                while (p < hi) {
                    const void *q = memchr(d + p, '>', hi - p);
                    if (!q) break;
                    p = (size_t)((const uint8_t *)q - d);
                    if (p == 0 || d[p - 1] == '\n') c++;
                    p++;
                }
                res[t].n_records = c;
            });
        for (auto &th : pool) th.join();
    }
    uint64_t total = 0;
    std::vector<uint64_t> first_row(n_threads);
    for (unsigned t = 0; t < n_threads; t++) {
        first_row[t] = total;
        total += res[t].n_records;
    }
    out.codes.resize((size_t)total * L);
    if (want_raw) out.raw.resize((size_t)total * L);
    // pass 2: parse + encode
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; t++)
            pool.emplace_back([&, t] {
                size_t lo = bounds[t];
                while (lo < bounds[t + 1] && (d[lo] == '\n' || d[lo] == '\r')) lo++;
                if (lo < bounds[t + 1])
                    parse_fasta_chunk(d, n, lo, bounds[t + 1], alphabet, L, first_row[t], out.codes.data(),
                                      want_raw ? out.raw.data() : nullptr, res[t]);
            });
        for (auto &th : pool) th.join();
    }
    // first error in file order decides
    uint64_t good = 0;
    for (unsigned t = 0; t < n_threads; t++) {
        good += res[t].good;
        if (res[t].err_kind) {
            out.err_kind = res[t].err_kind;
            out.err_len = res[t].err_len;
            out.err_msg = res[t].err_msg;
            break;
        }
    }
    out.n = good;
    out.codes.resize((size_t)good * L);
    if (want_raw) out.raw.resize((size_t)good * L);
    return SMAFA_OK;
}

}  // namespace smafa
