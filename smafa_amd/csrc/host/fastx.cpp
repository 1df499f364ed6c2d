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

// host/inflate_other.cpp
int inflate_bzip2(const char *path, const uint8_t *in, size_t n, std::vector<uint8_t> &out);
int inflate_xz(const char *path, const uint8_t *in, size_t n, std::vector<uint8_t> &out);

int FastxReader::open(const char *path) {
    path_ = path;
    FILE *f = fopen(path, "rb");
    if (!f) return set_error(SMAFA_ERR_IO, "valid path/file expected: %s: %s", path, strerror(errno));
    unsigned char magic[6] = {0};
    const size_t got = fread(magic, 1, sizeof magic, f);
    fclose(f);
    const bool bz2 = got >= 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h';
    const bool xz = got >= 6 && magic[0] == 0xfd && magic[1] == '7' && magic[2] == 'z' && magic[3] == 'X' && magic[4] == 'Z';
    data_.clear();
    if (map_) munmap(map_, size_);
    map_ = nullptr;
    base_ = nullptr;
    size_ = 0;
    const bool gz = got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (bz2 || xz) {  // needletail sniffs these two as well (Cargo.toml:27): the file image is mapped and decompressed whole
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return set_error(SMAFA_ERR_IO, "%s: cannot open", path);
        struct stat st;
        void *m = MAP_FAILED;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return set_error(SMAFA_ERR_IO, "%s: cannot map", path);
        const int rc = bz2 ? inflate_bzip2(path, (const uint8_t *)m, (size_t)st.st_size, data_)
                           : inflate_xz(path, (const uint8_t *)m, (size_t)st.st_size, data_);
        munmap(m, (size_t)st.st_size);
        if (rc) return rc;
    } else if (!gz) {  // plain file: mapped when it is a regular file, otherwise read to the end
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
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace smafa {

namespace {

// How much of the input is in memory: a mapped plain file is all there; a gzip file is inflated by one thread (a gzip
// stream cannot be split) while the parser threads work on the part that has already come out.
struct Progress {
    std::mutex m;
    std::condition_variable cv;
    size_t avail = 0;
    bool done = false, failed = false;
    void publish(size_t n, bool fin, bool fail = false) {
        {
            std::lock_guard<std::mutex> g(m);
            avail = n;
            done = done || fin;
            failed = failed || fail;
        }
        cv.notify_all();
    }
    // blocks until `need` bytes are available or the stream has ended; returns the bytes available
    size_t wait_for(size_t need) {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return avail >= need || done; });
        return avail;
    }
};

inline size_t line_end(const uint8_t *d, size_t limit, size_t from) {  // index of the '\n' ending the line, or limit
    const void *q = from < limit ? memchr(d + from, '\n', limit - from) : nullptr;
    return q ? (size_t)((const uint8_t *)q - d) : limit;
}

constexpr size_t kUndecided = (size_t)-1;

// Start of the first record at or after `from`, looking at d[0 .. limit) of a `total`-byte input.
//   FASTA: a '>' that begins a line — unambiguous.
//   FASTQ (4-line records): an '@' that begins a line, whose third line begins with '+' and whose fourth line is as long
//   as its second.  A quality line may begin with '@' too; then the "second line" is the next header and the "third" a
//   sequence line, which cannot begin with '+' in valid input — and the loader checks every chunk against its
//   neighbour afterwards, falling back to the one-thread reader if a boundary was wrong.
// Returns total when there is no further record, kUndecided when the answer needs bytes past `limit`.
size_t snap_record_start(const uint8_t *d, size_t limit, size_t total, size_t from, bool fastq) {
    size_t p = from;
    if (p > 0) {  // move to a line start
        const size_t e = line_end(d, limit, p - 1);
        if (e == limit) return limit < total ? kUndecided : total;
        p = e + 1;
    }
    while (true) {
        if (p >= limit) return limit < total ? kUndecided : total;
        if (d[p] == (fastq ? '@' : '>')) {
            if (!fastq) return p;
            const size_t e1 = line_end(d, limit, p);
            const size_t e2 = e1 < limit ? line_end(d, limit, e1 + 1) : limit;
            const size_t e3 = e2 < limit ? line_end(d, limit, e2 + 1) : limit;
            const size_t e4 = e3 < limit ? line_end(d, limit, e3 + 1) : limit;
            if (e4 == limit && limit < total) return kUndecided;
            if (e3 < limit && e2 + 1 < limit && d[e2 + 1] == '+') {
                size_t sl = e2 - (e1 + 1), ql = e4 - (e3 + 1);
                if (sl && d[e2 - 1] == '\r') sl--;
                if (ql && e4 > e3 + 1 && d[e4 - 1] == '\r') ql--;
                if (sl == ql) return p;
            }
        }
        const size_t e = line_end(d, limit, p);
        if (e == limit) return limit < total ? kUndecided : total;
        p = e + 1;
    }
}

struct ChunkOut {
    std::vector<uint8_t> codes, raw;
    uint64_t good = 0;    // records parsed without error, in order
    size_t L = 0;         // length of the chunk's first record
    bool have_L = false;
    size_t lo = 0, hi = 0, end_pos = 0;  // [lo, hi): where the chunk's records start; end_pos: where its parse stopped
    int err_kind = 0;     // as BulkRecords
    size_t err_len = 0;
    std::string err_msg;
    bool truncated = false;  // a line ran into the end of the bytes available so far: parse again when more are there
};

void bad_byte(ChunkOut &c, int alphabet, uint8_t byte, const uint8_t *id, size_t id_len, size_t pos) {
    char msg[512];
    snprintf(msg, sizeof msg, "Byte %u cannot be interpreted as %s, in sequence \"%.*s\" at position %zu", byte,
             alphabet_noun(alphabet), (int)std::min<size_t>(id_len, 300), (const char *)id, pos);
    c.err_kind = 1;
    c.err_msg = msg;
}

// parse the records starting in [c.lo, c.hi) of d[0..n) into c (FASTA, multi-line joined, or 4-line FASTQ); the whole
// input has `total` bytes, of which n are in memory
void parse_chunk(const uint8_t *d, size_t n, size_t total, bool fastq, int alphabet, bool want_raw, ChunkOut &c) {
    std::vector<uint8_t> seq;
    const uint8_t *tab = code_table(alphabet);  // (a call per byte was a third of the parse)
    size_t p = c.lo;
    auto line_end = [&](const uint8_t *dd, size_t lim, size_t from) {
        const size_t e = smafa::line_end(dd, lim, from);
        if (e == lim && lim < total) c.truncated = true;
        return e;
    };
    const char marker = fastq ? '@' : '>';
    while (p < c.hi && (d[p] == '\n' || d[p] == '\r')) p++;
    while (p < c.hi) {
        if (d[p] != marker) {
            c.err_kind = 4;
            c.err_msg = fastq ? "record does not start with '@'" : "record does not start with '>'";
            c.end_pos = p;
            return;
        }
        const size_t hs = p + 1, he = line_end(d, n, hs);
        size_t id_len = he - hs;
        if (id_len && d[hs + id_len - 1] == '\r') id_len--;
        size_t q = he < n ? he + 1 : n;
        const uint8_t *s;
        size_t slen;
        if (fastq) {
            const size_t le = line_end(d, n, q);
            slen = le - q;
            if (slen && d[q + slen - 1] == '\r') slen--;
            s = d + q;
            q = le < n ? le + 1 : n;
            if (q >= n || d[q] != '+') {
                c.err_kind = 4;
                c.err_msg = "FASTQ record without '+' line";
                c.end_pos = p;
                return;
            }
            const size_t pe = line_end(d, n, q);
            q = pe < n ? pe + 1 : n;
            const size_t qe = line_end(d, n, q);
            size_t qlen = qe - q;
            if (qlen && d[q + qlen - 1] == '\r') qlen--;
            if (qlen != slen) {
                c.err_kind = 4;
                c.err_msg = "sequence and quality lengths differ";
                c.end_pos = p;
                return;
            }
            q = qe < n ? qe + 1 : n;
        } else {
            const size_t le0 = line_end(d, n, q);
            const size_t q2 = le0 < n ? le0 + 1 : n;
            if ((q2 >= n || d[q2] == '>') && !(le0 > q && memchr(d + q, '\r', le0 - q))) {
                s = d + q;  // one sequence line, no CR: in place
                slen = le0 - q;
                q = q2;
            } else {
                seq.clear();
                while (q < n && d[q] != '>') {
                    const size_t le = line_end(d, n, q);
                    for (size_t i = q; i < le; i++)
                        if (d[i] != '\r') seq.push_back(d[i]);
                    q = le < n ? le + 1 : n;
                }
                s = seq.data();
                slen = seq.size();
            }
        }
        if (!c.have_L) {
            c.L = slen;
            c.have_L = true;
        }
        // encode first (src/lib.rs:150,235: from_bytes runs before any length check)
        const size_t base = c.codes.size();
        c.codes.resize(base + slen);
        {
            uint8_t *dst = c.codes.data() + base;
            uint8_t seen = 0;  // codes are < 32; 255 marks a byte outside the alphabet
            for (size_t i = 0; i < slen; i++) {
                const uint8_t code = tab[s[i]];
                dst[i] = code;
                seen |= code;
            }
            if (seen & 0x80) {
                size_t i = 0;
                while (tab[s[i]] != 255) i++;
                const uint8_t byte = s[i];
                c.codes.resize(base);
                bad_byte(c, alphabet, byte, d + hs, id_len, i);
                c.end_pos = p;
                return;
            }
        }
        if (slen != c.L) {
            c.codes.resize(base);
            c.err_kind = 2;
            c.err_len = slen;
            c.end_pos = p;
            return;
        }
        if (want_raw) c.raw.insert(c.raw.end(), s, s + slen);
        c.good++;
        p = q;
        while (p < c.hi && (d[p] == '\n' || d[p] == '\r')) p++;
    }
    c.end_pos = p;
}

// one-thread loader over FastxReader: the definition of the loader's behaviour, and its fallback
int load_sequential(FastxReader &reader, int alphabet, bool want_raw, BulkRecords &out) {
    FastxRecord rec;
    int rc;
    while ((rc = reader.next(rec)) == 1) {
        if (out.n == 0) out.L = rec.seq_len;
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

// a mapped file
struct Mapped {
    const uint8_t *p = nullptr;
    size_t len = 0;
    ~Mapped() {
        if (p) munmap((void *)p, len);
    }
    bool open(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
        if (ok) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            ok = m != MAP_FAILED;
            if (ok) {
                p = (const uint8_t *)m;
                len = (size_t)st.st_size;
            }
        }
        close(fd);
        return ok;
    }
};

// Threaded parse of d[0..total) — FASTA or 4-line FASTQ — as the bytes become available.  Returns false when the
// chunking could not be trusted (a boundary was not a record start, the inflate failed, ...): the caller then runs
// the one-thread reader, whose result is the definition.
bool load_parallel(const uint8_t *d, size_t total, Progress &pr, bool fastq, int alphabet, bool want_raw, unsigned T,
                   BulkRecords &out) {
    std::vector<ChunkOut> chunks(T);
    std::atomic<bool> distrust{false};
    const size_t slack = 1u << 20;  // a boundary is looked for within this much beyond the raw cut (then: wait for the end)
    auto boundary = [&](size_t raw_cut) -> size_t {
        if (raw_cut == 0) return 0;
        if (raw_cut >= total) return total;
        size_t have = pr.wait_for(std::min(total, raw_cut + slack));
        while (true) {
            const size_t b = snap_record_start(d, std::min(have, total), total, raw_cut, fastq);
            if (b != kUndecided) return b;
            if (have >= total) return total;
            const size_t more = pr.wait_for(std::min(total, have + slack));
            if (more == have) return total;  // stream ended short: the inflate thread reports the failure
            have = more;
        }
    };
    auto work = [&](unsigned t) {
        ChunkOut &c = chunks[t];
        c.lo = boundary((size_t)((unsigned __int128)total * t / T));
        c.hi = boundary((size_t)((unsigned __int128)total * (t + 1) / T));
        // every record that STARTS before hi is parsed to its end: wait for the bytes after hi as well
        const size_t need = std::min(total, c.hi + slack);
        const size_t have = pr.wait_for(need);
        if (have < need) {  // stream ended short of the announced size
            distrust = true;
            return;
        }
        c.codes.reserve(c.hi - c.lo);  // an upper bound: no reallocation while the chunk is parsed
        if (want_raw) c.raw.reserve(c.hi - c.lo);
        parse_chunk(d, std::min(have, total), total, fastq, alphabet, want_raw, c);
        if (c.truncated) {  // a record longer than the slack: once more with everything there
            const size_t lo = c.lo, hi = c.hi;
            c = ChunkOut();
            c.lo = lo;
            c.hi = hi;
            if (pr.wait_for(total) < total) {
                distrust = true;
                return;
            }
            parse_chunk(d, total, total, fastq, alphabet, want_raw, c);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; t++) pool.emplace_back(work, t);
    for (auto &th : pool) th.join();
    {
        std::lock_guard<std::mutex> g(pr.m);
        if (pr.failed) return false;
    }
    if (distrust) return false;
    // chunks in file order: each must end where the next begins; the first error decides
    unsigned last = T;  // chunks [0, last) contribute rows; chunk `last - 1` may carry the error
    if (!chunks[0].have_L) return false;
    for (unsigned t = 0; t < T; t++) {
        ChunkOut &c = chunks[t];
        if (c.err_kind == 4) return false;  // malformed record: the one-thread reader words the message
        if (t > 0 && c.lo != chunks[t - 1].hi) return false;
        if (t > 0 && c.have_L && chunks[0].have_L && c.L != chunks[0].L && c.good + (c.err_kind ? 1 : 0) > 0) {
            // its FIRST record has another length than the file's first record: that record is the offender (a bad
            // byte in it would have been reported first, src/lib.rs:235-238) — unless it already failed to encode
            if (!(c.err_kind == 1 && c.good == 0)) {
                c.err_kind = 2;
                c.err_len = c.L;
            }
            c.good = 0;
            c.codes.clear();
            c.raw.clear();
        }
        if (c.err_kind) {
            last = t + 1;
            break;
        }
        if (c.end_pos != c.hi) return false;  // a record straddled the boundary: the snap was wrong
    }
    const ChunkOut &first = chunks[0];
    out.L = first.have_L ? first.L : 0;
    if (first.L == 0) {
        out.err_kind = 3;  // the first record is empty (src/lib.rs:103-108)
        return true;
    }
    uint64_t rows = 0;
    std::vector<uint64_t> row0(last, 0);
    for (unsigned t = 0; t < last; t++) {
        row0[t] = rows;
        rows += chunks[t].good;
    }
    const size_t L = out.L;
    out.n = rows;
    out.codes.resize((size_t)rows * L);
    if (want_raw) out.raw.resize((size_t)rows * L);
    {
        std::vector<std::thread> copy;
        for (unsigned t = 0; t < last; t++)
            copy.emplace_back([&, t] {
                const ChunkOut &c = chunks[t];
                if (c.good == 0) return;
                memcpy(out.codes.data() + (size_t)row0[t] * L, c.codes.data(), (size_t)c.good * L);
                if (want_raw) memcpy(out.raw.data() + (size_t)row0[t] * L, c.raw.data(), (size_t)c.good * L);
            });
        for (auto &th : copy) th.join();
    }
    if (last > 0 && chunks[last - 1].err_kind) {
        out.err_kind = chunks[last - 1].err_kind;
        out.err_len = chunks[last - 1].err_len;
        out.err_msg = chunks[last - 1].err_msg;
    }
    return true;
}

}  // namespace

// One part of a plain FASTA/FASTQ file (see fastx.h).  cut(i) = the first record start at or after byte total * i / parts
// (snap_record_start: the rule the threaded loader cuts by); part p owns the records that start in [cut(p), cut(p + 1)).
int load_records_part(const char *path, int alphabet, unsigned part, unsigned parts, BulkRecords &out, bool *usable) {
    out = BulkRecords();
    *usable = false;
    if (parts == 0 || part >= parts) return set_error(SMAFA_ERR_INVALID, "part %u of %u", part, parts);
    Mapped file;
    if (!file.open(path)) return SMAFA_OK;  // unreadable or empty: the whole-file loader words the failure
    const bool gz = file.len >= 2 && file.p[0] == 0x1f && file.p[1] == 0x8b;
    if (gz || (file.p[0] != '>' && file.p[0] != '@')) return SMAFA_OK;  // a gzip stream cannot be cut
    const bool fastq = file.p[0] == '@';
    const size_t total = file.len;
    auto cut = [&](unsigned i) -> size_t {
        if (i == 0) return 0;
        if (i >= parts) return total;
        const size_t b = snap_record_start(file.p, total, total, (size_t)((unsigned __int128)total * i / parts), fastq);
        return b == kUndecided ? total : b;
    };
    const size_t lo = cut(part), hi = cut(part + 1);
    if (hi > lo && hi - lo >= (32u << 20)) {  // a big part: this process's threads share it (the part is a file of its own)
        const unsigned T = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        Progress pr;
        pr.publish(hi - lo, true);
        if (T >= 2 && load_parallel(file.p + lo, hi - lo, pr, fastq, alphabet, false, T, out)) {
            *usable = true;
            return SMAFA_OK;
        }
        out = BulkRecords();
    }
    ChunkOut c;
    c.lo = lo;
    c.hi = hi;
    c.codes.reserve(hi - lo);
    parse_chunk(file.p, total, total, fastq, alphabet, false, c);
    // the part must end exactly where the next one begins (else a cut was not a record start: nobody can use the parts);
    // a malformed record is worded by the one-thread reader, as in the threaded loader
    if (c.err_kind == 4 || (!c.err_kind && c.end_pos != hi)) return SMAFA_OK;
    out.L = c.have_L ? c.L : 0;
    out.n = c.good;
    out.codes.swap(c.codes);
    out.err_kind = c.err_kind;
    out.err_len = c.err_len;
    out.err_msg = c.err_msg;
    if (c.have_L && c.L == 0 && c.good == 0 && !c.err_kind) out.err_kind = 3;
    *usable = true;
    return SMAFA_OK;
}

uint64_t fastx_expanded_size(const char *path) {
    Mapped f;
    if (!f.open(path)) return 0;
    if (f.len >= 18 && f.p[0] == 0x1f && f.p[1] == 0x8b) {  // gzip: ISIZE, the last member's size mod 2^32
        uint32_t isize;
        memcpy(&isize, f.p + f.len - 4, 4);
        return std::max<uint64_t>(isize, f.len);
    }
    return f.len;
}

int load_records_bulk(const char *path, int alphabet, bool want_raw, BulkRecords &out) {
    out = BulkRecords();
    const unsigned n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    Mapped file;
    const bool mapped = file.open(path);
    const bool gz = mapped && file.len >= 18 && file.p[0] == 0x1f && file.p[1] == 0x8b;
    if (mapped && n_threads >= 2 && !gz && file.len >= (32u << 20) && (file.p[0] == '>' || file.p[0] == '@')) {
        // big plain file: every thread parses its share of the mapping
        Progress pr;
        pr.publish(file.len, true);
        if (load_parallel(file.p, file.len, pr, file.p[0] == '@', alphabet, want_raw, n_threads, out)) {
            log_line(2, "%s: %s, %llu records parsed by %u threads", path, file.p[0] == '@' ? "FASTQ" : "FASTA",
                     (unsigned long long)out.n, n_threads);
            return SMAFA_OK;
        }
        out = BulkRecords();
    } else if (gz && n_threads >= 2) {
        uint32_t isize;
        memcpy(&isize, file.p + file.len - 4, 4);
        if (isize >= (32u << 20) && (uint64_t)isize >= file.len / 2) {
            // big single-member gzip file: one thread inflates into a buffer of the announced size, the others parse
            // what has come out.  Anything unexpected (more members, another size) -> the one-thread reader below.
            std::vector<uint8_t> buf((size_t)isize);
            Progress pr;
            std::thread inflater([&] {
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, 15 + 16) != Z_OK) {
                    pr.publish(0, true, true);
                    return;
                }
                size_t in_pos = 0, out_done = 0;
                bool ok = true, end = false;
                while (ok && !end) {
                    if (zs.avail_in == 0) {
                        if (in_pos >= file.len) {
                            ok = false;  // input exhausted before the stream ended
                            break;
                        }
                        const size_t chunk = std::min<size_t>(file.len - in_pos, 1u << 30);
                        zs.next_in = const_cast<Bytef *>(file.p + in_pos);
                        zs.avail_in = (uInt)chunk;
                        in_pos += chunk;
                    }
                    const size_t room = buf.size() - out_done;
                    uint8_t probe;  // the announced size is used up: the stream must end without producing another byte
                    zs.next_out = room ? buf.data() + out_done : &probe;
                    zs.avail_out = room ? (uInt)std::min<size_t>(room, 4u << 20) : 1u;
                    const int r = inflate(&zs, Z_NO_FLUSH);
                    if (room) out_done = (size_t)(zs.next_out - buf.data());
                    else if (zs.avail_out == 0) ok = false;  // more output than announced
                    if (r == Z_STREAM_END) end = true;
                    else if (r != Z_OK && r != Z_BUF_ERROR) ok = false;
                    if (ok) pr.publish(out_done, false);
                }
                const bool whole = ok && end && out_done == buf.size() && in_pos - zs.avail_in == file.len;
                inflateEnd(&zs);
                pr.publish(out_done, true, !whole);
            });
            // format by first byte, once the first block is out
            const size_t have = pr.wait_for(1);
            bool ok = have >= 1 && (buf[0] == '>' || buf[0] == '@');
            if (ok) ok = load_parallel(buf.data(), buf.size(), pr, buf[0] == '@', alphabet, want_raw, n_threads - 1, out);
            inflater.join();
            if (ok) {
                log_line(2, "%s: gzip %s, %llu records parsed by %u threads while one thread inflated", path,
                         buf[0] == '@' ? "FASTQ" : "FASTA", (unsigned long long)out.n, n_threads - 1);
                return SMAFA_OK;
            }
            out = BulkRecords();
        }
    }
    FastxReader reader;
    int rc = reader.open(path);
    if (rc) return rc;
    rc = load_sequential(reader, alphabet, want_raw, out);
    log_line(2, "%s: %llu records through the one-thread reader", path, (unsigned long long)out.n);
    return rc;
}

}  // namespace smafa
