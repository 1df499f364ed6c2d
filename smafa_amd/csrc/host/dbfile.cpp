// dbfile.cpp — the smafa DB file.
//
// Version 2 is the reference's format: postcard 1.x serialisation of
//   WindowSet { version: u32, windows: Vec<SeqEncoding(Vec<u64>)>, len: Option<NonZeroUsize> }
// (/root/reference/src/lib.rs:54-60, written at :161-162, read at :208-218).  postcard writes every
// integer as a little-endian base-128 varint, a Vec as varint(length) + elements, an Option as a 0/1 tag.
// Each u64 holds 12 symbols, 5 bits each, one-hot: A=16 C=8 G=4 T=2 N=1 (src/lib.rs:31-46,171-178).
// Files written here are byte-identical to the reference's (tests/golden/*.smafadb), and files written
// by the reference load here.
//
// Version 3 is this build's container for stores the reference cannot express (amino acids):
//   varint(3) varint(alphabet) varint(n) varint(seq_len) then n*seq_len raw code bytes.
// The reference rejects it with its "Unsupported db file version: 3." panic, as it should.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../engine.h"
#include "packed.h"

namespace smafa {

namespace {

const uint8_t kOneHotOfCode[5] = {16, 8, 4, 2, 1};  // codes 0..4 = A C G T N

void put_varint(std::vector<uint8_t> &b, uint64_t v) {
    while (v >= 0x80) {
        b.push_back((uint8_t)(v | 0x80));
        v >>= 7;
    }
    b.push_back((uint8_t)v);
}

// max_bytes: 5 for u32, 10 for u64/usize
bool get_varint(const uint8_t *buf, size_t len, size_t &pos, int max_bytes, uint64_t &out) {
    uint64_t v = 0;
    for (int i = 0; i < max_bytes; i++) {
        if (pos >= len) return false;
        const uint8_t byte = buf[pos++];
        if (i == 9 && byte > 1) return false;
        v |= (uint64_t)(byte & 0x7f) << (7 * i);
        if (!(byte & 0x80)) {
            if (max_bytes == 5 && v > 0xffffffffull) return false;
            out = v;
            return true;
        }
    }
    return false;
}

// The bytes of a file: regular files are mapped (no copy — a 10M-row store is 460 MB, and the worker threads of the
// decoder fault its pages in side by side); anything else (pipes, /dev/stdin) is read into a buffer.
struct FileBytes {
    const uint8_t *p = nullptr;
    size_t len = 0;
    void *map = nullptr;
    std::vector<uint8_t> buf;
    ~FileBytes() {
        if (map) munmap(map, len);
    }
};

int read_file(const char *path, FileBytes &out) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return set_error(SMAFA_ERR_IO, "%s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            close(fd);
            out.map = m;
            out.p = (const uint8_t *)m;
            out.len = (size_t)st.st_size;
            return SMAFA_OK;
        }
    }
    std::vector<uint8_t> chunk(1u << 20);
    for (;;) {
        const ssize_t r = read(fd, chunk.data(), chunk.size());
        if (r < 0) {
            if (errno == EINTR) continue;
            const int e = errno;
            close(fd);
            return set_error(SMAFA_ERR_IO, "%s: %s", path, strerror(e));
        }
        if (r == 0) break;
        out.buf.insert(out.buf.end(), chunk.begin(), chunk.begin() + r);
    }
    close(fd);
    out.p = out.buf.data();
    out.len = out.buf.size();
    return SMAFA_OK;
}

}  // namespace

}  // namespace smafa

using namespace smafa;

extern "C" {

int smafa_dbfile_write(const char *path, int alphabet, const uint8_t *codes, uint64_t n, uint32_t seq_len) try {
    if (!path || (!codes && n)) return set_error(SMAFA_ERR_INVALID, "smafa_dbfile_write: NULL argument");
    std::vector<uint8_t> b;
    std::vector<std::vector<uint8_t>> parts;  // window bytes, one buffer per worker, in row order
    std::vector<uint8_t> tail;
    if (alphabet == SMAFA_ALPHABET_NT) {
        const size_t nw = ((size_t)seq_len + 11) / 12;
        put_varint(b, SMAFA_DB_VERSION);
        put_varint(b, n);
        unsigned n_threads = n >= (1u << 20) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        parts.resize(n_threads);
        std::vector<int> bad(n_threads, -1);
        auto encode_rows = [&](unsigned t) {
            const uint64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
            std::vector<uint8_t> &out = parts[t];
            // written through a raw pointer into a buffer sized for the worst case (10 bytes per varint), cut to size
            // at the end: a push_back per byte was most of makedb's serialisation time
            out.resize((size_t)(hi - lo) * (10 + nw * 10));
            uint8_t *p = out.data();
            auto put = [&](uint64_t v) {
                while (v >= 0x80) {
                    *p++ = (uint8_t)(v | 0x80);
                    v >>= 7;
                }
                *p++ = (uint8_t)v;
            };
            for (uint64_t j = lo; j < hi; j++) {
                const uint8_t *row = codes + (size_t)j * seq_len;
                put(nw);
                for (size_t w = 0; w < nw; w++) {
                    uint64_t word = 0;
                    const size_t lim = std::min<size_t>(12, seq_len - w * 12);
                    for (size_t i = 0; i < lim; i++) {
                        const uint8_t c = row[w * 12 + i];
                        if (c > 4) {
                            bad[t] = c;
                            return;
                        }
                        word |= (uint64_t)kOneHotOfCode[c] << (5 * i);
                    }
                    put(word);
                }
            }
            out.resize((size_t)(p - out.data()));
        };
        if (n_threads > 1) {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < n_threads; t++) pool.emplace_back(encode_rows, t);
            for (auto &th : pool) th.join();
        } else {
            encode_rows(0);
        }
        for (int c : bad)
            if (c >= 0) return set_error(SMAFA_ERR_INVALID, "code %d is not a nucleotide code", c);
        if (n == 0) {
            tail.push_back(0);  // len: None
        } else {
            tail.push_back(1);
            put_varint(tail, seq_len);
        }
    } else if (alphabet == SMAFA_ALPHABET_AA) {
        put_varint(b, 3);
        put_varint(b, (uint64_t)alphabet);
        put_varint(b, n);
        put_varint(b, seq_len);
        b.insert(b.end(), codes, codes + (size_t)n * seq_len);
    } else {
        return set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    }
    FILE *f = fopen(path, "wb");
    if (!f) return set_error(SMAFA_ERR_IO, "%s: %s", path, strerror(errno));
    bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
    for (const std::vector<uint8_t> &part : parts) ok = ok && fwrite(part.data(), 1, part.size(), f) == part.size();
    ok = ok && fwrite(tail.data(), 1, tail.size(), f) == tail.size();
    if (fclose(f) != 0 || !ok) return set_error(SMAFA_ERR_IO, "%s: write error", path);
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_dbfile_write");
}

int smafa_dbfile_read(const char *path, int *alphabet, uint8_t **codes, uint64_t *n, uint32_t *seq_len) try {
    if (!path || !alphabet || !codes || !n || !seq_len) return set_error(SMAFA_ERR_INVALID, "smafa_dbfile_read: NULL argument");
    *codes = nullptr;
    *n = 0;
    *seq_len = 0;
    FileBytes file;
    int rc = read_file(path, file);
    if (rc) return rc;
    struct {  // the decoder below was written against a vector: same two accessors
        const uint8_t *p_;
        size_t n_;
        const uint8_t *data() const { return p_; }
        size_t size() const { return n_; }
    } buf = {file.p, file.len};
    // src/lib.rs:214 decodes the version from &buffer[0..4]
    if (buf.size() < 4)
        return set_error(SMAFA_ERR_PANIC, "range end index 4 out of range for slice of length %zu", buf.size());
    size_t pos = 0;
    uint64_t version = 0;
    if (!get_varint(buf.data(), 4, pos, 5, version)) return set_error(SMAFA_ERR_FORMAT, "DeserializeUnexpectedEnd");
    if (is_packed_file(buf.data(), buf.size())) {  // version 3, kind 2: bit-plane tiles (host/packed.cpp): decode every row
        PackedStore pk;
        rc = pk.open(path);
        if (rc) return rc;
        const uint64_t cnt = pk.h.n, len = pk.h.seq_len;
        uint8_t *out = (uint8_t *)malloc(std::max<size_t>(cnt * len, 1));
        if (!out) return set_error(SMAFA_ERR_NOMEM, "out of host memory");
        const unsigned T = cnt >= (1u << 16) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        std::vector<int> rcs(T, SMAFA_OK);
        std::vector<std::string> msgs(T);
        auto work = [&](unsigned t) {
            for (uint64_t j = cnt * t / T, e = cnt * (t + 1) / T; j < e && rcs[t] == SMAFA_OK; j++) {
                rcs[t] = pk.row(j, out + (size_t)j * len);
                if (rcs[t]) msgs[t] = smafa_last_error();
            }
        };
        if (T == 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < T; t++) pool.emplace_back(work, t);
            for (auto &th : pool) th.join();
        }
        for (unsigned t = 0; t < T; t++)
            if (rcs[t]) {
                free(out);
                return set_error(rcs[t], "%s", msgs[t].c_str());
            }
        *alphabet = (int)pk.h.alphabet;
        *codes = out;
        *n = cnt;
        *seq_len = (uint32_t)len;
        return SMAFA_OK;
    }
    if (version == 3) {
        uint64_t a, cnt, len;
        if (!get_varint(buf.data(), buf.size(), pos, 5, a) || !get_varint(buf.data(), buf.size(), pos, 10, cnt) ||
            !get_varint(buf.data(), buf.size(), pos, 5, len) || a != SMAFA_ALPHABET_AA || len == 0 ||
            (buf.size() - pos) / len < cnt)
            return set_error(SMAFA_ERR_FORMAT, "%s: malformed version-3 store", path);
        uint8_t *out = (uint8_t *)malloc(std::max<size_t>(cnt * len, 1));
        if (!out) return set_error(SMAFA_ERR_NOMEM, "out of host memory");
        memcpy(out, buf.data() + pos, cnt * len);
        for (size_t i = 0; i < cnt * len; i++)
            if (out[i] >= 28) {
                free(out);
                return set_error(SMAFA_ERR_FORMAT, "%s: code byte outside the alphabet", path);
            }
        *alphabet = SMAFA_ALPHABET_AA;
        *codes = out;
        *n = cnt;
        *seq_len = (uint32_t)len;
        return SMAFA_OK;
    }
    if (version != SMAFA_DB_VERSION)  // src/lib.rs:215-217
        return set_error(SMAFA_ERR_PANIC,
                         "Unsupported db file version: %llu. This version of smafa only works with version %u "
                         "databases. The last version to support version 1 databases was v0.7.1.",
                         (unsigned long long)version, SMAFA_DB_VERSION);
    const uint8_t *p = buf.data();
    const size_t len = buf.size();
    uint64_t cnt = 0;
    if (!get_varint(p, len, pos, 10, cnt)) return set_error(SMAFA_ERR_FORMAT, "DeserializeUnexpectedEnd");
    // The sequence length comes LAST in the file (Option<NonZeroUsize> after the windows), but the first
    // window's word count nw bounds it: (nw-1)*12 < L <= nw*12.
    uint64_t nw = 0;
    if (cnt) {
        size_t peek = pos;
        if (!get_varint(p, len, peek, 10, nw)) return set_error(SMAFA_ERR_FORMAT, "DeserializeUnexpectedEnd");
        if (nw == 0 || nw > (1u << 26) || cnt > len / nw + 1) return set_error(SMAFA_ERR_FORMAT, "%s: corrupt store", path);
    }
    const size_t stride = (size_t)nw * 12;

    // Pass 1 — where do the windows end, and where can worker threads start?  Every window is
    // varint(nw) + nw varints.  A worker finds a window boundary inside its byte range by local
    // resynchronisation (a byte equal to nw followed by 48 windows that parse), and the pass is accepted only if
    // every worker's walk lands exactly on the next worker's start and the window counts add up to cnt;
    // otherwise (or for small files, or nw = 1 where the trailer byte is ambiguous) one thread walks the file.
    struct Span {
        size_t begin = 0, end = 0;
        uint64_t first = 0, count = 0;
        bool ok = true;
    };
    const size_t windows_begin = pos;
    unsigned n_threads = 1;
    if (cnt >= (1u << 20) && nw >= 2 && nw < 0x80) n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<Span> spans(n_threads);
    auto walk = [&](size_t from, size_t stop, uint64_t max_windows, size_t &at, uint64_t &count) -> bool {
        // skip whole windows from `from` until position >= stop or max_windows reached
        at = from;
        count = 0;
        while (at < stop && count < max_windows) {
            uint64_t k = 0;
            size_t q = at;
            if (!get_varint(p, len, q, 10, k) || k != nw) return false;
            for (uint64_t w = 0; w < nw; w++) {
                int i = 0;
                for (; i < 10 && q < len && (p[q] & 0x80); i++) q++;
                if (i == 10 || q >= len) return false;
                q++;
            }
            at = q;
            count++;
        }
        return true;
    };
    bool parallel_ok = n_threads > 1;
    if (parallel_ok) {
        const size_t chunk = (len - windows_begin) / n_threads;
        std::vector<size_t> starts(n_threads + 1, len);
        starts[0] = windows_begin;
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; t++)
            pool.emplace_back([&, t] {
                const size_t lo = windows_begin + chunk * t;
                for (size_t o = lo; o < lo + 4096 && o + 1 < len; o++) {
                    if (p[o] != (uint8_t)nw) continue;
                    size_t at;
                    uint64_t c;
                    if (walk(o, len, 48, at, c) && c == 48) {
                        starts[t] = o;
                        return;
                    }
                }
                starts[t] = 0;  // no boundary found
            });
        for (auto &th : pool) th.join();
        for (unsigned t = 1; t < n_threads; t++)
            if (starts[t] == 0 || starts[t] <= starts[t - 1]) parallel_ok = false;
        if (parallel_ok) {
            pool.clear();
            for (unsigned t = 0; t < n_threads; t++)
                pool.emplace_back([&, t] {
                    Span &sp = spans[t];
                    sp.begin = starts[t];
                    // the last range is walked afterwards, once the others say how many windows remain for it
                    if (t + 1 < n_threads) sp.ok = walk(sp.begin, starts[t + 1], UINT64_MAX, sp.end, sp.count);
                });
            for (auto &th : pool) th.join();
            uint64_t total = 0;
            for (unsigned t = 0; t + 1 < n_threads; t++) {
                if (!spans[t].ok || spans[t].end != starts[t + 1]) parallel_ok = false;
                total += spans[t].count;
            }
            if (parallel_ok) {
                Span &last = spans[n_threads - 1];
                size_t at = 0;
                uint64_t c = 0;
                if (cnt < total || !walk(last.begin, len, cnt - total, at, c) || c != cnt - total) parallel_ok = false;
                last.end = at;
                last.count = c;
                last.ok = true;
            }
            if (parallel_ok) {
                uint64_t first = 0;
                for (Span &sp : spans) {
                    sp.first = first;
                    first += sp.count;
                }
            }
        }
    }
    if (!parallel_ok) {
        spans.assign(1, Span());
        spans[0].begin = windows_begin;
        if (!walk(windows_begin, len, cnt, spans[0].end, spans[0].count) || spans[0].count != cnt)
            return set_error(SMAFA_ERR_FORMAT, cnt ? "%s: malformed window list" : "DeserializeUnexpectedEnd", path);
        n_threads = 1;
    }
    pos = spans.back().end;
    if (cnt == 0) pos = windows_begin;
    if (pos >= len) return set_error(SMAFA_ERR_FORMAT, "DeserializeUnexpectedEnd");
    const uint8_t tag = p[pos++];
    uint64_t L = 0;
    if (tag == 1) {
        if (!get_varint(p, len, pos, 10, L) || L == 0) return set_error(SMAFA_ERR_FORMAT, "DeserializeBadEncoding");
    } else if (tag != 0) {
        return set_error(SMAFA_ERR_FORMAT, "DeserializeBadOption");
    }
    if (cnt && (L == 0 || (L + 11) / 12 != nw || L > 0xffffffffull))
        return set_error(SMAFA_ERR_FORMAT, "%s: sequence length does not match the window size", path);

    // Pass 2 — decode: each window's words go straight into a row of nw*12 codes through a 32-entry table
    // (one-hot 5 bits -> code; 254 = empty slot, 255 = invalid, src/lib.rs:120-129); columns < L must hold a
    // symbol (the reference's get_as_string panics otherwise), slots past L are ignored.
    static const struct OneHot {
        uint8_t t[32];
        OneHot() {
            for (int i = 0; i < 32; i++) t[i] = 255;
            t[0] = 254;
            t[16] = 0, t[8] = 1, t[4] = 2, t[2] = 3, t[1] = 4;
        }
    } onehot;
    uint8_t *wide = (uint8_t *)malloc(std::max<size_t>((size_t)cnt * stride, 1));
    if (!wide) return set_error(SMAFA_ERR_NOMEM, "out of host memory");
    std::vector<int> bad_symbol(spans.size(), -1);  // per span: offending 5-bit value, or -1
    auto decode = [&](size_t si) {
        const Span &sp = spans[si];
        size_t q = sp.begin;
        for (uint64_t j = sp.first; j < sp.first + sp.count; j++) {
            q++;  // varint(nw): one byte, nw < 128 (or the sequential path's walk verified it)
            if (nw >= 0x80) {  // multi-byte count (only reachable on the one-thread path)
                q--;
                uint64_t k;
                get_varint(p, len, q, 10, k);
            }
            uint8_t *row = wide + (size_t)j * stride;
            for (uint64_t w = 0; w < nw; w++) {
                uint64_t v = 0;
                int i = 0;
                for (;; i++) {  // pass 1 proved every varint terminates within 10 bytes inside the buffer
                    v |= (uint64_t)(p[q + i] & 0x7f) << (7 * i);
                    if (!(p[q + i] & 0x80)) break;
                }
                q += (size_t)i + 1;
                uint8_t *dst = row + w * 12;
                for (int c = 0; c < 12; c++) dst[c] = onehot.t[(v >> (5 * c)) & 31u];
                if (v >> 60) bad_symbol[si] = (int)(v >> 60);  // top 4 bits are never set by from_bytes
            }
            uint8_t worst = 0;
            for (uint64_t i = 0; i < L; i++) worst = row[i] > worst ? row[i] : worst;
            if (worst > 4 && bad_symbol[si] < 0) bad_symbol[si] = worst == 254 ? 0 : 31;
        }
    };
    if (spans.size() > 1) {
        std::vector<std::thread> pool;
        for (size_t si = 0; si < spans.size(); si++) pool.emplace_back(decode, si);
        for (auto &th : pool) th.join();
    } else {
        decode(0);
    }
    // STRICTER than the reference, on purpose ("DB files"): a window group that is not one of the five
    // one-hot codes — or an empty group before column L — fails the load.  The reference would keep such words, count
    // their bits in every distance (src/lib.rs:80-88) and panic only when that subject is printed (src/lib.rs:127);
    // makedb never writes them, so this only concerns damaged or hand-made files, and the text says so.
    for (int b : bad_symbol)
        if (b >= 0) {
            free(wide);
            return set_error(SMAFA_ERR_FORMAT,
                             "%s: a window holds the 5-bit group %u, which is not a one-hot nucleotide code (A=16 C=8 G=4 "
                             "T=2 N=1): damaged store file (this build rejects it at load time; the reference would "
                             "panic \"Invalid character in query sequence\" on printing that subject)",
                             path, (unsigned)b);
        }
    uint8_t *out = wide;
    if (stride != L)  // compact in place (L < stride, ascending rows)
        for (uint64_t j = 0; j < cnt; j++) memmove(out + (size_t)j * L, wide + (size_t)j * stride, L);
    *alphabet = SMAFA_ALPHABET_NT;
    *codes = out;
    *n = cnt;
    *seq_len = (uint32_t)L;
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_dbfile_read");
}

}
