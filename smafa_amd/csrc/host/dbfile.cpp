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
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../engine.h"

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

int read_file(const char *path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return set_error(SMAFA_ERR_IO, "%s: %s", path, strerror(errno));
    out.clear();
    std::vector<uint8_t> chunk(1u << 20);
    size_t r;
    while ((r = fread(chunk.data(), 1, chunk.size(), f)) > 0) out.insert(out.end(), chunk.begin(), chunk.begin() + r);
    const bool bad = ferror(f);
    fclose(f);
    if (bad) return set_error(SMAFA_ERR_IO, "%s: read error", path);
    return SMAFA_OK;
}

}  // namespace

}  // namespace smafa

using namespace smafa;

extern "C" {

int smafa_dbfile_write(const char *path, int alphabet, const uint8_t *codes, uint64_t n, uint32_t seq_len) {
    if (!path || (!codes && n)) return set_error(SMAFA_ERR_INVALID, "smafa_dbfile_write: NULL argument");
    std::vector<uint8_t> b;
    if (alphabet == SMAFA_ALPHABET_NT) {
        const size_t nw = ((size_t)seq_len + 11) / 12;
        b.reserve(16 + (size_t)n * (1 + nw * 9));
        put_varint(b, SMAFA_DB_VERSION);
        put_varint(b, n);
        for (uint64_t j = 0; j < n; j++) {
            const uint8_t *row = codes + (size_t)j * seq_len;
            put_varint(b, nw);
            for (size_t w = 0; w < nw; w++) {
                uint64_t word = 0;
                const size_t lim = std::min<size_t>(12, seq_len - w * 12);
                for (size_t i = 0; i < lim; i++) {
                    const uint8_t c = row[w * 12 + i];
                    if (c > 4) return set_error(SMAFA_ERR_INVALID, "code %u is not a nucleotide code", c);
                    word |= (uint64_t)kOneHotOfCode[c] << (5 * i);
                }
                put_varint(b, word);
            }
        }
        if (n == 0) {
            b.push_back(0);  // len: None
        } else {
            b.push_back(1);
            put_varint(b, seq_len);
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
    const bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
    if (fclose(f) != 0 || !ok) return set_error(SMAFA_ERR_IO, "%s: write error", path);
    return SMAFA_OK;
}

int smafa_dbfile_read(const char *path, int *alphabet, uint8_t **codes, uint64_t *n, uint32_t *seq_len) {
    if (!path || !alphabet || !codes || !n || !seq_len) return set_error(SMAFA_ERR_INVALID, "smafa_dbfile_read: NULL argument");
    *codes = nullptr;
    *n = 0;
    *seq_len = 0;
    std::vector<uint8_t> buf;
    int rc = read_file(path, buf);
    if (rc) return rc;
    // src/lib.rs:214 decodes the version from &buffer[0..4]
    if (buf.size() < 4)
        return set_error(SMAFA_ERR_PANIC, "range end index 4 out of range for slice of length %zu", buf.size());
    size_t pos = 0;
    uint64_t version = 0;
    if (!get_varint(buf.data(), 4, pos, 5, version)) return set_error(SMAFA_ERR_FORMAT, "DeserializeUnexpectedEnd");
    if (version == 3) {
        uint64_t a, cnt, len;
        if (!get_varint(buf.data(), buf.size(), pos, 5, a) || !get_varint(buf.data(), buf.size(), pos, 10, cnt) ||
            !get_varint(buf.data(), buf.size(), pos, 5, len) || a != SMAFA_ALPHABET_AA || len == 0 ||
            (buf.size() - pos) / len < cnt)
            return set_error(SMAFA_ERR_FORMAT, "%s: malformed version-3 store", path);
        uint8_t *out = (uint8_t *)malloc(std::max<size_t>(cnt * len, 1));
        if (!out) return set_error(SMAFA_ERR_IO, "out of memory");
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
    // window's word count nw bounds it: (nw-1)*12 < L <= nw*12.  Windows are decoded straight into rows of
    // nw*12 codes through a 32-entry table (one-hot 5 bits -> code; 254 = empty slot, 255 = invalid,
    // src/lib.rs:120-129), then checked against L and compacted if L is not a multiple of 12.
    static const struct OneHot {
        uint8_t t[32];
        OneHot() {
            for (int i = 0; i < 32; i++) t[i] = 255;
            t[0] = 254;
            t[16] = 0, t[8] = 1, t[4] = 2, t[2] = 3, t[1] = 4;
        }
    } onehot;
    uint64_t nw = 0;
    uint8_t *wide = nullptr;
    size_t stride = 0;
    for (uint64_t j = 0; j < cnt; j++) {
        uint64_t k = 0;
        if (!get_varint(p, len, pos, 10, k)) {
            free(wide);
            return set_error(SMAFA_ERR_FORMAT, "DeserializeUnexpectedEnd");
        }
        if (j == 0) {
            nw = k;
            if (nw == 0 || nw > (1u << 26) || cnt > len / nw + 1) return set_error(SMAFA_ERR_FORMAT, "%s: corrupt store", path);
            stride = (size_t)nw * 12;
            wide = (uint8_t *)malloc(std::max<size_t>((size_t)cnt * stride, 1));
            if (!wide) return set_error(SMAFA_ERR_IO, "out of memory");
        } else if (k != nw) {
            free(wide);
            return set_error(SMAFA_ERR_FORMAT, "%s: windows of unequal size", path);
        }
        uint8_t *row = wide + (size_t)j * stride;
        for (uint64_t w = 0; w < nw; w++) {
            uint64_t v = 0;
            if (pos + 10 <= len) {  // fast path: a whole varint is in range, no per-byte bounds checks
                const uint8_t *q = p + pos;
                int i = 0;
                for (; i < 10; i++) {
                    v |= (uint64_t)(q[i] & 0x7f) << (7 * i);
                    if (!(q[i] & 0x80)) break;
                }
                if (i == 10 || (i == 9 && q[9] > 1)) {
                    free(wide);
                    return set_error(SMAFA_ERR_FORMAT, "DeserializeBadVarint");
                }
                pos += (size_t)i + 1;
            } else if (!get_varint(p, len, pos, 10, v)) {
                free(wide);
                return set_error(SMAFA_ERR_FORMAT, "DeserializeUnexpectedEnd");
            }
            if (v >> 60) {  // 12 symbols x 5 bits: the top 4 bits are never set by from_bytes
                free(wide);
                return set_error(SMAFA_ERR_PANIC, "Invalid character in query sequence: %u", (unsigned)(v >> 60));
            }
            uint8_t *dst = row + w * 12;
            for (int i = 0; i < 12; i++) dst[i] = onehot.t[(v >> (5 * i)) & 31u];
        }
    }
    if (pos >= len) {
        free(wide);
        return set_error(SMAFA_ERR_FORMAT, "DeserializeUnexpectedEnd");
    }
    const uint8_t tag = p[pos++];
    uint64_t L = 0;
    if (tag == 1) {
        if (!get_varint(p, len, pos, 10, L) || L == 0) {
            free(wide);
            return set_error(SMAFA_ERR_FORMAT, "DeserializeBadEncoding");
        }
    } else if (tag != 0) {
        free(wide);
        return set_error(SMAFA_ERR_FORMAT, "DeserializeBadOption");
    }
    if (cnt && (L == 0 || (L + 11) / 12 != nw || L > 0xffffffffull)) {
        free(wide);
        return set_error(SMAFA_ERR_FORMAT, "%s: sequence length does not match the window size", path);
    }
    // columns < L must hold a symbol (the reference's get_as_string panics otherwise); slots past L are ignored
    uint8_t *out = wide;
    for (uint64_t j = 0; j < cnt; j++) {
        const uint8_t *row = wide + (size_t)j * stride;
        uint8_t worst = 0;
        for (uint64_t i = 0; i < L; i++) worst = row[i] > worst ? row[i] : worst;
        if (worst > 4) {
            unsigned bad = 0;
            for (uint64_t i = 0; i < L; i++)
                if (row[i] > 4) {
                    bad = row[i] == 254 ? 0u : 31u;
                    break;
                }
            free(wide);
            return set_error(SMAFA_ERR_PANIC, "Invalid character in query sequence: %u", bad);
        }
        if (stride != L) memmove(out + (size_t)j * L, row, L);  // compact in place (L < stride, ascending j)
    }
    if (!out) out = (uint8_t *)malloc(1);
    *alphabet = SMAFA_ALPHABET_NT;
    *codes = out;
    *n = cnt;
    *seq_len = (uint32_t)L;
    return SMAFA_OK;
}

}
