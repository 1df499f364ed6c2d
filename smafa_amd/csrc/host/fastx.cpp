// fastx.cpp — see fastx.h.  Behaviour the reference relies on and its fixtures pin:
//  * format by first byte: '>' FASTA, '@' FASTQ; anything else (or an empty file) is an error —
//    the reference .expect()s a valid file (src/lib.rs:144,221);
//  * gzip input is decoded (tests/data/random_30_4.fq.gz: 4 reads, 120 bases, test_cmdline.rs:194-201);
//    bzip2/xz are refused here (no headers in this image);
//  * multi-line FASTA is joined, CR/LF dropped; the last record needs no trailing newline
//    (tests/data/subjects.fa); FASTQ is the 4-line form.
#include "fastx.h"

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
    gzFile g = gzopen(path, "rb");  // reads plain files unchanged
    if (!g) return set_error(SMAFA_ERR_IO, "%s: cannot open", path);
    gzbuffer(g, 1u << 20);
    data_.clear();
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
    if (data_.empty()) return set_error(SMAFA_ERR_FORMAT, "valid path/file expected: %s is empty", path);
    if (data_[0] != '>' && data_[0] != '@')
        return set_error(SMAFA_ERR_FORMAT, "valid path/file expected: %s does not start with '>' or '@'", path);
    fastq_ = data_[0] == '@';
    pos_ = 0;
    return SMAFA_OK;
}

int FastxReader::next(FastxRecord &rec) {
    const uint8_t *d = data_.data();
    const size_t n = data_.size();
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
