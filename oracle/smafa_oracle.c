/*
 * smafa_oracle.c — CPU restatement of the wwood/smafa v0.8.0 hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY — see smafa_oracle.h.  Every function cites the
 * reference lines it follows (file:line relative to /root/reference).  This is a
 * restatement written from the reference's behaviour, not a translation of its
 * text: data is held in flat arrays, selection uses a counting sort (same order
 * as the reference's tuple sort), and panics become error returns that keep
 * the reference's message text.
 */
#include "smafa_oracle.h"

#include <errno.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static __thread char g_err[1024];

const char *orc_last_error(void) { return g_err; }

static int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}

/* ------------------------------------------------------------------ encoding */

/* src/lib.rs:171-178: one-hot classes; everything else 0 (=> None at :195) */
uint8_t orc_lut_nt(uint8_t b) {
    switch (b) {
    case 'A': case 'a': return 0x10;
    case 'C': case 'c': return 0x08;
    case 'G': case 'g': return 0x04;
    case 'T': case 't': case 'U': case 'u': return 0x02;
    case 'N': case 'W': case 'S': case 'M': case 'K': case 'R': case 'Y': case 'B': case 'D': case 'H': case 'V':
    case '-':
    case 'n': case 'w': case 's': case 'm': case 'k': case 'r': case 'y': case 'b': case 'd': case 'h': case 'v':
        return 0x01;
    default: return 0;
    }
}

size_t orc_words_for(size_t len) { return (len + 11) / 12; /* seq.chunks(12), src/lib.rs:32 */ }

/* src/lib.rs:29-52: symbol i of a chunk goes to bits 5i..5i+4 (:44) */
int orc_encode_onehot(const uint8_t *seq, size_t len, uint64_t *out, size_t *bad_pos) {
    size_t nw = orc_words_for(len);
    for (size_t w = 0; w < nw; w++) out[w] = 0;
    for (size_t p = 0; p < len; p++) {
        uint8_t code = orc_lut_nt(seq[p]);
        if (code == 0) {
            if (bad_pos) *bad_pos = p;
            return -1;
        }
        out[p / 12] |= (uint64_t)code << (5 * (p % 12));
    }
    return 0;
}

static int encode_or_panic(const uint8_t *id, size_t id_len, const uint8_t *seq, size_t len, uint64_t *out) {
    size_t bad = 0;
    if (orc_encode_onehot(seq, len, out, &bad) != 0) {
        /* panic text of src/lib.rs:38-41 */
        return fail("Byte %u cannot be interpreted as nucleotide, in sequence \"%.*s\" at position %zu",
                    (unsigned)seq[bad], (int)id_len, (const char *)id, bad);
    }
    return 0;
}

/* build-defined code alphabets (extension; NT classes identical to the LUT above) */
uint8_t orc_code(int alphabet, uint8_t b) {
    if (alphabet == ORC_ALPHABET_NT) {
        switch (orc_lut_nt(b)) {
        case 0x10: return 0;
        case 0x08: return 1;
        case 0x04: return 2;
        case 0x02: return 3;
        case 0x01: return 4;
        default: return 255;
        }
    }
    if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 'a' + 'A');
    if (b >= 'A' && b <= 'Z') return (uint8_t)(b - 'A');
    if (b == '*') return 26;
    if (b == '-') return 27;
    return 255;
}

char orc_decode(int alphabet, uint8_t code) {
    if (alphabet == ORC_ALPHABET_NT) return code < 5 ? "ACGTN"[code] : '?';
    if (code < 26) return (char)('A' + code);
    if (code == 26) return '*';
    if (code == 27) return '-';
    return '?';
}

/* ----------------------------------------------------------------- WindowSet */

void orc_ws_init(orc_windowset *ws, uint32_t version) {
    memset(ws, 0, sizeof *ws);
    ws->version = version; /* src/lib.rs:63-69 */
}

void orc_ws_free(orc_windowset *ws) {
    free(ws->data);
    memset(ws, 0, sizeof *ws);
}

/* src/lib.rs:91-111 */
int orc_ws_push(orc_windowset *ws, const uint64_t *enc, size_t len) {
    if (ws->len != 0) {
        if (ws->len != len)
            return fail("WindowSet seq length is %zu, got a new sequence of length %zu", ws->len, len);
    } else {
        if (len == 0) return fail("Cannot add empty sequence to WindowSet");
        ws->len = len;
        ws->nw = orc_words_for(len);
    }
    if (ws->n == ws->cap) {
        size_t ncap = ws->cap ? ws->cap * 2 : 1024;
        uint64_t *p = (uint64_t *)realloc(ws->data, ncap * ws->nw * sizeof(uint64_t));
        if (!p) return fail("out of memory");
        ws->data = p;
        ws->cap = ncap;
    }
    memcpy(ws->data + ws->n * ws->nw, enc, ws->nw * sizeof(uint64_t));
    ws->n++;
    return 0;
}

/* src/lib.rs:71-89: sum of popcount(a ^ b) over the words, halved */
int orc_get_distances(const orc_windowset *ws, const uint64_t *enc, size_t len, size_t *distances) {
    if (ws->len != 0 && ws->len != len)
        return fail("Cannot compute distances between seq of length %zu and windows of lengths %zu", len, ws->len);
    const size_t nw = ws->nw;
    for (size_t j = 0; j < ws->n; j++) {
        const uint64_t *w = ws->data + j * nw;
        size_t bits = 0;
        for (size_t k = 0; k < nw; k++) bits += (size_t)__builtin_popcountll(w[k] ^ enc[k]);
        distances[j] = bits / 2;
    }
    return 0;
}

/* src/lib.rs:113-135 */
int orc_get_as_string(const orc_windowset *ws, size_t index, char *out) {
    const uint64_t *w = ws->data + index * ws->nw;
    for (size_t i = 0; i < ws->len; i++) {
        unsigned b = (unsigned)((w[i / 12] >> (5 * (i % 12))) & 31u);
        char c;
        switch (b) {
        case 0x10: c = 'A'; break;
        case 0x08: c = 'C'; break;
        case 0x04: c = 'G'; break;
        case 0x02: c = 'T'; break;
        case 0x01: c = 'N'; break;
        default: return fail("Invalid character in query sequence: %u", b);
        }
        out[i] = c;
    }
    out[ws->len] = 0;
    return 0;
}

/* ------------------------------------------------ postcard wire format (DB v2) */
/*
 * postcard 1.x (Cargo.toml:20) encodes u32/u64/usize as LEB128 varints, a Vec as
 * varint(len) + items, a newtype struct as its field, Option as 0x00 | 0x01+value.
 * WindowSet { version: u32, windows: Vec<SeqEncoding(Vec<u64>)>, len: Option<NonZeroUsize> }
 * (src/lib.rs:54-60) => the layout below; pinned by tests/data/random_3_2.fna.smafadb
 * (02 02 01 c8 10 01 90 21 01 03) and random_3_2_one_repeated.fna.smafadb.
 */
typedef struct {
    uint8_t *p;
    size_t len, cap;
} bytebuf;

static int bb_put(bytebuf *b, uint8_t v) {
    if (b->len == b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        uint8_t *np = (uint8_t *)realloc(b->p, nc);
        if (!np) return -1;
        b->p = np;
        b->cap = nc;
    }
    b->p[b->len++] = v;
    return 0;
}

static int bb_varint(bytebuf *b, uint64_t v) {
    while (v >= 0x80) {
        if (bb_put(b, (uint8_t)(v | 0x80))) return -1;
        v >>= 7;
    }
    return bb_put(b, (uint8_t)v);
}

int orc_db_serialize(const orc_windowset *ws, uint8_t **buf, size_t *len) {
    bytebuf b = {0, 0, 0};
    int rc = bb_varint(&b, ws->version);
    rc |= bb_varint(&b, ws->n);
    for (size_t j = 0; j < ws->n && !rc; j++) {
        rc |= bb_varint(&b, ws->nw);
        for (size_t k = 0; k < ws->nw; k++) rc |= bb_varint(&b, ws->data[j * ws->nw + k]);
    }
    if (ws->len == 0) {
        rc |= bb_put(&b, 0);
    } else {
        rc |= bb_put(&b, 1);
        rc |= bb_varint(&b, ws->len);
    }
    if (rc) {
        free(b.p);
        return fail("out of memory");
    }
    *buf = b.p;
    *len = b.len;
    return 0;
}

/* max_bytes: 5 for u32, 10 for u64/usize (postcard's varint_max) */
static int rd_varint(const uint8_t *buf, size_t len, size_t *pos, int max_bytes, uint64_t *out) {
    uint64_t v = 0;
    for (int i = 0; i < max_bytes; i++) {
        if (*pos >= len) return fail("DeserializeUnexpectedEnd");
        uint8_t byte = buf[(*pos)++];
        v |= (uint64_t)(byte & 0x7f) << (7 * i);
        if (!(byte & 0x80)) {
            if (max_bytes == 5 && v > 0xffffffffull) return fail("DeserializeBadVarint");
            if (i == 9 && byte > 1) return fail("DeserializeBadVarint");
            *out = v;
            return 0;
        }
    }
    return fail("DeserializeBadVarint");
}

int orc_db_deserialize(const uint8_t *buf, size_t len, orc_windowset *ws) {
    size_t pos = 0;
    uint64_t v, n;
    orc_ws_init(ws, 0);
    if (rd_varint(buf, len, &pos, 5, &v)) return -1;
    ws->version = (uint32_t)v;
    if (rd_varint(buf, len, &pos, 10, &n)) return -1;
    size_t nw = 0;
    for (uint64_t j = 0; j < n; j++) {
        uint64_t k;
        if (rd_varint(buf, len, &pos, 10, &k)) goto bad;
        if (j == 0) {
            nw = (size_t)k;
            if (nw == 0 || n > (SIZE_MAX / 8) / nw) {
                fail("corrupt db: bad window size");
                goto bad;
            }
            ws->data = (uint64_t *)malloc((size_t)n * nw * sizeof(uint64_t));
            if (!ws->data) {
                fail("out of memory");
                goto bad;
            }
            ws->cap = (size_t)n;
            ws->nw = nw;
        } else if (k != nw) {
            /* the reference would accept ragged windows and zip-truncate; a file written by
             * makedb never has them (push_encoding enforces equal len, src/lib.rs:92-101) */
            fail("corrupt db: ragged windows");
            goto bad;
        }
        for (size_t w = 0; w < nw; w++) {
            if (rd_varint(buf, len, &pos, 10, &v)) goto bad;
            ws->data[(size_t)j * nw + w] = v;
        }
    }
    ws->n = (size_t)n;
    if (pos >= len) {
        fail("DeserializeUnexpectedEnd");
        goto bad;
    }
    uint8_t tag = buf[pos++];
    if (tag == 0) {
        ws->len = 0;
    } else if (tag == 1) {
        if (rd_varint(buf, len, &pos, 10, &v)) goto bad;
        if (v == 0) {
            fail("DeserializeBadEncoding");
            goto bad;
        }
        ws->len = (size_t)v;
        if (ws->n && orc_words_for(ws->len) != ws->nw) {
            fail("corrupt db: len does not match window size");
            goto bad;
        }
        if (!ws->n) ws->nw = orc_words_for(ws->len);
    } else {
        fail("DeserializeBadOption");
        goto bad;
    }
    return 0;
bad:
    free(ws->data);
    ws->data = NULL;
    ws->n = ws->cap = 0;
    return -1;
}

/* -------------------------------------------------------------- FASTX reader */
/*
 * Stand-in for needletail 0.5 parse_fastx_file (Cargo.toml:27): format sniffed
 * from the first byte ('>' FASTA, '@' FASTQ), gzip via zlib, multi-line FASTA
 * joined with '\n'/'\r' removed, id = the whole header line, 4-line FASTQ.
 * bzip2/xz inputs are refused (no headers in this image).  Pinned only by the
 * reference's small fixtures (subjects.fa has no trailing newline;
 * random_30_4.fq.gz = 4 reads / 120 bases, tests/test_cmdline.rs:194-201).
 */
struct orc_fastx {
    uint8_t *buf;
    size_t len, pos;
    int fastq;
    uint8_t *seq;
    size_t seq_cap;
};

static int slurp(const char *path, uint8_t **out, size_t *out_len) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail("%s: %s", path, strerror(errno));
    uint8_t magic[6] = {0};
    size_t got = fread(magic, 1, 6, f);
    fclose(f);
    if (got >= 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h')
        return fail("%s: bzip2 input is not supported in this build", path);
    if (got >= 6 && magic[0] == 0xfd && magic[1] == '7' && magic[2] == 'z' && magic[3] == 'X' && magic[4] == 'Z')
        return fail("%s: xz input is not supported in this build", path);
    gzFile g = gzopen(path, "rb"); /* transparent for plain files */
    if (!g) return fail("%s: cannot open", path);
    gzbuffer(g, 1 << 20);
    size_t cap = 1 << 20, len = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) {
        gzclose(g);
        return fail("out of memory");
    }
    for (;;) {
        if (len == cap) {
            cap *= 2;
            uint8_t *nb = (uint8_t *)realloc(buf, cap);
            if (!nb) {
                free(buf);
                gzclose(g);
                return fail("out of memory");
            }
            buf = nb;
        }
        size_t want = cap - len;
        if (want > (1u << 30)) want = 1u << 30;
        int r = gzread(g, buf + len, (unsigned)want);
        if (r < 0) {
            free(buf);
            gzclose(g);
            return fail("%s: read error", path);
        }
        if (r == 0) break;
        len += (size_t)r;
    }
    gzclose(g);
    *out = buf;
    *out_len = len;
    return 0;
}

orc_fastx *orc_fastx_open(const char *path) {
    uint8_t *buf;
    size_t len;
    if (slurp(path, &buf, &len)) return NULL;
    if (len == 0) {
        free(buf);
        fail("%s: empty file", path); /* needletail: EmptyFile */
        return NULL;
    }
    if (buf[0] != '>' && buf[0] != '@') {
        fail("%s: not a FASTA/FASTQ file (starts with byte %u)", path, (unsigned)buf[0]);
        free(buf);
        return NULL;
    }
    orc_fastx *r = (orc_fastx *)calloc(1, sizeof *r);
    r->buf = buf;
    r->len = len;
    r->fastq = buf[0] == '@';
    return r;
}

static size_t line_end(const orc_fastx *r, size_t from) {
    const uint8_t *nl = (const uint8_t *)memchr(r->buf + from, '\n', r->len - from);
    return nl ? (size_t)(nl - r->buf) : r->len;
}

static int seq_reserve(orc_fastx *r, size_t n) {
    if (n <= r->seq_cap) return 0;
    size_t nc = r->seq_cap ? r->seq_cap : 256;
    while (nc < n) nc *= 2;
    uint8_t *p = (uint8_t *)realloc(r->seq, nc);
    if (!p) return fail("out of memory");
    r->seq = p;
    r->seq_cap = nc;
    return 0;
}

int orc_fastx_next(orc_fastx *r, const uint8_t **id, size_t *id_len, const uint8_t **seq, size_t *seq_len) {
    /* skip blank lines between records */
    while (r->pos < r->len && (r->buf[r->pos] == '\n' || r->buf[r->pos] == '\r')) r->pos++;
    if (r->pos >= r->len) return 0;
    uint8_t marker = r->fastq ? '@' : '>';
    if (r->buf[r->pos] != marker) return fail("invalid record start (byte %u)", (unsigned)r->buf[r->pos]);
    size_t hs = r->pos + 1, he = line_end(r, hs);
    size_t hlen = he - hs;
    if (hlen && r->buf[hs + hlen - 1] == '\r') hlen--;
    *id = r->buf + hs;
    *id_len = hlen;
    size_t p = he < r->len ? he + 1 : r->len;
    size_t n = 0;
    if (!r->fastq) {
        /* sequence = every line up to the next line that starts with '>' */
        while (p < r->len && r->buf[p] != '>') {
            size_t le = line_end(r, p);
            if (seq_reserve(r, n + (le - p) + 1)) return -1;
            for (size_t i = p; i < le; i++)
                if (r->buf[i] != '\r') r->seq[n++] = r->buf[i];
            p = le < r->len ? le + 1 : r->len;
        }
    } else {
        size_t le = line_end(r, p);
        if (seq_reserve(r, (le - p) + 1)) return -1;
        for (size_t i = p; i < le; i++)
            if (r->buf[i] != '\r') r->seq[n++] = r->buf[i];
        p = le < r->len ? le + 1 : r->len;
        if (p >= r->len || r->buf[p] != '+') return fail("invalid FASTQ record: missing '+' line");
        le = line_end(r, p);
        p = le < r->len ? le + 1 : r->len;
        le = line_end(r, p);
        size_t qn = le - p;
        if (qn && r->buf[p + qn - 1] == '\r') qn--;
        if (qn != n) return fail("invalid FASTQ record: sequence and quality lengths differ");
        p = le < r->len ? le + 1 : r->len;
    }
    if (seq_reserve(r, 1)) return -1;
    r->pos = p;
    *seq = r->seq;
    *seq_len = n;
    return 1;
}

void orc_fastx_close(orc_fastx *r) {
    if (!r) return;
    free(r->buf);
    free(r->seq);
    free(r);
}

/* -------------------------------------------------------------------- makedb */

/* src/lib.rs:137-165 */
int orc_makedb(const char *subject_fasta, const char *db_path) {
    orc_fastx *r = orc_fastx_open(subject_fasta);
    if (!r) return -1; /* .expect("valid path/file of subject fasta") */
    orc_windowset ws;
    orc_ws_init(&ws, ORC_DB_VERSION);
    uint64_t *enc = NULL;
    size_t enc_cap = 0;
    int rc = 0;
    for (;;) {
        const uint8_t *id, *seq;
        size_t id_len, seq_len;
        int got = orc_fastx_next(r, &id, &id_len, &seq, &seq_len);
        if (got < 0) { rc = -1; break; }
        if (!got) break;
        size_t nw = orc_words_for(seq_len);
        if (nw > enc_cap) {
            enc = (uint64_t *)realloc(enc, (nw + 1) * sizeof(uint64_t));
            enc_cap = nw;
        }
        if (encode_or_panic(id, id_len, seq, seq_len, enc)) { rc = -1; break; }
        if (orc_ws_push(&ws, enc, seq_len)) { rc = -1; break; }
    }
    orc_fastx_close(r);
    free(enc);
    if (!rc) {
        uint8_t *buf;
        size_t len;
        rc = orc_db_serialize(&ws, &buf, &len);
        if (!rc) {
            FILE *f = fopen(db_path, "wb"); /* File::create(db_path)? and write_all(..)?, :161-162: an Err, not a panic */
            if (!f) {
                fail("%s: %s", db_path, strerror(errno));
                rc = ORC_ERR_RESULT;
            } else {
                if (fwrite(buf, 1, len, f) != len) {
                    fail("%s: write error", db_path);
                    rc = ORC_ERR_RESULT;
                }
                fclose(f);
            }
            free(buf);
        }
    }
    orc_ws_free(&ws);
    return rc;
}

/* ----------------------------------------------------------------- selection */

/*
 * src/lib.rs:241-315 for one query.  The reference builds (distance, index) tuples
 * and sorts them (:243-250); a counting sort on distance that keeps index order is
 * the same permutation.
 */
int64_t orc_select(const size_t *distances, size_t n, int64_t max_divergence, int64_t max_num_hits,
                   int64_t limit_per_sequence, orc_same_seq_fn same_seq, void *ctx, orc_hit *sel, size_t cap) {
    /* :224 — 1 is a special case, equivalent to None */
    int kmode = (max_num_hits != ORC_NO_LIMIT && max_num_hits != 1);
    int64_t nsel = 0;
    if (n == 0) {
        /* :254 / :255 / :298 all unwrap or index an empty collection */
        return fail("called `Option::unwrap()` on a `None` value");
    }
    if (kmode) {
        size_t maxd = 0;
        for (size_t j = 0; j < n; j++)
            if (distances[j] > maxd) maxd = distances[j];
        size_t *count = (size_t *)calloc(maxd + 2, sizeof(size_t));
        for (size_t j = 0; j < n; j++) count[distances[j] + 1]++;
        for (size_t d = 0; d <= maxd; d++) count[d + 1] += count[d];
        /* count[d] = number of subjects with distance < d */
        size_t max_distance;
        uint32_t k = (uint32_t)max_num_hits;
        if (k > (uint32_t)n) {
            max_distance = maxd; /* :253-254 */
        } else {
            if (k == 0) {
                free(count);
                return fail("index out of bounds: the len is %zu but the index is 4294967295", n); /* :255, (0-1) as usize */
            }
            size_t d = 0;
            while (count[d + 1] < (size_t)k) d++; /* distance of the (k-1)-th tuple, :255 */
            max_distance = d;
        }
        /* rows in (distance, index) order: for each distance value, indices ascending */
        size_t lim_d = max_distance;
        if (max_divergence != ORC_NO_LIMIT && (uint64_t)max_divergence < lim_d) lim_d = (size_t)max_divergence;
        int have_last = 0;
        size_t last = 0;
        uint32_t last_count = 0;
        for (size_t d = 0; d <= lim_d && d <= maxd; d++) {
            if (count[d + 1] == count[d]) continue;
            for (size_t j = 0; j < n; j++) {
                if (distances[j] != d) continue;
                if (limit_per_sequence != ORC_NO_LIMIT) { /* :269-289 */
                    if (have_last && same_seq(last, j, ctx)) {
                        if (last_count >= (uint32_t)limit_per_sequence) continue;
                        last_count++;
                    } else {
                        last_count = 1;
                    }
                    have_last = 1;
                    last = j;
                }
                if ((size_t)nsel < cap) {
                    sel[nsel].subject = (uint32_t)j;
                    sel[nsel].dist = (uint32_t)d;
                }
                nsel++;
            }
        }
        free(count);
    } else {
        size_t min_distance = distances[0]; /* :298 */
        for (size_t j = 1; j < n; j++)
            if (distances[j] < min_distance) min_distance = distances[j];
        if (limit_per_sequence != ORC_NO_LIMIT) /* :301-303 */
            return fail("limit_per_sequence is implemented unless max_num_hits > 1. It can be implemented by "
                        "analogy, just haven't gotten around to it.");
        if (max_divergence == ORC_NO_LIMIT || min_distance <= (uint64_t)max_divergence) { /* :306 */
            for (size_t j = 0; j < n; j++) {
                if (distances[j] != min_distance) continue;
                if ((size_t)nsel < cap) {
                    sel[nsel].subject = (uint32_t)j;
                    sel[nsel].dist = (uint32_t)min_distance;
                }
                nsel++;
            }
        }
    }
    return nsel;
}

/* --------------------------------------------------------------------- query */

static int ws_same_seq(size_t a, size_t b, void *ctx) {
    const orc_windowset *ws = (const orc_windowset *)ctx;
    return memcmp(ws->data + a * ws->nw, ws->data + b * ws->nw, ws->nw * sizeof(uint64_t)) == 0;
}

/* src/lib.rs:198-325 */
int orc_query(const char *db_path, const char *query_fasta, int64_t max_divergence, int64_t max_num_hits,
              int64_t limit_per_sequence, FILE *out) {
    uint8_t *buf;
    size_t len;
    if (slurp(db_path, &buf, &len)) return ORC_ERR_RESULT; /* File::open(db_path)? / read_to_end(..)?, :208-210 */
    if (len < 4) {                              /* &buffer[0..4], :214 */
        free(buf);
        return fail("range end index 4 out of range for slice of length %zu", len);
    }
    size_t pos = 0;
    uint64_t version;
    if (rd_varint(buf, 4, &pos, 5, &version)) { /* postcard::from_bytes(&buffer[0..4])?, :214 */
        free(buf);
        return ORC_ERR_RESULT;
    }
    if (version != ORC_DB_VERSION) { /* :215-217 */
        free(buf);
        return fail("Unsupported db file version: %u. This version of smafa only works with version %u databases. "
                    "The last version to support version 1 databases was v0.7.1.",
                    (unsigned)version, ORC_DB_VERSION);
    }
    orc_windowset ws;
    if (orc_db_deserialize(buf, len, &ws)) { /* postcard::from_bytes(&buffer)?, :218 */
        free(buf);
        return ORC_ERR_RESULT;
    }
    free(buf);
    orc_fastx *r = orc_fastx_open(query_fasta); /* .expect("valid path/file of query fasta"), :221 */
    if (!r) {
        orc_ws_free(&ws);
        return -1;
    }
    size_t *distances = (size_t *)calloc(ws.n ? ws.n : 1, sizeof(size_t)); /* :227 */
    orc_hit *sel = (orc_hit *)malloc((ws.n ? ws.n : 1) * sizeof(orc_hit));
    char *str = (char *)malloc(ws.len + 1);
    uint64_t *enc = NULL;
    size_t enc_cap = 0;
    uint32_t query_number = 0;
    int rc = 0;
    for (;;) {
        const uint8_t *id, *seq;
        size_t id_len, seq_len;
        int got = orc_fastx_next(r, &id, &id_len, &seq, &seq_len);
        if (got < 0) { rc = -1; break; }
        if (!got) break;
        size_t nw = orc_words_for(seq_len);
        if (nw + 1 > enc_cap) {
            enc = (uint64_t *)realloc(enc, (nw + 1) * sizeof(uint64_t));
            enc_cap = nw + 1;
        }
        if (encode_or_panic(id, id_len, seq, seq_len, enc)) { rc = -1; break; }          /* :235 */
        if (orc_get_distances(&ws, enc, seq_len, distances)) { rc = -1; break; }         /* :238 */
        int64_t nsel = orc_select(distances, ws.n, max_divergence, max_num_hits, limit_per_sequence, ws_same_seq,
                                  &ws, sel, ws.n);
        if (nsel < 0) { rc = -1; break; }
        for (int64_t i = 0; i < nsel; i++) {
            if (orc_get_as_string(&ws, sel[i].subject, str)) { rc = -1; break; }
            fprintf(out, "%u\t%u\t%u\t%s\n", query_number, sel[i].subject, sel[i].dist, str); /* :292, :310 */
        }
        if (rc) break;
        query_number++; /* :317 */
    }
    orc_fastx_close(r);
    free(distances);
    free(sel);
    free(str);
    free(enc);
    orc_ws_free(&ws);
    return rc;
}

/* ------------------------------------------------------------------- cluster */

/* a small open-addressing set of fixed-size keys, standing in for HashSet<Vec<u64>> (src/cluster.rs:24) */
typedef struct {
    size_t key_bytes, cap, n;
    uint8_t *keys;
    uint8_t *used;
} keyset;

static uint64_t hash_bytes(const uint8_t *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h ^ (h >> 29);
}

static void ks_init(keyset *s, size_t key_bytes) {
    s->key_bytes = key_bytes;
    s->cap = 1024;
    s->n = 0;
    s->keys = (uint8_t *)malloc(s->cap * key_bytes);
    s->used = (uint8_t *)calloc(s->cap, 1);
}

static void ks_free(keyset *s) {
    free(s->keys);
    free(s->used);
}

static int ks_insert_raw(keyset *s, const uint8_t *key) {
    size_t i = hash_bytes(key, s->key_bytes) & (s->cap - 1);
    while (s->used[i]) {
        if (memcmp(s->keys + i * s->key_bytes, key, s->key_bytes) == 0) return 0;
        i = (i + 1) & (s->cap - 1);
    }
    s->used[i] = 1;
    memcpy(s->keys + i * s->key_bytes, key, s->key_bytes);
    s->n++;
    return 1;
}

/* returns 1 if newly inserted, 0 if already present */
static int ks_insert(keyset *s, const uint8_t *key) {
    if ((s->n + 1) * 2 > s->cap) {
        keyset big = *s;
        big.cap = s->cap * 2;
        big.n = 0;
        big.keys = (uint8_t *)malloc(big.cap * s->key_bytes);
        big.used = (uint8_t *)calloc(big.cap, 1);
        for (size_t i = 0; i < s->cap; i++)
            if (s->used[i]) ks_insert_raw(&big, s->keys + i * s->key_bytes);
        ks_free(s);
        *s = big;
    }
    return ks_insert_raw(s, key);
}

/* src/cluster.rs:13-94 */
int orc_cluster(const char *input_fasta, uint32_t max_divergence, FILE *out) {
    orc_fastx *r = orc_fastx_open(input_fasta);
    if (!r) return -1;
    orc_windowset centroids;
    orc_ws_init(&centroids, 0); /* :22 */
    keyset seen;
    int seen_init = 0;
    size_t seen_len = 0;
    size_t *distances = NULL;
    size_t dist_cap = 0;
    uint64_t *enc = NULL;
    size_t enc_cap = 0;
    char *str = NULL;
    int rc = 0;
    for (;;) {
        const uint8_t *id, *seq;
        size_t id_len, seq_len;
        int got = orc_fastx_next(r, &id, &id_len, &seq, &seq_len);
        if (got < 0) { rc = -1; break; }
        if (!got) break;
        size_t nw = orc_words_for(seq_len);
        if (nw + 1 > enc_cap) {
            enc = (uint64_t *)realloc(enc, (nw + 1) * sizeof(uint64_t));
            enc_cap = nw + 1;
        }
        if (encode_or_panic(id, id_len, seq, seq_len, enc)) { rc = -1; break; } /* :42-43 */
        /* :46-48 — the HashSet key is the Vec<u64>; vectors of different length never compare
         * equal, and a different length panics in get_distances right after, so one key width
         * per run is enough */
        if (!seen_init) {
            ks_init(&seen, (nw ? nw : 1) * sizeof(uint64_t));
            seen_init = 1;
            seen_len = nw;
        }
        if (nw == seen_len) {
            uint64_t zero = 0;
            if (!ks_insert(&seen, nw ? (const uint8_t *)enc : (const uint8_t *)&zero)) continue;
        }
        if (orc_get_distances(&centroids, enc, seq_len, distances)) { rc = -1; break; } /* :51 */
        size_t min_distance = (size_t)max_divergence * 2 + 2;                            /* :54-58 */
        if (centroids.n) {
            min_distance = distances[0];
            for (size_t j = 1; j < centroids.n; j++)
                if (distances[j] < min_distance) min_distance = distances[j];
        }
        size_t assigned = 0;
        if (min_distance <= (size_t)max_divergence) { /* :62-68: first index holding the minimum */
            for (size_t j = 0; j < centroids.n; j++)
                if (distances[j] == min_distance) {
                    assigned = j;
                    break;
                }
        } else { /* :69-74 */
            assigned = centroids.n;
            if (orc_ws_push(&centroids, enc, seq_len)) { rc = -1; break; }
            if (centroids.n > dist_cap) {
                dist_cap = dist_cap ? dist_cap * 2 : 1024;
                distances = (size_t *)realloc(distances, dist_cap * sizeof(size_t));
            }
        }
        if (!str) str = (char *)malloc(centroids.len + 1);
        if (orc_get_as_string(&centroids, assigned, str)) { rc = -1; break; }
        fwrite(seq, 1, seq_len, out); /* :79-84: column 1 is the record's raw sequence bytes */
        fprintf(out, "\t%s\n", str);
    }
    orc_fastx_close(r);
    if (seen_init) ks_free(&seen);
    free(distances);
    free(enc);
    free(str);
    orc_ws_free(&centroids);
    return rc;
}

/* --------------------------------------------------------------------- count */

/* src/lib.rs:378-398; JSON shape pinned by tests/test_cmdline.rs:184-201 */
int orc_count(const char *const *paths, size_t n_paths, FILE *out) {
    bytebuf b = {0, 0, 0};
    char tmp[64];
    bb_put(&b, '[');
    for (size_t i = 0; i < n_paths; i++) {
        orc_fastx *r = orc_fastx_open(paths[i]);
        if (!r) {
            free(b.p);
            return ORC_ERR_RESULT; /* `?`, src/lib.rs:381,385 */
        }
        size_t reads = 0, bases = 0;
        for (;;) {
            const uint8_t *id, *seq;
            size_t id_len, seq_len;
            int got = orc_fastx_next(r, &id, &id_len, &seq, &seq_len);
            if (got < 0) {
                orc_fastx_close(r);
                free(b.p);
                return ORC_ERR_RESULT; /* `?`, src/lib.rs:381,385 */
            }
            if (!got) break;
            reads++;
            bases += seq_len;
        }
        orc_fastx_close(r);
        if (i) bb_put(&b, ',');
        const char *pre = "{\"path\":\"";
        for (const char *c = pre; *c; c++) bb_put(&b, (uint8_t)*c);
        for (const char *c = paths[i]; *c; c++) {
            if (*c == '"' || *c == '\\') bb_put(&b, '\\');
            bb_put(&b, (uint8_t)*c);
        }
        snprintf(tmp, sizeof tmp, "\",\"num_reads\":%zu,\"num_bases\":%zu}", reads, bases);
        for (const char *c = tmp; *c; c++) bb_put(&b, (uint8_t)*c);
    }
    bb_put(&b, ']');
    fwrite(b.p, 1, b.len, out);
    fputc('\n', out);
    free(b.p);
    return 0;
}

/* ------------------------------------------------- array-level entry points */

static int hit_cmp(const void *a, const void *b) {
    const orc_hit *x = (const orc_hit *)a, *y = (const orc_hit *)b;
    if (x->query != y->query) return x->query < y->query ? -1 : 1;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    if (x->subject != y->subject) return x->subject < y->subject ? -1 : 1;
    return 0;
}

int64_t orc_scan_onehot(const uint8_t *subjects_ascii, size_t n, const uint8_t *queries_ascii, size_t q, size_t L,
                        uint32_t max_div, orc_hit *out, size_t cap) {
    orc_windowset ws;
    orc_ws_init(&ws, ORC_DB_VERSION);
    size_t nw = orc_words_for(L);
    uint64_t *enc = (uint64_t *)malloc((nw + 1) * sizeof(uint64_t));
    size_t bad;
    int64_t total = 0;
    for (size_t j = 0; j < n; j++) {
        if (orc_encode_onehot(subjects_ascii + j * L, L, enc, &bad) || orc_ws_push(&ws, enc, L)) {
            total = -1;
            goto done;
        }
    }
    size_t *distances = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
    for (size_t i = 0; i < q; i++) {
        if (orc_encode_onehot(queries_ascii + i * L, L, enc, &bad)) {
            total = -1;
            break;
        }
        orc_get_distances(&ws, enc, L, distances);
        size_t first = (size_t)total;
        for (size_t j = 0; j < n; j++) {
            if (distances[j] <= max_div) {
                if ((size_t)total < cap) {
                    out[total].query = (uint32_t)i;
                    out[total].subject = (uint32_t)j;
                    out[total].dist = (uint32_t)distances[j];
                }
                total++;
            }
        }
        size_t stored_end = (size_t)total < cap ? (size_t)total : cap;
        if (stored_end > first) qsort(out + first, stored_end - first, sizeof(orc_hit), hit_cmp);
    }
    free(distances);
done:
    free(enc);
    orc_ws_free(&ws);
    return total;
}

void orc_distances_codes(const uint8_t *subject_codes, size_t n, const uint8_t *query_codes, size_t L,
                         uint32_t *distances) {
    for (size_t j = 0; j < n; j++) {
        const uint8_t *s = subject_codes + j * L;
        uint32_t d = 0;
        for (size_t c = 0; c < L; c++) d += (s[c] != query_codes[c]);
        distances[j] = d;
    }
}

int64_t orc_scan_codes(const uint8_t *subject_codes, size_t n, const uint8_t *query_codes, size_t q, size_t L,
                       uint32_t max_div, orc_hit *out, size_t cap) {
    uint32_t *distances = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    int64_t total = 0;
    for (size_t i = 0; i < q; i++) {
        orc_distances_codes(subject_codes, n, query_codes + i * L, L, distances);
        size_t first = (size_t)total;
        for (size_t j = 0; j < n; j++) {
            if (distances[j] <= max_div) {
                if ((size_t)total < cap) {
                    out[total].query = (uint32_t)i;
                    out[total].subject = (uint32_t)j;
                    out[total].dist = distances[j];
                }
                total++;
            }
        }
        size_t stored_end = (size_t)total < cap ? (size_t)total : cap;
        if (stored_end > first) qsort(out + first, stored_end - first, sizeof(orc_hit), hit_cmp);
    }
    free(distances);
    return total;
}

typedef struct {
    const uint8_t *codes;
    size_t L;
} codes_ctx;

static int codes_same_seq(size_t a, size_t b, void *ctx) {
    const codes_ctx *c = (const codes_ctx *)ctx;
    return memcmp(c->codes + a * c->L, c->codes + b * c->L, c->L) == 0;
}

int orc_query_codes(int alphabet, const uint8_t *subject_codes, size_t n, const uint8_t *query_codes, size_t q,
                    size_t L, int64_t max_divergence, int64_t max_num_hits, int64_t limit_per_sequence, FILE *out) {
    uint32_t *d32 = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    size_t *distances = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
    orc_hit *sel = (orc_hit *)malloc((n ? n : 1) * sizeof(orc_hit));
    char *str = (char *)malloc(L + 1);
    codes_ctx ctx = {subject_codes, L};
    int rc = 0;
    for (size_t i = 0; i < q && !rc; i++) {
        orc_distances_codes(subject_codes, n, query_codes + i * L, L, d32);
        for (size_t j = 0; j < n; j++) distances[j] = d32[j];
        int64_t nsel = orc_select(distances, n, max_divergence, max_num_hits, limit_per_sequence, codes_same_seq,
                                  &ctx, sel, n);
        if (nsel < 0) {
            rc = -1;
            break;
        }
        for (int64_t k = 0; k < nsel; k++) {
            const uint8_t *s = subject_codes + (size_t)sel[k].subject * L;
            for (size_t c = 0; c < L; c++) str[c] = orc_decode(alphabet, s[c]);
            str[L] = 0;
            fprintf(out, "%zu\t%u\t%u\t%s\n", i, sel[k].subject, sel[k].dist, str);
        }
    }
    free(d32);
    free(distances);
    free(sel);
    free(str);
    return rc;
}

/* src/cluster.rs:13-94 on code bytes; `assigned` (optional) gets the centroid ordinal of each record */
int orc_cluster_codes(int alphabet, const uint8_t *codes, const uint8_t *raw, size_t n, size_t L,
                      uint32_t max_divergence, FILE *out, uint32_t *assigned_out) {
    keyset seen;
    ks_init(&seen, L ? L : 1);
    size_t ncent = 0, cent_cap = 1024;
    uint8_t *cent = (uint8_t *)malloc(cent_cap * (L ? L : 1));
    char *str = (char *)malloc(L + 1);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *rec = codes + i * L;
        if (!ks_insert(&seen, rec)) { /* src/cluster.rs:46-48 */
            if (assigned_out) assigned_out[i] = UINT32_MAX;
            continue;
        }
        size_t min_distance = (size_t)max_divergence * 2 + 2, arg = 0;
        for (size_t j = 0; j < ncent; j++) {
            const uint8_t *s = cent + j * L;
            size_t d = 0;
            for (size_t c = 0; c < L; c++) d += (s[c] != rec[c]);
            if (j == 0 || d < min_distance) { /* strict <: first index with the minimum, :62-68 */
                min_distance = d;
                arg = j;
            }
        }
        size_t assigned;
        if (ncent && min_distance <= max_divergence) {
            assigned = arg;
        } else {
            if (ncent == cent_cap) {
                cent_cap *= 2;
                cent = (uint8_t *)realloc(cent, cent_cap * (L ? L : 1));
            }
            memcpy(cent + ncent * L, rec, L);
            assigned = ncent++;
        }
        if (assigned_out) assigned_out[i] = (uint32_t)assigned;
        if (out) {
            const uint8_t *s = cent + assigned * L;
            for (size_t c = 0; c < L; c++) str[c] = orc_decode(alphabet, s[c]);
            str[L] = 0;
            fwrite(raw + i * L, 1, L, out);
            fprintf(out, "\t%s\n", str);
        }
    }
    ks_free(&seen);
    free(cent);
    free(str);
    return 0;
}

/* ------------------------------------------------------------ cpu_baseline */

int orc_encode_rows(const uint8_t *rows, size_t n, size_t L, uint64_t *out) {
    size_t nw = orc_words_for(L), bad;
    for (size_t i = 0; i < n; i++)
        if (orc_encode_onehot(rows + i * L, L, out + i * nw, &bad))
            return fail("Byte %u cannot be interpreted as nucleotide, in row %zu at position %zu",
                        (unsigned)rows[i * L + bad], i, bad);
    return 0;
}

int orc_ws_from_ascii(orc_windowset *ws, const uint8_t *rows, size_t n, size_t L) {
    orc_ws_init(ws, ORC_DB_VERSION);
    if (n == 0) return 0;
    if (L == 0) return fail("Cannot add empty sequence to WindowSet");
    ws->len = L;
    ws->nw = orc_words_for(L);
    ws->data = (uint64_t *)malloc(n * ws->nw * sizeof(uint64_t));
    if (!ws->data) return fail("out of memory");
    ws->cap = n;
    if (orc_encode_rows(rows, n, L, ws->data)) {
        orc_ws_free(ws);
        return -1;
    }
    ws->n = n;
    return 0;
}

int64_t orc_bench_besthit_onehot(const orc_windowset *ws, const uint64_t *query_enc, size_t q, size_t len,
                                 int64_t max_divergence) {
    size_t *distances = (size_t *)malloc((ws->n ? ws->n : 1) * sizeof(size_t));
    int64_t rows = 0;
    for (size_t i = 0; i < q; i++) {
        orc_get_distances(ws, query_enc + i * ws->nw, len, distances); /* src/lib.rs:238 */
        size_t mn = distances[0];                                      /* :298 */
        for (size_t j = 1; j < ws->n; j++)
            if (distances[j] < mn) mn = distances[j];
        if (max_divergence == ORC_NO_LIMIT || mn <= (uint64_t)max_divergence) /* :306-312 */
            for (size_t j = 0; j < ws->n; j++) rows += (distances[j] == mn);
    }
    free(distances);
    return rows;
}

/*
 * The K branch as the reference runs it (src/lib.rs:242-295), for the cpu_baseline of the k-th modes: per query a fresh
 * Vec of N (distance, index) tuples (:243-247, 16 bytes each), a full comparison sort of it (:250 — `sort()` on tuples: a
 * stable merge sort with the lexicographic compare inlined; restated as a top-down merge sort with insertion-sorted runs,
 * NOT the counting sort orc_select uses for its answers), the k-th tuple's distance (:253-256), and a walk over ALL tuples
 * that counts the rows that would be printed (:261-293; the reference does not stop at the first tuple beyond the bound).
 */
typedef struct {
    size_t d, i;
} orc_pair;

static inline int pair_less(const orc_pair *a, const orc_pair *b) { return a->d < b->d || (a->d == b->d && a->i < b->i); }

static void merge_sort_pairs(orc_pair *a, orc_pair *tmp, size_t n) {
    if (n <= 20) { /* insertion sort of short runs, as the standard library does */
        for (size_t x = 1; x < n; x++) {
            orc_pair v = a[x];
            size_t y = x;
            while (y > 0 && pair_less(&v, &a[y - 1])) {
                a[y] = a[y - 1];
                y--;
            }
            a[y] = v;
        }
        return;
    }
    size_t h = n / 2;
    merge_sort_pairs(a, tmp, h);
    merge_sort_pairs(a + h, tmp, n - h);
    if (!pair_less(&a[h], &a[h - 1])) return; /* already in order */
    memcpy(tmp, a, h * sizeof(orc_pair));
    size_t x = 0, y = h, o = 0;
    while (x < h && y < n) a[o++] = pair_less(&a[y], &tmp[x]) ? a[y++] : tmp[x++];
    while (x < h) a[o++] = tmp[x++];
}

int64_t orc_bench_kmode_onehot(const orc_windowset *ws, const uint64_t *query_enc, size_t q, size_t len,
                               int64_t max_divergence, int64_t max_num_hits) {
    const size_t n = ws->n;
    if (n == 0 || max_num_hits < 2) return 0;
    size_t *distances = (size_t *)malloc(n * sizeof(size_t)); /* :227, allocated once */
    orc_pair *tmp = (orc_pair *)malloc((n / 2 + 1) * sizeof(orc_pair));
    int64_t rows = 0;
    for (size_t qi = 0; qi < q; qi++) {
        orc_get_distances(ws, query_enc + qi * ws->nw, len, distances); /* :238 */
        orc_pair *v = (orc_pair *)malloc(n * sizeof(orc_pair));      /* :243-247: collect() allocates per query */
        for (size_t j = 0; j < n; j++) {
            v[j].d = distances[j];
            v[j].i = j;
        }
        merge_sort_pairs(v, tmp, n); /* :250 */
        size_t max_distance;
        if ((uint64_t)max_num_hits > (uint64_t)n) { /* :253-254 */
            max_distance = 0;
            for (size_t j = 0; j < n; j++)
                if (distances[j] > max_distance) max_distance = distances[j];
        } else {
            max_distance = v[max_num_hits - 1].d; /* :255 */
        }
        for (size_t j = 0; j < n; j++) /* :261-264 */
            if (v[j].d <= max_distance && (max_divergence == ORC_NO_LIMIT || v[j].d <= (uint64_t)max_divergence)) rows++;
        free(v);
    }
    free(tmp);
    free(distances);
    return rows;
}

int64_t orc_bench_besthit_codes(const uint8_t *subject_codes, size_t n, const uint8_t *query_codes, size_t q,
                                size_t L, int64_t max_divergence) {
    uint32_t *distances = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    int64_t rows = 0;
    for (size_t i = 0; i < q; i++) {
        orc_distances_codes(subject_codes, n, query_codes + i * L, L, distances);
        uint32_t mn = distances[0];
        for (size_t j = 1; j < n; j++)
            if (distances[j] < mn) mn = distances[j];
        if (max_divergence == ORC_NO_LIMIT || mn <= (uint64_t)max_divergence)
            for (size_t j = 0; j < n; j++) rows += (distances[j] == mn);
    }
    free(distances);
    return rows;
}
