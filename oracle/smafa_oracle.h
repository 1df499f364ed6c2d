/*
 * smafa_oracle.h — CPU restatement of the wwood/smafa v0.8.0 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it,
 * and only as the checker.  The product (smafa_amd/) never links, imports or
 * executes anything here and has no CPU fallback.
 *
 * Parity status: PINNED for the nucleotide path — every live golden vector of
 * the reference (src/lib.rs:334-367, src/cluster.rs:102-143,
 * tests/test_cmdline.rs:10-247 and the two v2 DB byte images in tests/data) is
 * checked against this code by tests/test_oracle_golden.py.  The reference is
 * Rust and cannot be built in this image (no rustc/cargo, crates not vendored),
 * so there is no oracle/_ref build.
 * The amino-acid ("code bytes") functions are a build-defined extension: the
 * reference rejects amino-acid letters (src/lib.rs:171-178), so their parity
 * is UNPINNED upstream; they are cross-pinned to the nucleotide path on
 * nucleotide-alphabet inputs (orc_scan_codes == orc_scan_onehot).
 *
 * Citations are file:line relative to /root/reference.
 */
#ifndef SMAFA_ORACLE_H
#define SMAFA_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DB_VERSION 2u /* src/lib.rs:18 */
#define ORC_NO_LIMIT (-1)

/* last error text (the reference panics; we return nonzero and keep the panic text) */
const char *orc_last_error(void);
/* Return codes of the drivers (orc_makedb / orc_query / orc_cluster / orc_count): 0, or
 *   ORC_ERR_PANIC   where the reference panics (panic!, .expect(), .unwrap(), slice index) — the binary exits 101;
 *   ORC_ERR_RESULT  where it returns Err through `?` (File::open / File::create / postcard / everything in count:
 *                   src/lib.rs:161-162,208-210,214,218,381,385) — main prints "Error: .." and the binary exits 1. */
#define ORC_ERR_PANIC (-1)
#define ORC_ERR_RESULT (-2)

/* ---- encoding: src/lib.rs:167-196 (LUT), src/lib.rs:29-52 (12 symbols per u64) ---- */
uint8_t orc_lut_nt(uint8_t byte); /* 0 = not a nucleotide */
/* words needed for a sequence of len symbols: seq.chunks(12) */
size_t orc_words_for(size_t len);
/* returns 0, or -1 and *bad_pos = offending position (src/lib.rs:36-42) */
int orc_encode_onehot(const uint8_t *seq, size_t len, uint64_t *out, size_t *bad_pos);

/* ---- build-defined code-byte alphabets (not in the reference) ---- */
#define ORC_ALPHABET_NT 0 /* codes A=0 C=1 G=2 T/U=3 N(+IUPAC,-)=4 : same classes as the LUT */
#define ORC_ALPHABET_AA 1 /* codes 'A'..'Z' -> 0..25, '*' -> 26, '-' -> 27, case-folded */
/* returns code 0..27, or 255 for a byte outside the alphabet */
uint8_t orc_code(int alphabet, uint8_t byte);
char orc_decode(int alphabet, uint8_t code);

/* ---- WindowSet: src/lib.rs:54-135 ---- */
typedef struct {
    uint32_t version;
    size_t n;     /* windows.len() */
    size_t len;   /* Option<NonZeroUsize>: 0 = None */
    size_t nw;    /* words per window (0 until len known) */
    uint64_t *data; /* n * nw words, contiguous (the reference uses one Vec per window) */
    size_t cap;
} orc_windowset;

void orc_ws_init(orc_windowset *ws, uint32_t version);
void orc_ws_free(orc_windowset *ws);
/* src/lib.rs:91-111 ; returns -1 with the panic text on length mismatch / empty */
int orc_ws_push(orc_windowset *ws, const uint64_t *enc, size_t len);
/* src/lib.rs:71-89 ; distances has ws->n entries */
int orc_get_distances(const orc_windowset *ws, const uint64_t *enc, size_t len, size_t *distances);
/* src/lib.rs:113-135 ; out needs len+1 bytes */
int orc_get_as_string(const orc_windowset *ws, size_t index, char *out);

/* ---- DB file (postcard 1.x wire format of WindowSet): src/lib.rs:161-162, 208-218 ---- */
/* serialise into a malloc'd buffer */
int orc_db_serialize(const orc_windowset *ws, uint8_t **buf, size_t *len);
int orc_db_deserialize(const uint8_t *buf, size_t len, orc_windowset *ws);

/* ---- FASTX reader (needletail's role: src/lib.rs:144,221; src/cluster.rs:28) ---- */
typedef struct orc_fastx orc_fastx;
orc_fastx *orc_fastx_open(const char *path);
/* 1 = record, 0 = EOF, -1 = error.  id/seq point into reader-owned memory valid until next call */
int orc_fastx_next(orc_fastx *r, const uint8_t **id, size_t *id_len, const uint8_t **seq, size_t *seq_len);
void orc_fastx_close(orc_fastx *r);

/* ---- drivers: the crate's pub fns ---- */
/* src/lib.rs:137-165 */
int orc_makedb(const char *subject_fasta, const char *db_path);
/* src/lib.rs:198-325 ; max_divergence/max_num_hits/limit_per_sequence: ORC_NO_LIMIT = None */
int orc_query(const char *db_path, const char *query_fasta, int64_t max_divergence, int64_t max_num_hits,
              int64_t limit_per_sequence, FILE *out);
/* src/cluster.rs:13-94 */
int orc_cluster(const char *input_fasta, uint32_t max_divergence, FILE *out);
/* src/lib.rs:378-398 */
int orc_count(const char *const *paths, size_t n_paths, FILE *out);

/* ---- array-level entry points used by the GPU parity tests and bench cpu_baseline ---- */
typedef struct {
    uint32_t query, subject, dist;
} orc_hit;

/*
 * Reference-faithful scan on ASCII nucleotide rows: encode every row with the
 * 5-bit one-hot code, distance = sum popcount(a^b)/2 (src/lib.rs:80-88), keep
 * (q, j, d) with d <= max_div, ordered by (query, dist, subject) — the order of
 * src/lib.rs:243-250 / 307-311.  Returns the number of hits (may exceed cap; only
 * the first cap are stored), or -1 on an encoding error.
 */
int64_t orc_scan_onehot(const uint8_t *subjects_ascii, size_t n, const uint8_t *queries_ascii, size_t q, size_t L,
                        uint32_t max_div, orc_hit *out, size_t cap);
/* Same contract on code bytes (any alphabet): distance = number of differing columns. */
int64_t orc_scan_codes(const uint8_t *subject_codes, size_t n, const uint8_t *query_codes, size_t q, size_t L,
                       uint32_t max_div, orc_hit *out, size_t cap);
/* all N distances of one query on code bytes */
void orc_distances_codes(const uint8_t *subject_codes, size_t n, const uint8_t *query_codes, size_t L,
                         uint32_t *distances);

/*
 * The selection rules of src/lib.rs:241-315 applied to one query's N distances.
 * Appends the selected (subject, dist) rows in print order to sel (capacity cap);
 * returns the number selected, or -1 for the reference's panics.  `same_seq(a,b,ctx)`
 * tells whether subjects a and b decode to the same string (limit-per-sequence).
 */
typedef int (*orc_same_seq_fn)(size_t a, size_t b, void *ctx);
int64_t orc_select(const size_t *distances, size_t n, int64_t max_divergence, int64_t max_num_hits,
                   int64_t limit_per_sequence, orc_same_seq_fn same_seq, void *ctx, orc_hit *sel, size_t cap);

/* query / cluster on code bytes (amino-acid extension and NT cross-check); same TSV as the reference */
int orc_query_codes(int alphabet, const uint8_t *subject_codes, size_t n, const uint8_t *query_codes, size_t q,
                    size_t L, int64_t max_divergence, int64_t max_num_hits, int64_t limit_per_sequence, FILE *out);
/* raw = the records' original bytes (column 1 of src/cluster.rs:79-84), n rows of L */
int orc_cluster_codes(int alphabet, const uint8_t *codes, const uint8_t *raw, size_t n, size_t L,
                      uint32_t max_divergence, FILE *out, uint32_t *assigned /* optional, n entries, UINT32_MAX = skipped */);

/*
 * bench.py cpu_baseline ("port"): the reference's per-query work on one thread —
 * fill N distances (src/lib.rs:238), min pass (:298), equality pass (:307) —
 * over queries [0,q).  Returns the number of rows that would be printed.
 */
/* bulk helpers for the baseline: encode n ASCII rows of L columns (src/lib.rs:29-52 per row) */
int orc_ws_from_ascii(orc_windowset *ws, const uint8_t *rows, size_t n, size_t L);
int orc_encode_rows(const uint8_t *rows, size_t n, size_t L, uint64_t *out);
int64_t orc_bench_besthit_onehot(const orc_windowset *ws, const uint64_t *query_enc, size_t q, size_t len,
                                 int64_t max_divergence);
/* the K branch's per-query work (src/lib.rs:242-295: tuple vector, full sort, k-th tuple, walk), max_num_hits >= 2 */
int64_t orc_bench_kmode_onehot(const orc_windowset *ws, const uint64_t *query_enc, size_t q, size_t len,
                               int64_t max_divergence, int64_t max_num_hits);
int64_t orc_bench_besthit_codes(const uint8_t *subject_codes, size_t n, const uint8_t *query_codes, size_t q,
                                size_t L, int64_t max_divergence);

#ifdef __cplusplus
}
#endif
#endif
