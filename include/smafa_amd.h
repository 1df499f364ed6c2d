/*
 * smafa_amd.h — C ABI of the MI355X-native smafa scan engine (libsmafa_amd.so).
 *
 * Drop-in boundary for the ONE hot path of wwood/smafa v0.8.0: the fixed-length
 * Hamming scan behind `smafa query` and `smafa cluster`.  The reference has no FFI
 * of its own; each entry point below names the reference code it replaces
 * (file:line relative to the reference tree).  Plain pointers and sizes only — a
 * Rust host binds these 1:1 (see INTEGRATION.md for the `extern "C"` block).
 *
 * Conventions
 *  - every function returns SMAFA_OK (0) or a negative SMAFA_ERR_* code; the text
 *    of the failure (the reference's panic message where there is one) is kept
 *    per thread in smafa_last_error().  Nothing aborts or throws across the ABI.
 *  - sequences cross the boundary as CODE BYTES, one byte per column, row-major
 *    (n rows of seq_len columns), produced by smafa_encode().
 *  - one handle = one owner thread at a time (the reference is single-threaded).
 *  - there is NO CPU fallback: without a HIP device every scan entry point fails
 *    with SMAFA_ERR_DEVICE.
 */
#ifndef SMAFA_AMD_H
#define SMAFA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMAFA_OK 0
#define SMAFA_ERR_INVALID (-1)  /* bad argument */
#define SMAFA_ERR_DEVICE (-2)   /* no HIP device / HIP runtime failure */
#define SMAFA_ERR_CAPACITY (-3) /* caller's hit buffer too small: *n_out = rows needed, grow and retry */
#define SMAFA_ERR_IO (-4)       /* file could not be read / written */
#define SMAFA_ERR_FORMAT (-5)   /* malformed FASTX / DB file, unsupported DB version */
#define SMAFA_ERR_PANIC (-6)    /* input on which the reference panics (message preserved) */
#define SMAFA_ERR_NOMEM (-7)    /* host memory (or threads) ran out inside the call; the handle it was given may only be destroyed */

#define SMAFA_ALPHABET_NT 0 /* A C G T/U N — classes of BYTE_LUT, src/lib.rs:171-178; codes 0..4 */
#define SMAFA_ALPHABET_AA 1 /* build-defined extension: A-Z * - (case-folded), codes 0..27; not in the reference */

#define SMAFA_NONE UINT32_MAX /* "option absent" for max_div / max_num_hits / limit_per_sequence */

#define SMAFA_DB_VERSION 2u /* CURRENT_DB_VERSION, src/lib.rs:18 */

typedef struct smafa_db smafa_db;     /* subject store resident in HBM — WindowSet, src/lib.rs:54-60 */
typedef struct smafa_qset smafa_qset; /* a packed query batch resident in HBM */
typedef struct smafa_group smafa_group; /* one subject store replicated over several GPUs of a node */

/* one scan result row: the (query_number, i, distance) of src/lib.rs:292 / :310 */
typedef struct {
    uint32_t query;
    uint32_t subject;
    uint32_t dist;
} smafa_hit;

typedef struct {
    uint64_t n_subjects;
    uint32_t seq_len;
    int32_t alphabet;
    int32_t device;
    uint32_t planes;          /* bit-planes stored per subject: 5 (AA), 3 (NT), or 2 (NT store without any N) */
    uint32_t words_per_plane; /* ceil(seq_len / 32) */
    uint64_t hbm_bytes;       /* bytes of the packed subject block in HBM */
    uint64_t bytes_per_subject;
} smafa_db_info_t;

/* state of a store's block index (smafa_db_build_index) */
typedef struct {
    int32_t mode;              /* smafa_set_index */
    int32_t current;           /* 1: an index exists and matches the store as it is now */
    uint32_t blocks;           /* column blocks the index holds: bounds up to blocks - 1 */
    uint32_t usable_blocks;    /* ... of which this many have no run of equal keys too long to probe */
    uint32_t max_div_served;   /* largest bound a big batch is answered from the index at (SMAFA_NONE: none) */
    uint32_t probe_launches;   /* scans of this handle's life that were answered from an index */
    uint64_t bytes;            /* HBM the index occupies */
    uint64_t longest_run;      /* most subjects sharing one block's columns exactly */
    double candidates_per_query; /* subjects a query drawn like the store's rows is compared with at max_div_served */
    double build_ms;
} smafa_index_info_t;

/* ------------------------------------------------------------------ library */
const char *smafa_last_error(void);
int smafa_device_count(void); /* 0 when no MI355X is visible; never fails */
/* Progress / timing lines of the drivers on stderr (the reference logs through env_logger, src/lib.rs:206,230,
 * 320-323; src/cluster.rs:33,87-92): 0 = errors only (--quiet), 1 = info (default of the reference), 2 = debug (-v).
 * The library default is 0 so that stderr stays clean for hosts that do not ask. */
void smafa_set_verbosity(int level);
/* Short hash of the kernel sources this library was built from (recorded beside profiles, so a bench run can tell
 * whether a committed counter profile belongs to the binary it is timing). */
const char *smafa_build_id(void);
/* Measurement helper: the device's empirical HBM read-stream rate in GB/s — a trivial sum kernel over
 * `bytes` (use >= 4 GiB: far more than the 256 MiB Infinity Cache), best of a few repetitions and of two read forms
 * (grid-stride default-policy loads; one contiguous span per workgroup with non-temporal loads). */
int smafa_hbm_read_probe(int device, uint64_t bytes, double *gb_per_s);

/* ----------------------------------------------------------------- encoding */
/* Replaces create_lut/BYTE_LUT/encode_single (src/lib.rs:167-196) and the per-byte half of
 * SeqEncodingLength::from_bytes (src/lib.rs:29-52).  On a byte outside the alphabet returns
 * SMAFA_ERR_PANIC and stores its offset in *bad_pos (the reference's panic at src/lib.rs:36-42). */
int smafa_encode(int alphabet, const uint8_t *ascii, uint64_t len, uint8_t *codes, uint64_t *bad_pos);
/* Replaces WindowSet::get_as_string (src/lib.rs:113-135): codes -> "ACGTN" (or the AA letters). */
int smafa_decode(int alphabet, const uint8_t *codes, uint64_t len, char *out);

/* ------------------------------------------------------------ subject store */
/* WindowSet::new (src/lib.rs:63-69) on `device`; seq_len fixes the equal-length invariant of
 * src/lib.rs:91-111 up front (the host checks lengths before it calls append). */
int smafa_db_create(smafa_db **out, int device, int alphabet, uint32_t seq_len);
/* push_encoding x n (src/lib.rs:91-111): packs n rows of code bytes into the HBM bit-plane block.
 * The host buffer is borrowed for the call only. */
int smafa_db_append(smafa_db *db, const uint8_t *codes, uint64_t n);
/* The packed store file: a resident store saved exactly as it lies in HBM — layout tables, bit-plane
 * tiles, subject order, zone words — so that loading it is a memory map and three host-to-device copies instead of the
 * postcard decode of src/lib.rs:208-218 (<= 46 bytes of varints per subject) plus a re-pack.  The file starts with
 * varint(3): the reference rejects it with its own "Unsupported db file version" panic (src/lib.rs:214-217).
 * Written by `smafa makedb --packed`, accepted wherever a DB file is (smafa_query, smafa_dbfile_read). */
int smafa_db_save(smafa_db *db, const char *path);
int smafa_db_load(smafa_db **out, int device, const char *path);
int smafa_db_info(const smafa_db *db, smafa_db_info_t *info);
/* Launch on a caller-owned HIP stream (hipStream_t as void*) instead of the handle's own; NULL restores it. */
int smafa_db_set_stream(smafa_db *db, void *hip_stream);
void smafa_db_destroy(smafa_db *db); /* NULL-safe */

/* ---------------------------------------------------------------- the scan */
/*
 * Replaces WindowSet::get_distances (src/lib.rs:71-89) + the threshold half of the
 * selection in query (src/lib.rs:241-315) / cluster (src/cluster.rs:51-68).
 *
 * Emits every (query, subject, dist) with
 *      dist <= max_div                       (max_div = SMAFA_NONE: no bound)
 *  and dist <= kth(query)                    (max_num_hits = k >= 1: kth = the k-th smallest
 *                                             distance of that query over the whole store, ties
 *                                             included — src/lib.rs:250-256; SMAFA_NONE/0: no bound)
 * ordered by (query, dist, subject) — the reference's print order (src/lib.rs:243-250, 307-311).
 * With k = 1 this is "all subjects at the minimum distance" (src/lib.rs:296-313) and also
 * cluster's argmin with ties to the lowest index (src/cluster.rs:54-68: first row per query).
 * The device may emit rows above kth(query) (its threshold only ever tightens); those are
 * removed before this call returns.
 * cap = capacity of `out` in rows.  If more rows qualify: SMAFA_ERR_CAPACITY, *n_out = rows needed; the handle
 * keeps those rows, and the same call repeated with a larger buffer (same query bytes and bounds, store
 * unchanged) is answered from them without a second scan.  Any other call to smafa_scan_hits drops them.
 */
int smafa_scan_hits(smafa_db *db, const uint8_t *query_codes, uint64_t n_queries, uint32_t max_div,
                    uint32_t max_num_hits, smafa_hit *out, uint64_t cap, uint64_t *n_out);

/* The literal get_distances seam (src/lib.rs:71-89): all N distances of ONE query, as u32.
 * Costs N*4 bytes over PCIe per query — for tests and debugging, not the production path. */
int smafa_distances(smafa_db *db, const uint8_t *query_codes, uint32_t *distances);

/* ---- device-resident batch form (what bench.py times; inputs already in HBM) ---- */
int smafa_qset_create(smafa_qset **out, smafa_db *db, const uint8_t *query_codes, uint64_t n_queries);
void smafa_qset_destroy(smafa_qset *qs);
/*
 * Asynchronous scan of a resident query set against the resident store on the handle's stream.
 * d_hits: device buffer of cap smafa_hit rows (unordered on return); d_count: device uint64 that
 * receives the number of qualifying rows — exact at any capacity with a fixed bound (max_num_hits absent): it may
 * exceed cap, only the first cap rows to arrive are stored, and a caller can size its buffer from it.  In the
 * tightening modes (max_num_hits = k) a value above cap only says "did not fit".  A fixed-bound scan is ONE kernel
 * launch (sets of up to 64 queries: a one-workgroup kernel that zeroes *d_count in front of it); nothing has to be reset between calls.  (The first scan after a store has grown by a quarter through many
 * appends first sorts it again on the device and waits for that — milliseconds, outside the timed kernel.)
 * max_div / max_num_hits as in smafa_scan_hits, except that rows above kth(query) may remain: rows are kept
 * when dist <= the device's final bound of their query, which is exact for k = 1 and >= kth(query) for k >= 2.
 */
int smafa_scan_launch(smafa_db *db, smafa_qset *qs, uint32_t max_div, uint32_t max_num_hits, void *d_hits,
                      uint64_t cap, void *d_count);
/* One store pass PER QUERY of a resident set — north_star's literal "each query is broadcast against all subjects" — all
 * enqueued back to back by this one call (no host round trip between passes): query i's rows go to
 * d_hits + i * cap_per_query (rows), its exact count to the i-th uint64 of d_counts (zeroed once for all passes, then each
 * pass reserves its rows from its own counter).  Each pass is the
 * fixed-bound scan smafa_scan_launch runs for a one-query set; with the zone level off (smafa_set_zone_level 0) it
 * streams the prefilter's whole bit-plane: the HBM-bound form bench.py's `stream` leg times.  use_graph != 0: the passes
 * are captured once as a HIP graph and replayed while the arguments stay the same.  Replaces the same loop as
 * smafa_scan_launch (src/lib.rs:232-318 around get_distances, :238), one query per iteration as the reference runs it. */
int smafa_scan_each(smafa_db *db, smafa_qset *qs, uint32_t max_div, void *d_hits, uint64_t cap_per_query, void *d_counts,
                    int use_graph);
int smafa_sync(smafa_db *db);
/* Device time in ms of the scan kernel(s) of the most recent smafa_scan_launch / smafa_scan_hits on
 * this handle, from HIP events recorded on the launch stream; also how many kernel launches it took. */
int smafa_last_scan_ms(smafa_db *db, float *ms, uint32_t *n_launches);
/* Totals over the most recent smafa_scan_hits call on this handle (a best-hit or k-th call without a tight bound is
 * several scans: the near-hit ladder, then the tightening path): device time of the scan kernels, kernel launches, scans. */
int smafa_last_call_stats(smafa_db *db, float *kernel_ms, uint32_t *n_launches, uint32_t *n_scans);
/* Where this handle's scan kernels were really launched: the HIP device that was current on the launching host thread at
 * the most recent launch (-1: none yet), and how many launches of the handle's life were issued while another device than
 * smafa_db_info().device was current (must stay 0).  A multi-device caller — smafa_group_*, smafa_cluster_multi, one host
 * thread per member — checks with this that replica g really runs on devices[g] and not silently on device 0. */
int smafa_launch_device(smafa_db *db, int *device_at_last_launch, uint64_t *launches_off_device);
/* How the most recent scan kernel launch on this handle was laid out: whether it kept only the prefilter's plane
 * of each subject resident (then a sparse-hit scan streams words_per_plane*4 bytes per subject instead of
 * bytes_per_subject), wave tiles per wave, and query blocks (= passes over the store). */
int smafa_last_scan_plan(smafa_db *db, uint32_t *filter_plane_resident, uint32_t *tiles_per_wave, uint32_t *query_blocks);
/* Name of the scan kernel instantiation the most recent launch on this handle used, spelled the way rocprofv3
 * lists it (e.g. "smafa::scan_lazy_kernel<5, 5, 2, 4, false>"), so that a bench line and a profile can be matched. */
int smafa_last_scan_kernel(smafa_db *db, char *name, uint64_t cap);
/* Tuning knob: queries per workgroup pass (0 = automatic). */
int smafa_set_query_block(smafa_db *db, uint32_t queries_per_block);
/* 1 (default): the scan evaluates an exact lower bound first and runs the full comparison only where it can
 * still qualify; 0: every (query, subject) pair gets the full comparison.  Results are identical either way. */
int smafa_set_prefilter(smafa_db *db, int enabled);
/* The zone level of a sorted store (scan_zone_kernel: a lower bound on the distances of all 256 subjects of a wave tile
 * at once, from the filter bits they share): 1 (default) = where the store's sorted runs make it prune, 0 = never (every
 * pass streams the prefilter's plane: the HBM-bound form), 2 = whenever the filter-plane-resident kernel runs.  Results
 * are identical in every mode. */
int smafa_set_zone_level(smafa_db *db, int mode);
/* The block index of a resident store — for callers that scan the same store many times with a tight fixed bound (a service,
 * `smafa cluster`'s batches).  The L columns are cut into max_divergence + 1 disjoint blocks; a subject within d <= max_divergence
 * of a query agrees with it on every column of at least one of any d + 1 blocks (it has at most d mismatching columns), so a
 * fixed-bound scan probes d + 1 blocks per query in per-block sorted key arrays and compares in full only the subjects that
 * share a block with the query: the rows of get_distances (src/lib.rs:71-89) + the bound test of :252/:299 without visiting
 * the other subjects.  Exact — the rows are those of the scan kernels, in the same unordered list.  smafa_scan_launch /
 * smafa_scan_hits / the group and session calls use it by themselves when it is current (no append, re-sort or re-plane since
 * it was built), the bound is within it, the batch has more than 64 queries and the store's blocks are selective enough
 * (smafa_index_info().max_div_served; dense families and low-complexity columns are left to the scan kernels); otherwise
 * they scan as before.  Costs 8 bytes x blocks per subject of HBM and about a millisecond per block and 10M subjects to build.
 * Rows of up to 128 columns. */
int smafa_db_build_index(smafa_db *db, uint32_t max_divergence);
int smafa_db_drop_index(smafa_db *db);
int smafa_index_info(const smafa_db *db, smafa_index_info_t *info);
/* 0: a built index is never used; 1 (default): used where it pays; 2: also BUILT by the first big fixed-bound scan that could
 * use one (a synchronous build inside that call); 3: built once the scans that could have used one have cost as much kernel
 * time as the build would (rent or buy: at most twice the best choice in hindsight).  Results are identical in every mode. */
int smafa_set_index(smafa_db *db, int mode);

/* ------------------------------------------------- the same store on several GPUs */
/*
 * "queries sharded across the handle's devices internally".  A group is ONE subject store replicated on every
 * entry of `devices` (an entry may repeat: several handles on one GPU); what is sharded is the reference's per-query loop,
 * src/lib.rs:232-318, around get_distances (:238) — it carries no state between queries but the running query number.
 * smafa_group_scan_hits has the contract of smafa_scan_hits (same bounds, same order, same grow-and-retry): the batch
 * is cut into ndev contiguous blocks, block g is scanned on devices[g] by its own host thread, and the blocks' rows are
 * concatenated in block order, so the rows do not depend on ndev.  No collective: replicas never exchange anything.
 * One group = one owner thread at a time.  smafa_query_multi is a caller of this.
 */
int smafa_group_create(smafa_group **out, const int *devices, int ndev, int alphabet, uint32_t seq_len);
/* every member from the same packed store file (smafa_db_load per device; the file is mapped once) */
int smafa_group_load(smafa_group **out, const int *devices, int ndev, const char *path);
/* push_encoding x n on every replica (src/lib.rs:91-111); subject indices agree across the members */
int smafa_group_append(smafa_group *grp, const uint8_t *codes, uint64_t n);
int smafa_group_scan_hits(smafa_group *grp, const uint8_t *query_codes, uint64_t n_queries, uint32_t max_div,
                          uint32_t max_num_hits, smafa_hit *out, uint64_t cap, uint64_t *n_out);
/* smafa_db_build_index on every member, side by side (each replica keeps its own index on its device) */
int smafa_group_build_index(smafa_group *grp, uint32_t max_divergence);
int smafa_group_size(const smafa_group *grp);
/* member `index` (borrowed; owned by the group): for smafa_db_info, the tuning knobs, or device-resident launches.
 * Rows are appended through smafa_group_append only (the replicas must stay identical); never destroy a member. */
smafa_db *smafa_group_member(smafa_group *grp, int index);
void smafa_group_destroy(smafa_group *grp); /* NULL-safe */

/* -------------------------------------------------------- host-side selection */
/*
 * The row-selection rules of query (src/lib.rs:241-315) applied to a hit list already ordered by
 * (query, dist, subject) that holds, per query, at least every subject within min(max_div, kth).
 * subject_codes (n_subjects rows of seq_len) is needed only for limit_per_sequence (adjacent equal
 * strings, src/lib.rs:269-289).  n_queries = number of queries the hit list covers (query ids
 * 0..n_queries-1); a query with no hit at all while max_div is SMAFA_NONE means an empty store.
 * Returns SMAFA_ERR_PANIC for the inputs the reference panics on (empty store, k = 0,
 * limit_per_sequence without max_num_hits > 1).  Output rows are in print order.
 */
int smafa_select_rows(const smafa_hit *hits, uint64_t n_hits, uint64_t n_queries, uint64_t n_subjects,
                      const uint8_t *subject_codes, uint32_t seq_len, uint32_t max_div, uint32_t max_num_hits,
                      uint32_t limit_per_sequence, smafa_hit *rows, uint64_t cap, uint64_t *n_rows);

/* Print rows the way query does (src/lib.rs:292,310): "{query_offset + query}\t{subject}\t{dist}\t{subject string}\n"
 * per row, to out_fd.  For hosts that select rows themselves (the multi-GPU driver on rank 0). */
int smafa_write_rows(const smafa_hit *rows, uint64_t n_rows, const uint8_t *subject_codes, uint64_t n_subjects,
                     uint32_t seq_len, int alphabet, uint32_t query_offset, int out_fd);

/* ------------------------------------------------------------------ DB file */
/* Serialise / parse the reference's v2 DB file (postcard wire format of WindowSet,
 * src/lib.rs:54-60,161-162,208-218).  NT only.  smafa_dbfile_write: codes -> file bytes identical to
 * the reference's makedb.  smafa_dbfile_read: file -> malloc'd code rows (free with smafa_free);
 * a version other than 2 fails with the reference's "Unsupported db file version" text.
 * Version 3 is this build's extension container for amino-acid stores (alphabet byte + raw code rows). */
int smafa_dbfile_write(const char *path, int alphabet, const uint8_t *codes, uint64_t n, uint32_t seq_len);
int smafa_dbfile_read(const char *path, int *alphabet, uint8_t **codes, uint64_t *n, uint32_t *seq_len);
/* Read a FASTA/FASTQ(+gzip) file of equal-length records into malloc'd code rows (free with smafa_free) —
 * parse_fastx_file + from_bytes per record (src/lib.rs:221,235).  Fails like the reference on a byte outside
 * the alphabet (message of src/lib.rs:38-41); records of unequal length fail with SMAFA_ERR_PANIC. */
int smafa_fastx_load(const char *path, int alphabet, uint8_t **codes, uint64_t *n, uint32_t *seq_len);
/* The same, stopping quietly at the first offending record: returns the rows before it, and in *pending the code
 * smafa_fastx_load would have failed with (message in smafa_last_error()) or SMAFA_OK.  For hosts that — like the
 * reference's loop, src/lib.rs:232-318 — answer the queries in front of a bad record before they fail. */
int smafa_fastx_load_partial(const char *path, int alphabet, uint8_t **codes, uint64_t *n, uint32_t *seq_len, int *pending);
/* One PART of a plain FASTA/FASTQ file, for hosts that shard a query file over processes: the records that start in this
 * part's byte range (part p of `parts`: the first record start at or after byte size*p/parts up to the next part's) — the
 * parts partition the records in file order and a process reads only its own bytes.  *usable = 0 (and no rows): the file
 * cannot be taken in parts (gzip, a cut that did not hold, a malformed record) — load the whole file instead.  *seq_len =
 * the length of the part's first record; *pending as in smafa_fastx_load_partial.  (parse_fastx_file + the loop of
 * src/lib.rs:221,232-235, one share of it.) */
int smafa_fastx_load_part(const char *path, int alphabet, uint32_t part, uint32_t parts, uint8_t **codes, uint64_t *n,
                          uint32_t *seq_len, int *pending, int *usable);
void smafa_free(void *p);

/* ------------------------------------------- drivers: the crate's pub fns */
/* makedb(subject_fasta, db_path) — src/lib.rs:137-165.  Host only (no GPU needed). */
int smafa_makedb(const char *subject_fasta, const char *db_path, int alphabet);
/* makedb with the packed store file as output: the subjects are packed on `device` (the layout is the one a query
 * would build) and saved with smafa_db_save.  device < 0, or no GPU visible: packed by host threads instead — the same
 * bytes (host/layout.cpp restates the device's packing; the two files are compared in tests/test_gpu_layout.py). */
int smafa_makedb_packed(const char *subject_fasta, const char *db_path, int alphabet, int device);
/* query(db_path, query_fasta, max_divergence, max_num_hits, limit_per_sequence) — src/lib.rs:198-325.
 * Options use SMAFA_NONE for None.  TSV rows go to out_fd (the reference prints to stdout). */
int smafa_query(const char *db_path, const char *query_fasta, uint32_t max_divergence, uint32_t max_num_hits,
                uint32_t limit_per_sequence, int out_fd, int device);
/* The same query spread over several GPUs of one node by ONE process ("queries sharded across the handle's
 * devices internally"): one handle per entry of `devices` (an entry may repeat: several handles on one GPU), the store
 * replicated on each, every chunk of queries cut into ndev contiguous blocks scanned by one host thread per handle,
 * rows printed in block order.  The loop being sharded is src/lib.rs:232-318, which carries no state between queries
 * but the running query number, so the bytes written do not depend on ndev.  No collective, no torch. */
int smafa_query_multi(const char *db_path, const char *query_fasta, uint32_t max_divergence, uint32_t max_num_hits,
                      uint32_t limit_per_sequence, int out_fd, const int *devices, int ndev);
/* ---- `query` for hosts that run ONE PROCESS PER GPU (smafa_amd/dist.py over torch.distributed / RCCL, or MPI) ----
 * The loop of src/lib.rs:232-318 carries no state between records but the running query number, so the query FILE is cut
 * into `parts` contiguous shares in rank order.  A process opens the DB once (a packed store file is mapped and copied to
 * HBM: no decode, no host code rows), answers its share — reading ONLY its byte range of a plain FASTA/FASTQ file — and
 * rank 0 prints the gathered rows, decoding subject strings for the hit rows only.
 *   smafa_qsession_scan_part: rows (malloc'd, free with smafa_free) are numbered from 0 within the share and already
 *   selected (src/lib.rs:241-315); *n_queries = records of the share that were answered; *pending = SMAFA_OK or the code of
 *   the first record the reference's loop fails on inside this share (text in smafa_last_error()): the caller prints the
 *   rows of the ranks up to and including the first such rank, then fails with that text.
 *   whole_file = 0: the share is the records that start in this part's byte range; *n_before = UINT64_MAX (the caller
 *   adds up the counts of the ranks in front); *retry_whole = 1 (and nothing else) when the file cannot be taken in parts
 *   (gzip, a cut that did not hold, a malformed record): EVERY rank must then call again with whole_file = 1 — the whole
 *   file is parsed, the share is block [part*Q/parts, (part+1)*Q/parts) of its Q usable records and *n_before its start. */
typedef struct smafa_qsession smafa_qsession;
int smafa_qsession_open(smafa_qsession **out, const char *db_path, int device); /* device < 0: no store in HBM (printing only) */
int smafa_qsession_info(const smafa_qsession *s, uint64_t *n_subjects, uint32_t *seq_len, int *alphabet);
int smafa_qsession_scan_part(smafa_qsession *s, const char *query_fasta, uint32_t max_divergence, uint32_t max_num_hits,
                             uint32_t limit_per_sequence, uint32_t part, uint32_t parts, int whole_file, smafa_hit **rows,
                             uint64_t *n_rows, uint64_t *n_queries, uint64_t *n_before, int *pending, int *retry_whole);
int smafa_qsession_write(smafa_qsession *s, const smafa_hit *rows, uint64_t n_rows, int out_fd);
void smafa_qsession_close(smafa_qsession *s); /* NULL-safe */
/* cluster(input_fasta, max_divergence, print_stream) — src/cluster.rs:13-94. */
int smafa_cluster(const char *input_fasta, uint32_t max_divergence, int out_fd, int device, int alphabet);
/* The same clustering spread over `world` processes, one GPU each (cluster mode): every rank reads
 * the whole input and keeps a replica of the centroid store on its own device; each batch of unseen records is
 * split into contiguous slices, rank r scans slice r, and the per-record results are exchanged through
 * `allgather` twice per batch (nearest old centroid: 8 bytes per record; in-range (record, candidate) rows:
 * 12 bytes each).  Every rank then resolves the batch identically and appends the same new centroids to its
 * replica — there is no broadcast.  Rank 0 writes the reference's output bytes to out_fd; other ranks write
 * nothing.  The result does not depend on `world`.
 * `allgather(ctx, send, send_bytes, &recv, &recv_bytes)` must return 0 and leave in *recv the blocks of ALL
 * ranks concatenated in rank order (block sizes differ between ranks); the buffer stays valid until the next
 * call.  The library never touches the transport: the caller supplies RCCL (torch.distributed in
 * smafa_amd/dist.py), MPI, ... */
typedef int (*smafa_allgather_fn)(void *ctx, const void *send, uint64_t send_bytes, const void **recv,
                                  uint64_t *recv_bytes);
int smafa_cluster_sharded(const char *input_fasta, uint32_t max_divergence, int out_fd, int device, int alphabet,
                          uint32_t rank, uint32_t world, smafa_allgather_fn allgather, void *ctx);
/* The same clustering over several GPUs of one node by ONE process (no torch, no collective library): one host thread per entry
 * of `devices` (entries may repeat) plays a rank of smafa_cluster_sharded — its own replica of the centroid store, its slice of
 * every batch — the input is parsed and de-duplicated once for all of them, and the two exchanges per batch go through memory.
 * Output bytes do not depend on ndev (src/cluster.rs:13-94 semantics). */
int smafa_cluster_multi(const char *input_fasta, uint32_t max_divergence, int out_fd, const int *devices, int ndev, int alphabet);
/* count(paths) — src/lib.rs:378-398 (JSON to out_fd).  Host only. */
int smafa_count(const char *const *paths, uint64_t n_paths, int out_fd);

#ifdef __cplusplus
}
#endif
#endif
