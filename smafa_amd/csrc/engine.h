// engine.h — internal interfaces shared by the device half (engine.hip) and the host half (host/*.cpp)
// of libsmafa_amd.so.  No HIP types here: host files are compiled with g++.
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/smafa_amd.h"

namespace smafa {

// records the message for smafa_last_error() and returns `code`
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// Inside `catch (...)` of an exported function (every int-returning entry point is a function-try-block: no C++ exception
// leaves the C ABI): std::bad_alloc / std::length_error / std::system_error -> SMAFA_ERR_NOMEM, anything else ->
// SMAFA_ERR_INVALID, the message recorded.  (Worker threads do not catch: running out of memory there ends the process,
// as an allocation failure does in the reference — Rust aborts.)
int exception_code(const char *where) noexcept;

// Scan `n_queries` code rows against the store; rows ordered by (query, dist, subject); rows above the
// k-th smallest distance of their query already removed.  max_num_hits: SMAFA_NONE = no k bound.
int scan_to_host(smafa_db *db, const uint8_t *query_codes, uint64_t n_queries, uint32_t max_div, uint32_t max_num_hits,
                 std::vector<smafa_hit> &out);

// Bring the HIP runtime and the device context up (a few hundred ms the first time in a process).  The drivers call it
// on a helper thread while they read and decode their input; failures are ignored here — the first real call reports.
void warm_device(int device);
// device time of the scan kernels of every host-buffer scan on this handle so far, and their launches
void db_life_stats(const smafa_db *db, double *kernel_ms, uint64_t *launches);
// Forget the subjects but keep the handle's device memory, stream and scratch (cluster's per-batch candidate store).
int db_clear(smafa_db *db);
class PackedStore;
// Where the drivers read a subject's symbols from: code rows in host memory (a decoded version-2 file) or the mapped
// bit-plane tiles of a packed store file (host/packed.cpp), decoded row by row — only hit rows are ever read.
struct SubjectRows {
    const uint8_t *codes = nullptr;
    const PackedStore *packed = nullptr;
    uint32_t L = 0;
    int get(uint64_t j, uint8_t *out) const;  // L code bytes of subject j; fails on a damaged packed store
};
// selection rules of src/lib.rs:241-315 (see smafa_select_rows in the public header)
int select_rows(const smafa_hit *hits, uint64_t n_hits, uint64_t n_queries, uint64_t n_subjects,
                const SubjectRows &subjects, uint32_t max_div, uint32_t max_num_hits, uint32_t limit_per_sequence,
                std::vector<smafa_hit> &rows);
// "{q_base + query}\t{subject}\t{distance}\t{subject string}\n" per row (src/lib.rs:292,310) to fd (host/drivers.cpp)
int write_rows_text(const smafa_hit *rows, size_t n, const SubjectRows &subjects, int alphabet, uint32_t q_base, int fd);
// IO / format failures of the FASTX layer become the reference's .expect(what) panic (host/drivers.cpp)
int expect_fastx(int rc, const char *what);
// a store handle whose HBM image comes straight from a mapped packed store file
int db_load_packed(smafa_db **out, int device, const PackedStore &pk);

// ---- a store replicated over several devices behind one handle (host/group.cpp; public: smafa_group_*)
// fn(g) for every member on its own host thread; first failure in member order
int group_on_every_handle(smafa_group *grp, const std::function<int(int)> &fn);
int group_load_packed(smafa_group **out, const int *devices, int ndev, const PackedStore &pk);
smafa_db *group_member(smafa_group *grp, int g);
int group_size(const smafa_group *grp);

// stderr logging of the drivers (host/common.cpp): level 1 = info, 2 = debug
int verbosity();
void log_line(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
double now_seconds();

// alphabet tables (host/alphabet.cpp)
uint8_t code_of(int alphabet, uint8_t byte);  // 255 = outside the alphabet
const uint8_t *code_table(int alphabet);      // the same as a 256-entry table (for loops over many bytes)
char letter_of(int alphabet, uint8_t code);
const char *alphabet_noun(int alphabet);  // "nucleotide" / "amino acid" for the panic text

}  // namespace smafa
