// group.cpp — one subject store replicated over several GPUs of a node behind ONE handle ("queries sharded
// across the handle's devices internally").  The loop being sharded is the reference's per-query loop
// (/root/reference/src/lib.rs:232-318): it carries no state between queries but the running query number, so a batch is
// cut into contiguous blocks, block g is scanned on device g by its own host thread (HIP's current device is per thread),
// and the blocks' rows — each already ordered (query, dist, subject) — are concatenated in block order with their query
// numbers shifted by the block's offset.  No collective, no torch: the replicas never talk to each other.
#include <algorithm>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../engine.h"
#include "packed.h"

struct smafa_group {
    std::vector<smafa_db *> dbs;
    std::vector<int> devices;
    // rows of a scan that ended in SMAFA_ERR_CAPACITY, kept for the caller's "grow and retry" (as smafa_scan_hits does)
    std::vector<smafa_hit> retry_rows;
    std::vector<uint8_t> retry_codes;
    uint64_t retry_nq = 0, generation = 0, retry_generation = 0;
    uint32_t retry_div = 0, retry_k = 0;
    bool retry_valid = false;
    bool poisoned = false;  // an append failed on some member: the replicas may hold different rows, no scan may mix them
};

namespace smafa {

// fn(g) for every member on its own host thread; the first failure in member order, its message carried over
// (smafa_last_error() is per thread)
int group_on_every_handle(smafa_group *grp, const std::function<int(int)> &fn) {
    const int ndev = (int)grp->dbs.size();
    std::vector<int> rcs((size_t)ndev, SMAFA_OK);
    std::vector<std::string> msgs((size_t)ndev);
    auto body = [&](int g) noexcept {
        try {
            rcs[g] = fn(g);
        } catch (...) {  // no exception ends a worker thread (std::terminate): it becomes this member's error code
            rcs[g] = exception_code("a group member's worker thread");
        }
        if (rcs[g]) {
            try {
                msgs[g] = smafa_last_error();
            } catch (...) {
            }
        }
    };
    if (ndev == 1) {
        body(0);
    } else {
        std::vector<std::thread> pool;
        int started = 0, start_rc = SMAFA_OK;
        try {
            pool.reserve((size_t)ndev);
            for (; started < ndev; started++) pool.emplace_back(body, started);
        } catch (...) {  // a thread could not be created: the ones already running are joined before anything propagates
            start_rc = exception_code("starting a group member's worker thread");
        }
        for (auto &th : pool) th.join();
        if (start_rc) return start_rc;
    }
    for (int g = 0; g < ndev; g++)
        if (rcs[g]) return set_error(rcs[g], "%s", msgs[g].c_str());
    return SMAFA_OK;
}

static int group_alloc(smafa_group **out, const int *devices, int ndev) {
    if (!out) return set_error(SMAFA_ERR_INVALID, "smafa_group: out is NULL");
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64) return set_error(SMAFA_ERR_INVALID, "smafa_group: 1 to 64 devices expected");
    smafa_group *grp = new smafa_group();
    grp->devices.assign(devices, devices + ndev);
    grp->dbs.assign((size_t)ndev, nullptr);
    *out = grp;
    return SMAFA_OK;
}

// a group whose members are loaded from an already mapped packed store file (smafa_query_multi maps it once and also
// reads subject strings from it)
int group_load_packed(smafa_group **out, const int *devices, int ndev, const PackedStore &pk) {
    int rc = group_alloc(out, devices, ndev);
    if (rc) return rc;
    smafa_group *grp = *out;
    rc = group_on_every_handle(grp, [&](int g) { return db_load_packed(&grp->dbs[g], grp->devices[g], pk); });
    if (rc) {
        smafa_group_destroy(grp);
        *out = nullptr;
    }
    return rc;
}

smafa_db *group_member(smafa_group *grp, int g) { return grp->dbs[(size_t)g]; }
int group_size(const smafa_group *grp) { return (int)grp->dbs.size(); }

}  // namespace smafa

using namespace smafa;

extern "C" {

int smafa_group_create(smafa_group **out, const int *devices, int ndev, int alphabet, uint32_t seq_len) try {
    int rc = group_alloc(out, devices, ndev);
    if (rc) return rc;
    smafa_group *grp = *out;
    rc = group_on_every_handle(grp, [&](int g) { return smafa_db_create(&grp->dbs[g], grp->devices[g], alphabet, seq_len); });
    if (rc) {
        smafa_group_destroy(grp);
        *out = nullptr;
    }
    return rc;
} catch (...) {
    if (out && *out) {  // a group half built when the exception came
        smafa_group_destroy(*out);
        *out = nullptr;
    }
    return smafa::exception_code("smafa_group_create");
}

int smafa_group_load(smafa_group **out, const int *devices, int ndev, const char *path) try {
    if (!path) return set_error(SMAFA_ERR_INVALID, "smafa_group_load: NULL path");
    if (out) *out = nullptr;
    PackedStore pk;
    int rc = pk.open(path);
    if (rc) return rc;
    return group_load_packed(out, devices, ndev, pk);
} catch (...) {
    if (out && *out) {  // a group half built when the exception came
        smafa_group_destroy(*out);
        *out = nullptr;
    }
    return smafa::exception_code("smafa_group_load");
}

int smafa_group_append(smafa_group *grp, const uint8_t *codes, uint64_t n) try {
    if (!grp || (!codes && n)) return set_error(SMAFA_ERR_INVALID, "smafa_group_append: NULL argument");
    if (grp->poisoned) return set_error(SMAFA_ERR_INVALID, "smafa_group_append: an earlier append failed on some member; the group is unusable");
    grp->generation++;
    // every replica receives the same rows in the same order: subject indices agree across the members
    int rc = group_on_every_handle(grp, [&](int g) { return smafa_db_append(grp->dbs[g], codes, n); });
    if (rc) {  // some replicas may have taken the rows and others not: later scans would mix different stores
        const std::string why = smafa_last_error();
        grp->poisoned = true;
        return set_error(rc, "%s", why.c_str());
    }
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_group_append");
}

int smafa_group_build_index(smafa_group *grp, uint32_t max_divergence) try {
    if (!grp) return set_error(SMAFA_ERR_INVALID, "smafa_group_build_index: NULL handle");
    if (grp->poisoned) return set_error(SMAFA_ERR_INVALID, "smafa_group_build_index: an append failed on some member; the replicas may differ");
    // every replica builds its own (each on its device, side by side); a member whose build fails just keeps scanning
    return group_on_every_handle(grp, [&](int g) { return smafa_db_build_index(grp->dbs[g], max_divergence); });
} catch (...) {
    return smafa::exception_code("smafa_group_build_index");
}

int smafa_group_size(const smafa_group *grp) { return grp ? (int)grp->dbs.size() : 0; }

smafa_db *smafa_group_member(smafa_group *grp, int index) {
    if (!grp || index < 0 || index >= (int)grp->dbs.size()) return nullptr;
    return grp->dbs[(size_t)index];
}

int smafa_group_scan_hits(smafa_group *grp, const uint8_t *query_codes, uint64_t n_queries, uint32_t max_div,
                          uint32_t max_num_hits, smafa_hit *out, uint64_t cap, uint64_t *n_out) try {
    if (!grp || !n_out || (!query_codes && n_queries) || (!out && cap))
        return set_error(SMAFA_ERR_INVALID, "smafa_group_scan_hits: NULL argument");
    if (n_queries > 0xfffffff0ull) return set_error(SMAFA_ERR_INVALID, "too many queries in one batch");
    if (max_num_hits == 0) max_num_hits = SMAFA_NONE;
    if (grp->poisoned)
        return set_error(SMAFA_ERR_INVALID, "smafa_group_scan_hits: an append failed on some member; the replicas may differ");
    const int ndev = (int)grp->dbs.size();
    smafa_db_info_t info;
    int rc = smafa_db_info(grp->dbs[0], &info);
    if (rc) return rc;
    const size_t L = info.seq_len;
    std::vector<smafa_hit> rows;
    if (grp->retry_valid && grp->retry_generation == grp->generation && grp->retry_nq == n_queries && grp->retry_div == max_div &&
        grp->retry_k == max_num_hits && grp->retry_codes.size() == (size_t)n_queries * L &&
        memcmp(grp->retry_codes.data(), query_codes, grp->retry_codes.size()) == 0) {
        rows.swap(grp->retry_rows);  // the retry after SMAFA_ERR_CAPACITY: same request, same store
    } else {
        std::vector<std::vector<smafa_hit>> block((size_t)ndev);
        rc = group_on_every_handle(grp, [&](int g) -> int {
            const uint64_t lo = n_queries * (uint64_t)g / (uint64_t)ndev, hi = n_queries * (uint64_t)(g + 1) / (uint64_t)ndev;
            if (hi == lo || info.n_subjects == 0) return SMAFA_OK;
            int r = scan_to_host(grp->dbs[g], query_codes + (size_t)lo * L, hi - lo, max_div, max_num_hits, block[g]);
            if (r) return r;
            for (smafa_hit &h : block[g]) h.query += (uint32_t)lo;  // the caller's query numbers
            return SMAFA_OK;
        });
        if (rc) return rc;
        size_t total = 0;
        for (const auto &b : block) total += b.size();
        rows.reserve(total);
        for (const auto &b : block) rows.insert(rows.end(), b.begin(), b.end());  // blocks in order: (query, dist, subject) kept
    }
    grp->retry_valid = false;
    std::vector<smafa_hit>().swap(grp->retry_rows);
    *n_out = rows.size();
    if (rows.size() > cap) {
        const size_t need = rows.size();
        grp->retry_rows.swap(rows);
        grp->retry_codes.assign(query_codes, query_codes + (size_t)n_queries * L);
        grp->retry_nq = n_queries;
        grp->retry_div = max_div;
        grp->retry_k = max_num_hits;
        grp->retry_generation = grp->generation;
        grp->retry_valid = true;
        return set_error(SMAFA_ERR_CAPACITY, "hit buffer too small: %zu rows needed, capacity %llu", need, (unsigned long long)cap);
    }
    std::vector<uint8_t>().swap(grp->retry_codes);
    if (!rows.empty()) memcpy(out, rows.data(), rows.size() * sizeof(smafa_hit));
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_group_scan_hits");
}

void smafa_group_destroy(smafa_group *grp) {
    if (!grp) return;
    for (smafa_db *db : grp->dbs) smafa_db_destroy(db);
    delete grp;
}

}  // extern "C"
