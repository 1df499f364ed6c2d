// select.cpp — which scan rows `smafa query` prints (/root/reference/src/lib.rs:241-315).
//
// The reference sorts all N (distance, index) tuples per query and walks them; here the walk runs over
// the already-thresholded, already-ordered hit list the device produced.  The list holds, per query,
// at least every subject within min(max_divergence, k-th smallest distance), ordered by
// (distance, subject) — the prefix of the reference's sorted tuple vector that can be printed at all.
#include <cstring>

#include "../engine.h"

namespace smafa {

int select_rows(const smafa_hit *hits, uint64_t n_hits, uint64_t n_queries, uint64_t n_subjects,
                const SubjectRows &subjects, uint32_t max_div, uint32_t max_num_hits, uint32_t limit_per_sequence,
                std::vector<smafa_hit> &rows) {
    const uint32_t seq_len = subjects.L;
    std::vector<uint8_t> last_row, this_row;  // limit_per_sequence compares decoded subjects
    rows.clear();
    if (n_queries == 0) return SMAFA_OK;
    // src/lib.rs:224 — max_num_hits == 1 means the same as absent
    const bool kmode = max_num_hits != SMAFA_NONE && max_num_hits != 1;
    if (n_subjects == 0)  // :254 / :298 unwrap the max / min of an empty vector
        return set_error(SMAFA_ERR_PANIC, "called `Option::unwrap()` on a `None` value");
    if (kmode && max_num_hits == 0)  // :255 indexes [(0 - 1) as usize]
        return set_error(SMAFA_ERR_PANIC, "index out of bounds: the len is %llu but the index is 4294967295",
                         (unsigned long long)n_subjects);
    if (!kmode && limit_per_sequence != SMAFA_NONE)  // :301-303
        return set_error(SMAFA_ERR_PANIC,
                         "limit_per_sequence is implemented unless max_num_hits > 1. It can be implemented by analogy, "
                         "just haven't gotten around to it.");
    if (limit_per_sequence != SMAFA_NONE && !subjects.codes && !subjects.packed)
        return set_error(SMAFA_ERR_INVALID, "limit_per_sequence needs the subject codes");
    if (limit_per_sequence != SMAFA_NONE) {
        last_row.resize(seq_len);
        this_row.resize(seq_len);
    }
    uint64_t i = 0;
    while (i < n_hits) {
        uint64_t j = i;
        const uint32_t q = hits[i].query;
        while (j < n_hits && hits[j].query == q) j++;
        const uint64_t cnt = j - i;
        if (kmode) {
            // :253-256 — k larger than the store: everything; else the distance of the k-th tuple.
            // Fewer than k rows in the list means fewer than k subjects lie within max_divergence, so the
            // k-th distance exceeds it and max_divergence alone decides.
            uint32_t kth = UINT32_MAX;
            if (!(max_num_hits > (uint32_t)n_subjects) && cnt >= max_num_hits) kth = hits[i + max_num_hits - 1].dist;
            bool have_last = false;
            uint32_t last_count = 0;
            for (uint64_t t = i; t < j; t++) {
                const smafa_hit &h = hits[t];
                if (h.dist > kth || h.dist > max_div) break;  // ordered by distance: nothing further qualifies
                if (limit_per_sequence != SMAFA_NONE) {       // :269-289, adjacent equal strings only
                    const int grc = subjects.get(h.subject, this_row.data());
                    if (grc) return grc;
                    const bool same = have_last && memcmp(last_row.data(), this_row.data(), seq_len) == 0;
                    if (same) {
                        if (last_count >= limit_per_sequence) continue;
                        last_count++;
                    } else {
                        last_count = 1;
                    }
                    have_last = true;
                    last_row.swap(this_row);
                }
                rows.push_back(h);
            }
        } else {
            // :296-313 — all subjects at the minimum distance, if the minimum is within max_divergence
            const uint32_t dmin = hits[i].dist;
            if (dmin <= max_div)
                for (uint64_t t = i; t < j && hits[t].dist == dmin; t++) rows.push_back(hits[t]);
        }
        i = j;
    }
    return SMAFA_OK;
}

}  // namespace smafa

extern "C" int smafa_select_rows(const smafa_hit *hits, uint64_t n_hits, uint64_t n_queries, uint64_t n_subjects,
                                 const uint8_t *subject_codes, uint32_t seq_len, uint32_t max_div,
                                 uint32_t max_num_hits, uint32_t limit_per_sequence, smafa_hit *rows, uint64_t cap,
                                 uint64_t *n_rows) try {
    if (!n_rows || (!hits && n_hits) || (!rows && cap)) return smafa::set_error(SMAFA_ERR_INVALID, "smafa_select_rows: NULL argument");
    std::vector<smafa_hit> out;
    smafa::SubjectRows subjects;
    subjects.codes = subject_codes;
    subjects.L = seq_len;
    const int rc = smafa::select_rows(hits, n_hits, n_queries, n_subjects, subjects, max_div, max_num_hits, limit_per_sequence, out);
    if (rc) return rc;
    *n_rows = out.size();
    if (out.size() > cap)
        return smafa::set_error(SMAFA_ERR_CAPACITY, "row buffer too small: %zu rows needed", out.size());
    if (!out.empty()) memcpy(rows, out.data(), out.size() * sizeof(smafa_hit));
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_select_rows");
}
