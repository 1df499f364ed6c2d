// Sanitizer harness for the per-part FASTX loader (smafa_fastx_load_part): every part of every file named on the command
// line, for several part counts; the parts that are usable must add up to the whole file's usable records.
//   asan_parts FILE...      (built and run by tools/asan_host.sh)
#include <cstdio>
#include <cstdlib>
#include <initializer_list>

#include "../include/smafa_amd.h"

int main(int argc, char **argv) {
    int bad = 0;
    for (int f = 1; f < argc; f++) {
        uint8_t *whole = nullptr;
        uint64_t n_whole = 0;
        uint32_t L = 0;
        int pending = 0;
        const int wrc = smafa_fastx_load_partial(argv[f], SMAFA_ALPHABET_NT, &whole, &n_whole, &L, &pending);
        smafa_free(whole);
        for (uint32_t parts : {1u, 2u, 3u, 7u, 50u}) {
            uint64_t sum = 0;
            bool usable_all = true, stopped = false;
            for (uint32_t p = 0; p < parts; p++) {
                uint8_t *codes = nullptr;
                uint64_t n = 0;
                uint32_t l = 0;
                int pend = 0, usable = 0;
                const int rc = smafa_fastx_load_part(argv[f], SMAFA_ALPHABET_NT, p, parts, &codes, &n, &l, &pend, &usable);
                smafa_free(codes);
                if (rc != SMAFA_OK || !usable) usable_all = false;
                if (!stopped) sum += n;
                if (pend) stopped = true;  // records behind the first bad one do not count
            }
            // (a length mismatch between parts is the caller's to detect: it knows the store's length — so only files
            // without a pending error are compared)
            if (usable_all && wrc == SMAFA_OK && pending == 0 && sum != n_whole) {
                printf("MISMATCH %s parts %u: %llu vs %llu\n", argv[f], parts, (unsigned long long)sum, (unsigned long long)n_whole);
                bad++;
            }
        }
    }
    return bad ? 1 : 0;
}
