/*
 * oracle_cli.c — command-line front end of the CPU oracle, with the reference's
 * flag names (src/main.rs:64-116).  TEST INFRASTRUCTURE ONLY (see smafa_oracle.h):
 * tests run it as a black box the way tests/test_cmdline.rs runs the reference
 * binary, and diff the product CLI against it.
 */
#include "smafa_oracle.h"

#include <stdlib.h>
#include <string.h>

static int usage(void) {
    fprintf(stderr, "usage: smafa_oracle makedb -i FASTA -d DB\n"
                    "       smafa_oracle query -d DB -q FASTX [--max-divergence N] [--max-num-hits N] "
                    "[--limit-per-sequence N]\n"
                    "       smafa_oracle cluster -i FASTA -d N\n"
                    "       smafa_oracle count -i FASTX...\n");
    return 2;
}

static int parse_u32(const char *s, int64_t *out) {
    char *end;
    if (!s || !*s) return -1;
    unsigned long long v = strtoull(s, &end, 10);
    if (*end || v > 0xffffffffull || s[0] == '-') return -1;
    *out = (int64_t)v;
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return usage();
    const char *cmd = argv[1];
    const char *input = NULL, *database = NULL, *query = NULL;
    const char *count_paths[256];
    size_t n_count = 0;
    int64_t max_div = ORC_NO_LIMIT, max_hits = ORC_NO_LIMIT, limit = ORC_NO_LIMIT;
    int is_cluster = strcmp(cmd, "cluster") == 0;
    for (int i = 2; i < argc; i++) {
        const char *a = argv[i];
        const char *v = i + 1 < argc ? argv[i + 1] : NULL;
        if (!strcmp(a, "-i") || !strcmp(a, "--input")) {
            if (!strcmp(cmd, "count")) {
                while (i + 1 < argc && argv[i + 1][0] != '-' && n_count < 256) count_paths[n_count++] = argv[++i];
            } else {
                input = v;
                i++;
            }
        } else if (!strcmp(a, "--database") || (!strcmp(a, "-d") && !is_cluster)) {
            database = v;
            i++;
        } else if (!strcmp(a, "--max-divergence") || (!strcmp(a, "-d") && is_cluster)) {
            if (parse_u32(v, &max_div)) return usage();
            i++;
        } else if (!strcmp(a, "-q") || !strcmp(a, "--query")) {
            query = v;
            i++;
        } else if (!strcmp(a, "--max-num-hits")) {
            if (parse_u32(v, &max_hits)) return usage();
            i++;
        } else if (!strcmp(a, "--limit-per-sequence")) {
            if (parse_u32(v, &limit)) return usage();
            i++;
        } else if (!strcmp(a, "-v") || !strcmp(a, "--verbose") || !strcmp(a, "--quiet")) {
            /* logging flags: stderr only, nothing to do */
        } else {
            return usage();
        }
    }
    int rc;
    if (!strcmp(cmd, "makedb")) {
        if (!input || !database) return usage();
        rc = orc_makedb(input, database);
    } else if (!strcmp(cmd, "query")) {
        if (!database || !query) return usage();
        rc = orc_query(database, query, max_div, max_hits, limit, stdout);
    } else if (is_cluster) {
        if (!input) return usage();
        if (max_div == ORC_NO_LIMIT) { /* src/main.rs:43 unwraps the option */
            fprintf(stderr, "called `Option::unwrap()` on a `None` value\n");
            return 101;
        }
        rc = orc_cluster(input, (uint32_t)max_div, stdout);
    } else if (!strcmp(cmd, "count")) {
        if (!n_count) return usage();
        rc = orc_count(count_paths, n_count, stdout);
    } else {
        return usage();
    }
    fflush(stdout);
    if (rc == ORC_ERR_RESULT) { /* main returned Err: "Error: .." on stderr, exit status 1 */
        fprintf(stderr, "Error: %s\n", orc_last_error());
        return 1;
    }
    if (rc) {
        fprintf(stderr, "%s\n", orc_last_error());
        return 101; /* a Rust panic exits with 101 */
    }
    return 0;
}
