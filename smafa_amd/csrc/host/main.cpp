// main.cpp — the `smafa` command line over libsmafa_amd.so.  Subcommands, flag names and value types
// follow the reference's clap tree (/root/reference/src/main.rs:64-116):
//   makedb  -i/--input FILE  -d/--database FILE
//   query   -d/--database FILE  -q/--query FILE  [--max-divergence INT] [--max-num-hits INT]
//           [--limit-per-sequence INT]
//   cluster -i/--input FILE  -d/--max-divergence INT
//   count   -i/--input FILE...
// plus -v/--verbose and --quiet (logging only; results are the only thing on stdout).
// Additions of this build: --device N (query, cluster), --gpus N (query, cluster: GPUs 0..N-1, one handle and host thread
// each; the output does not depend on N), --devices a,b,.. (query, cluster: explicit list, entries may repeat),
// --alphabet nt|aa (makedb, cluster), --packed (makedb: the packed store file, host/packed.cpp).
// Exit status: 0 ok, 101 where the reference panics (a Rust panic exits 101), 1 other failures, 2 usage.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../../../include/smafa_amd.h"

static int usage(const char *msg, FILE *to = stderr) {
    if (msg) fprintf(stderr, "error: %s\n\n", msg);
    fprintf(to,
            "Usage: smafa <COMMAND>\n\n"
            "Commands:\n"
            "  makedb   Generate a searchable database\n"
            "  query    Search a database. See query --help for more information about output format.\n"
            "  cluster  Cluster sequences by similarity\n"
            "  count    Print the number of reads/bases in a possibly gzipped FASTX file\n\n"
            "makedb  -i, --input <FILE>  -d, --database <FILE>  [--alphabet nt|aa] [--packed [--device <N>]]\n"
            "        --packed: write the store as it lies in GPU memory (loads without decoding); packed on the GPU, or by\n"
            "        host threads when there is none or with --no-gpu (the same bytes)\n"
            "query   -d, --database <FILE>  -q, --query <FILE>  [--max-divergence <INT>] [--max-num-hits <INT>]\n"
            "        [--limit-per-sequence <INT>] [--device <N> | --gpus <N> | --devices <a,b,..>]\n"
            "        Output columns (tab-separated): query number (0-indexed), subject number (0-indexed),\n"
            "        divergence, subject sequence (dashes and degenerate bases shown as N)\n"
            "cluster -i, --input <FILE>  -d, --max-divergence <INT>  [--alphabet nt|aa] [--device <N> | --gpus <N> | --devices <a,b,..>]\n"
            "count   -i, --input <FILE>...\n");
    return 2;
}

static bool parse_u32(const char *s, uint32_t *out) {
    if (!s || !*s || *s == '-' || *s == '+') return false;
    char *end = nullptr;
    const unsigned long long v = strtoull(s, &end, 10);
    if (*end || v > 0xffffffffull) return false;
    *out = (uint32_t)v;
    return true;
}

static double wall_now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
    const double t_main = wall_now();
    // top-level -v/--verbose and -q/--quiet in front of the subcommand (src/main.rs:67-68); the reference takes its log level
    // from the subcommand's own flags (set_log_level(m, true), :17,34,40,47), so these are accepted and change nothing
    int first = 1;
    while (first < argc && (!strcmp(argv[first], "-v") || !strcmp(argv[first], "--verbose") || !strcmp(argv[first], "-q") ||
                            !strcmp(argv[first], "--quiet")))
        first++;
    if (first >= argc) {  // no subcommand: the help text on stdout, and success (src/main.rs:52-56)
        usage(nullptr, stdout);
        printf("\n");
        return 0;
    }
    const std::string cmd = argv[first];
    if (cmd == "-h" || cmd == "--help") {
        usage(nullptr, stdout);
        return 0;
    }
    const bool is_cluster = cmd == "cluster";
    const char *input = nullptr, *database = nullptr, *query = nullptr;
    std::vector<const char *> count_paths;
    uint32_t max_div = SMAFA_NONE, max_hits = SMAFA_NONE, limit = SMAFA_NONE, device = 0;
    bool have_max_div = false, packed = false, no_gpu = false;
    std::vector<int> devices;  // query: more than one handle
    int alphabet = SMAFA_ALPHABET_NT;
    int verbosity = 1;  // the reference logs at info level unless told otherwise (bird_tool_utils set_log_level)
    if (const char *e = getenv("SMAFA_LOG")) verbosity = atoi(e);
    for (int i = first + 1; i < argc; i++) {
        const std::string a = argv[i];
        auto value = [&]() -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
        if (a == "-i" || a == "--input") {
            if (cmd == "count") {
                while (i + 1 < argc && argv[i + 1][0] != '-') count_paths.push_back(argv[++i]);
            } else if (!(input = value())) {
                return usage("--input needs a value");
            }
        } else if (a == "--database" || (a == "-d" && !is_cluster)) {
            if (!(database = value())) return usage("--database needs a value");
        } else if (a == "--max-divergence" || (a == "-d" && is_cluster)) {
            if (!parse_u32(value(), &max_div)) return usage("--max-divergence needs an unsigned integer");
            have_max_div = true;
        } else if (a == "-q" || a == "--query") {
            if (cmd == "query") {
                if (!(query = value())) return usage("--query needs a value");
            } else {
                verbosity = 0;  // elsewhere -q is --quiet
            }
        } else if (a == "--max-num-hits") {
            if (!parse_u32(value(), &max_hits)) return usage("--max-num-hits needs an unsigned integer");
        } else if (a == "--limit-per-sequence") {
            if (!parse_u32(value(), &limit)) return usage("--limit-per-sequence needs an unsigned integer");
        } else if (a == "--device") {
            if (!parse_u32(value(), &device)) return usage("--device needs an unsigned integer");
        } else if (a == "--packed") {
            packed = true;
        } else if (a == "--no-gpu") {
            no_gpu = true;
        } else if (a == "--gpus") {
            uint32_t g = 0;
            if (!parse_u32(value(), &g) || g < 1 || g > 64) return usage("--gpus needs a count between 1 and 64");
            devices.clear();
            for (uint32_t d = 0; d < g; d++) devices.push_back((int)d);
        } else if (a == "--devices") {
            const char *v = value();
            if (!v) return usage("--devices needs a comma-separated list");
            devices.clear();
            std::string tok;
            for (const char *c = v;; c++) {
                if (*c == ',' || *c == 0) {
                    uint32_t d = 0;
                    if (!parse_u32(tok.c_str(), &d)) return usage("--devices needs a comma-separated list of GPU ordinals");
                    devices.push_back((int)d);
                    tok.clear();
                    if (*c == 0) break;
                } else {
                    tok.push_back(*c);
                }
            }
        } else if (a == "--alphabet") {
            const char *v = value();
            if (v && !strcmp(v, "nt")) alphabet = SMAFA_ALPHABET_NT;
            else if (v && !strcmp(v, "aa")) alphabet = SMAFA_ALPHABET_AA;
            else return usage("--alphabet is nt or aa");
        } else if (a == "-v" || a == "--verbose") {
            verbosity = 2;  // debug
        } else if (a == "--quiet") {
            verbosity = 0;  // errors only
        } else if (a == "-h" || a == "--help") {
            usage(nullptr, stdout);
            return 0;
        } else {
            return usage(("unexpected argument " + a).c_str());
        }
    }
    smafa_set_verbosity(verbosity);
    int rc;
    if (cmd == "makedb") {
        if (!input || !database) return usage("makedb needs --input and --database");
        rc = packed ? smafa_makedb_packed(input, database, alphabet, no_gpu ? -1 : (int)device) : smafa_makedb(input, database, alphabet);
    } else if (cmd == "query") {
        if (!database || !query) return usage("query needs --database and --query");
        if (devices.empty()) devices.push_back((int)device);
        rc = smafa_query_multi(database, query, max_div, max_hits, limit, 1, devices.data(), (int)devices.size());
    } else if (is_cluster) {
        if (!input) return usage("cluster needs --input");
        if (!have_max_div) {  // src/main.rs:43 unwraps the option
            fprintf(stderr, "called `Option::unwrap()` on a `None` value\n");
            return 101;
        }
        if (devices.empty()) devices.push_back((int)device);
        rc = smafa_cluster_multi(input, max_div, 1, devices.data(), (int)devices.size(), alphabet);
    } else if (cmd == "count") {
        if (count_paths.empty()) return usage("count needs --input");
        rc = smafa_count(count_paths.data(), count_paths.size(), 1);
    } else {
        return usage(("unrecognized subcommand " + cmd).c_str());
    }
    if (verbosity >= 2) fprintf(stderr, "[DEBUG smafa] %s: %.3f s inside main()\n", cmd.c_str(), wall_now() - t_main);
    int code = 0;
    if (rc != SMAFA_OK) {
        fprintf(stderr, "%s\n", smafa_last_error());
        code = rc == SMAFA_ERR_PANIC ? 101 : 1;
    }
    // Everything has been written (rows go out through write(2)) and every handle is closed: leave without running
    // the HIP runtime's exit handlers, which cost a noticeable part of a sub-second run.
    fflush(stdout);
    fflush(stderr);
    if (getenv("SMAFA_NO_FAST_EXIT")) return code;  // under a profiler: its tool library writes its files from an exit handler
    _exit(code);
}
