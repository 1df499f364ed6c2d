// common.cpp — error channel of the C ABI.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <new>
#include <stdexcept>
#include <system_error>

#include "../engine.h"

namespace smafa {

static thread_local char g_error[2048];

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

int exception_code(const char *where) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return set_error(SMAFA_ERR_NOMEM, "%s: out of host memory", where);
    } catch (const std::length_error &e) {
        return set_error(SMAFA_ERR_NOMEM, "%s: out of host memory (%s)", where, e.what());
    } catch (const std::system_error &e) {
        return set_error(SMAFA_ERR_NOMEM, "%s: %s", where, e.what());
    } catch (const std::exception &e) {
        return set_error(SMAFA_ERR_INVALID, "%s: %s", where, e.what());
    } catch (...) {
        return set_error(SMAFA_ERR_INVALID, "%s: unknown failure", where);
    }
}

static int g_verbosity = 0;

int verbosity() { return g_verbosity; }

double now_seconds() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void log_line(int level, const char *fmt, ...) {
    if (g_verbosity < level) return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "[%s smafa] %s\n", level >= 2 ? "DEBUG" : "INFO", buf);
}

}  // namespace smafa

extern "C" {

void smafa_set_verbosity(int level) { smafa::g_verbosity = level; }

const char *smafa_last_error(void) { return smafa::g_error; }

void smafa_free(void *p) { free(p); }

}
