// common.cpp — error channel of the C ABI.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

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

}  // namespace smafa

extern "C" {

const char *smafa_last_error(void) { return smafa::g_error; }

void smafa_free(void *p) { free(p); }

}
