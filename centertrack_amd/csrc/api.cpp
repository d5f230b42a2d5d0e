// Error reporting and version of libcentertrack_hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "centertrack_hip.h"

static thread_local char g_err[512] = "";

void ct_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ct_last_error(void) { return g_err; }
extern "C" int ct_version(void) { return 100; }
