// Error string + ABI version for libbdvcil_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "bdvcil_hip.h"

static thread_local char g_err[512] = "";

void bdv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* bdv_last_error(void) { return g_err; }
extern "C" int bdv_abi_version(void) { return 29; }
extern "C" const char* bdv_source_hash(void) {
    return
#include "src_hash.inc"
        ;
}
