// thread-local error text + version for libmdil_hip.so
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mdil_hip.h"

static thread_local char g_err[512] = "";

void mdil_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mdil_last_error(void) { return g_err; }
extern "C" int mdil_version(void) { return 110; }   // 110: tickets (finalize inside the producing launch), mdil_tapconv_bn_train, mdil_bn_backward_apply
