// The library's last-error string (include/concepthash_hip.h: ch_last_error).  Thread-local: a status and its message belong to the
// thread that made the call.  Plain C++ so that host-only builds (the sanitizer build of jpeg_host.cpp) link it too.
#include "ch_host.h"

static thread_local std::string g_last_error;

void ch_set_error(const std::string &msg) { g_last_error = msg; }
extern "C" const char *ch_last_error(void) { return g_last_error.c_str(); }
