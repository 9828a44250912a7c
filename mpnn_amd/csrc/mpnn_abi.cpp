// Error plumbing + version of the C ABI (include/mpnn_hip.h).  No exceptions cross the ABI:
// launchers return 0 or a non-zero code and leave a message here.
#include <string.h>

static thread_local char g_err[256] = "";

extern "C" void mpnn_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* mpnn_last_error(void) { return g_err; }

extern "C" int mpnn_version(void) { return 1; }
