// Host-side plumbing of libdeepinteraction_hip.so: thread-local error string, ABI version.
#include <stdarg.h>

#include "di_common.h"

namespace di {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DI_ERR_LAUNCH;
  }
  return DI_OK;
}
}  // namespace di

extern "C" {
int di_abi_version(void) { return 1; }
const char *di_last_error(void) { return di::g_err; }

// number of nodes of a captured hipGraph_t (measurement plumbing for bench.py: "graph_nodes"); < 0 on error
long long di_graph_node_count(void *graph) {
  size_t n = 0;
  if (graph == nullptr || hipGraphGetNodes((hipGraph_t)graph, nullptr, &n) != hipSuccess) {
    di::set_error("hipGraphGetNodes failed");
    return DI_ERR_LAUNCH;
  }
  return (long long)n;
}
}
