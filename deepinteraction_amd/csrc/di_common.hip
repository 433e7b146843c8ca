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

namespace di {
static const unsigned long long *g_i2p_seed_ptr = nullptr;
const unsigned long long *i2p_seed_ptr() { return g_i2p_seed_ptr; }
}  // namespace di

extern "C" {
int di_i2p_set_seed_ptr(const void *dev_ptr) {
  di::g_i2p_seed_ptr = reinterpret_cast<const unsigned long long *>(dev_ptr);
  return 0;
}
int di_abi_version(void) { return 2; }   // 2: di_tok_heads gained qpos2 / pos2_out
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

namespace di {

int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    set_error("cannot query the current device");
    return 0;
  }
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      set_error("cannot query the CU count of device %d", dev);
      return 0;
    }
    cus[dev] = n;
  }
  return cus[dev];
}

int ensure_lds(LdsRaised &state, const void *kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    set_error("cannot query the current device");
    return DI_ERR_LAUNCH;
  }
  if (state.done & (1ull << dev)) return DI_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(%d bytes of LDS): %s", bytes, hipGetErrorString(e));
    return DI_ERR_LAUNCH;
  }
  state.done |= 1ull << dev;
  return DI_OK;
}

}  // namespace di
