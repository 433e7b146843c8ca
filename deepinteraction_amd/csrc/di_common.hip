// Host-side plumbing of libdeepinteraction_hip.so: thread-local error string, ABI version.
#include <stdarg.h>

#include <stdlib.h>

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

static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static thread_local int g_ev_consumed = 0;

bool take_launch_events(hipEvent_t &start, hipEvent_t &stop) {
  if (g_ev_start == nullptr) return false;
  start = g_ev_start;
  stop = g_ev_stop;
  g_ev_start = g_ev_stop = nullptr;
  g_ev_consumed = 1;
  return true;
}

}  // namespace di

extern "C" {
int di_timed_begin(void **start_ev, void **stop_ev) {
  hipEvent_t a = nullptr, b = nullptr;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
    di::set_error("hipEventCreate failed");
    return DI_ERR_LAUNCH;
  }
  di::g_ev_start = a;
  di::g_ev_stop = b;
  di::g_ev_consumed = 0;
  *start_ev = a;
  *stop_ev = b;
  return DI_OK;
}
// 1 when the launch issued since di_timed_begin took the events (else they were never recorded), and disarms the thread
int di_timed_consumed(void) {
  di::g_ev_start = di::g_ev_stop = nullptr;
  return di::g_ev_consumed;
}
int di_timed_elapsed_us(void *start_ev, void *stop_ev, int recorded, float *us) {
  int rc = DI_OK;
  if (recorded) {
    float ms = 0.f;
    if (hipEventSynchronize((hipEvent_t)stop_ev) != hipSuccess ||
        hipEventElapsedTime(&ms, (hipEvent_t)start_ev, (hipEvent_t)stop_ev) != hipSuccess) {
      di::set_error("hipEventElapsedTime failed");
      rc = DI_ERR_LAUNCH;
    }
    *us = ms * 1e3f;
  }
  (void)hipEventDestroy((hipEvent_t)start_ev);
  (void)hipEventDestroy((hipEvent_t)stop_ev);
  return rc;
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
    // DI_CUS (measurement): the CU count the persistent kernels size their grids by - below the device's count it leaves CUs to
    // the narrow launches of other streams
    static const int env = getenv("DI_CUS") ? atoi(getenv("DI_CUS")) : 0;
    cus[dev] = (env > 0 && env < n) ? env : n;
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
