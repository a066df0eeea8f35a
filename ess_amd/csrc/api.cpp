// Error reporting and version for libess_hip.so.  No mutable global state besides the thread-local last-error buffer and the
// memo of dynamic-LDS opt-ins already made with the runtime (below).
#include "common.h"

static thread_local char g_err[512] = "";

void ess_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ess_last_error(void) { return g_err; }
extern "C" int ess_version(void) { return 110; }  // 110 (round 6): workspace_bytes on the loss entry points, ESS_COMPUTE_F16, the mixed-configuration bridges

// ---- dynamic-LDS opt-in, once per (kernel, device): see ess_allow_lds in common.h
#include <map>
#include <mutex>
#include <utility>
void ess_allow_lds_impl(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  size_t& g = granted[{kernel, dev}];
  if (bytes <= g) return;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) g = bytes;
}

// ---- tuning switches (process-wide; every setting gives identical results -- they choose between kernels of equal arithmetic)
#include <string.h>
namespace essconv { void set_wide_mode(int m); int wide_mode(); int device_cus(); }
void set_in_small_threads(int v);
int in_small_threads();
extern "C" int ess_tuning_set(const char* key, int32_t value) {
  if (key && !strcmp(key, "conv_wide")) { essconv::set_wide_mode(value); return ESS_OK; }
  if (key && !strcmp(key, "in_small_threads")) { set_in_small_threads(value); return ESS_OK; }
  ess_set_error("tuning_set: unknown key '%s'", key ? key : "(null)");
  return ESS_EINVAL;
}
extern "C" int ess_tuning_get(const char* key, int32_t* value) {
  if (key && value && !strcmp(key, "conv_wide")) { *value = essconv::wide_mode(); return ESS_OK; }
  if (key && value && !strcmp(key, "in_small_threads")) { *value = in_small_threads(); return ESS_OK; }
  if (key && value && !strcmp(key, "device_cus")) { *value = essconv::device_cus(); return ESS_OK; }  // (read-only: the dispatcher's CU count)
  ess_set_error("tuning_get: unknown key '%s'", key ? key : "(null)");
  return ESS_EINVAL;
}
