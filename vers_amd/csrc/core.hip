// core.hip -- error plumbing and misc entry points of libvers_hip.so.
#include <cstdlib>

#include "common.hpp"

namespace vers {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int32_t fail(int32_t status, const std::string& msg) {
  g_last_error = msg;
  return status;
}
uint32_t scan_debug_flags() {
  static const uint32_t flags = [] {
    const char* e = getenv("VERS_SCAN_DEBUG");
    return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
  }();
  return flags;
}
}  // namespace vers

extern "C" {
const char* vers_last_error(void) { return vers::g_last_error.c_str(); }
int32_t vers_abi_version(void) { return 2; }  // 2: c_stride_bytes on vers_ivf_build*
int32_t vers_device_count(int32_t* out_count) {
  if (!out_count) return vers::fail(VERS_ERR_INVALID, "null out_count");
  int n = 0;
  VERS_HIP_TRY(hipGetDeviceCount(&n));
  *out_count = n;
  return VERS_OK;
}
}
