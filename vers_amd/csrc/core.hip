// core.hip -- error plumbing and misc entry points of libvers_hip.so.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.hpp"

namespace vers {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int32_t fail(int32_t status, const std::string& msg) {
  g_last_error = msg;
  return status;
}
// ---- the option table -----------------------------------------------------------------------------------------------------------
// Every switch of the library is a named 64-bit option: set by vers_set_option(name, value) (include/vers_hip.h lists the names), read
// where it is used.  The ENVIRONMENT is read for exactly three names, once, when the first option is read:
//   VERS_OPTIONS="name=value,name=value"   the same names and values as vers_set_option (A/B runs and tests of a process one does not
//                                          control from inside: value in any base strtoll(.., 0) takes)
//   VERS_SHADOW=0|1, VERS_ROWMAJOR=0|1     the two memory switches INTEGRATION.md documents (= options "shadow", "rowmajor")
// (the optional RCCL adapter, a separate library, reads VERS_RCCL_TIMEOUT_S).
namespace {
struct OptEntry { const char* name; std::atomic<int64_t> v{0}; std::atomic<bool> set{false}; };
OptEntry g_opts[] = {
    {"shadow"}, {"rowmajor"}, {"single_shadow"}, {"scan_events"}, {"scan_reserve_cus"}, {"pre_min_batch"}, {"host_spin"}, {"gemm_x3"},
    {"prescan"}, {"pre_slack"}, {"seg_rows"}, {"pre_narrow"}, {"pre_wide"}, {"pre_hi_only"}, {"coarse"}, {"coarse1"}, {"scan1t"},
    {"ref_as_nprobe1"}, {"assign"}, {"assign_tiles"}, {"assign_tiles_min"}, {"upload_stage_mb"}, {"scan_debug"},
    {"poison_alloc"}, {"poison_slack_bits"}, {"test_fail_sharded"}, {"memory"}, {"wide_k"}, {"assign_terms"}, {"assign_glds"},
};
OptEntry* opt_find(const char* name) {
  for (OptEntry& e : g_opts)
    if (std::strcmp(e.name, name) == 0) return &e;
  return nullptr;
}
void opt_env_once() {
  static std::once_flag once;
  std::call_once(once, [] {
    auto put = [](const std::string& n, const char* val) {
      if (OptEntry* e = opt_find(n.c_str())) { e->v.store(strtoll(val, nullptr, 0)); e->set.store(true); }
      else fprintf(stderr, "[vers] VERS_OPTIONS: unknown option '%s' ignored\n", n.c_str());
    };
    if (const char* s = getenv("VERS_SHADOW")) put("shadow", s);
    if (const char* s = getenv("VERS_ROWMAJOR")) put("rowmajor", s);
    if (const char* s = getenv("VERS_OPTIONS")) {
      std::string all(s);
      size_t i = 0;
      while (i < all.size()) {
        size_t j = all.find(',', i);
        if (j == std::string::npos) j = all.size();
        const std::string kv = all.substr(i, j - i);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos && eq > 0) put(kv.substr(0, eq), kv.c_str() + eq + 1);
        i = j + 1;
      }
    }
  });
}
}  // namespace
int64_t opt_get(const char* name, int64_t dflt) {
  opt_env_once();
  const OptEntry* e = opt_find(name);
  return e && e->set.load(std::memory_order_relaxed) ? e->v.load(std::memory_order_relaxed) : dflt;
}
bool opt_set(const char* name, int64_t v) {
  opt_env_once();
  OptEntry* e = opt_find(name);
  if (!e) return false;
  e->v.store(v); e->set.store(true);
  return true;
}
uint32_t scan_debug_flags() { return (uint32_t)opt_get("scan_debug", 0); }
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
