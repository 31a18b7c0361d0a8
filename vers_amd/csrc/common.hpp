// Shared declarations for libvers_hip.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/vers_hip.h"

namespace vers {

// ---- error plumbing (thread-local message behind vers_last_error) ----------
void set_error(const std::string& msg);
int32_t fail(int32_t status, const std::string& msg);

#define VERS_HIP_TRY(expr)                                                                     \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return ::vers::fail(VERS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
  } while (0)

// ---- geometry of the scan engine (scan.hip) ---------------------------------
constexpr int kWave = 64;        // CDNA4 wavefront
constexpr int kChunk = 32;       // f32 columns consumed per step
constexpr int kLoads = kChunk / 4;  // float4 loads per lane per step (1 KiB per wave each)
constexpr int kColAlign = 2 * kChunk;  // blocked matrices / padded queries: columns padded to this
constexpr int kWavesPerBlock = 4;
constexpr int kMaxTopK = 64;     // one sorted key per lane

constexpr uint64_t kKeyMax = 0xFFFFFFFFFFFFFFFFull;

// the option table (core.hip): vers_set_option / VERS_OPTIONS="name=value,..." -- the value when set, else `dflt`
int64_t opt_get(const char* name, int64_t dflt);
bool opt_set(const char* name, int64_t v);  // false: no such option
uint32_t scan_debug_flags();  // option "scan_debug" (diagnosis only; 0 in production)

inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }
inline uint64_t round_up64(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }
// a work item is addressed through one 32-bit buffer descriptor: keep it under 2 GiB
inline uint64_t max_seg_rows(uint32_t ld) { return ((1ull << 31) / ((uint64_t)ld * 4)) / 64 * 64; }

// Device-side view of the result of a scan: `k` ascending keys per slot.
// key = (order-preserving f32 bits << 32) | seq, seq = tie-break position.

}  // namespace vers
