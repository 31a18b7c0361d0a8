// flat_shadow.hpp -- the flat index's single query on an fp16 shadow of its rows (round 5): what flat.hip owns and calls; the
// kernels (flat1h_kernel, and the inverted-list path's exact finish: ivf_rescore_kernel<16>, fallback_kernel) live in ivf_search.hip.
#pragma once
#include "common.hpp"

namespace vers {

struct FlatShadow {
  uint16_t* rows_h = nullptr;   // fp16 shadow of the blocked rows (prescan.hip.h: rows_to_f16_kernel's layout)
  float* xnorm = nullptr;       // |x|^2 per storage row
  uint32_t* row_ids = nullptr;  // row -> vec id (the row itself; 0xFFFFFFFF for the last tile's padding rows)
  uint32_t* misc = nullptr;     // [0] max |x|^2 (bits) [1] failed certificates [2] max |x - fp16(x)|^2 (bits) | [8 ..] the finish's tables
  uint64_t* slots = nullptr;    // partial slots [n_slots][64]
  uint64_t* fb_part = nullptr;  // the exact re-scan's partial lists
  uint32_t* fb_ctr = nullptr;
  uint64_t rows_built = 0;      // rows the shadow was derived from (0: none)
  uint32_t n_slots = 0;
  size_t bytes = 0;
  void release();
};

// after the rows changed: (re)build the shadow.  Optional memory: a failed allocation leaves the f32 scan in charge (VERS_OK).
int32_t flat_shadow_derive(FlatShadow& s, const float* rows_blocked, uint64_t n, uint32_t ld, int n_cu);
// may a single query with these parameters take the shadow path?  (vers_set_option "shadow" / "single_shadow", the shadow's measured
// residual finite, k + slack <= 64 keys, the shadow addressable by one buffer descriptor)
bool flat_shadow_usable(const FlatShadow& s, uint64_t n, uint32_t ld, uint32_t top_k);
// one padded query (ld floats, device) -> top_k (id, distance) pairs in the reference's order, exactly (utils.rs:68-82); queued on st
int32_t flat_shadow_search1(FlatShadow& s, const float* rows_blocked, uint64_t n, uint32_t ld, int n_cu, const float* q_padded, uint32_t top_k,
                            uint32_t metric, uint32_t* status, uint64_t* out_ids, float* out_dist, uint32_t* out_count, hipStream_t st,
                            hipEvent_t ev0, hipEvent_t ev1);

}  // namespace vers
