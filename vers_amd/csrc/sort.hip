// sort.hip -- stable grouping of row indices by cluster id (plumbing, not the hot path):
// ids sorted by (cluster, ascending row index) == the reference's inverted lists
// `ids[cluster].push(vec_id)` in ascending vec_id (ivfflat.rs:123-127).  Uses rocPRIM's
// LSD radix sort, which is stable.
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "kmeans.hpp"

namespace vers {

__global__ void iota_kernel(uint32_t* v, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

size_t group_by_cluster_temp_bytes(uint32_t n, uint32_t k) {
  size_t bytes = 0;
  unsigned bits = 1;
  while ((1ull << bits) < k) ++bits;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                  (uint32_t*)nullptr, (size_t)n, 0u, bits, (hipStream_t) nullptr);
  return bytes + 3ull * n * sizeof(uint32_t) + 256;
}

// assign[n] -> sorted_ids[n]; temp must hold group_by_cluster_temp_bytes(n, k).
int32_t group_by_cluster(const uint32_t* assign, uint32_t n, uint32_t k, uint32_t* sorted_ids, void* temp,
                         size_t temp_bytes, hipStream_t st) {
  if (n == 0) return VERS_OK;
  unsigned bits = 1;
  while ((1ull << bits) < k) ++bits;
  uint32_t* iota = (uint32_t*)temp;
  uint32_t* keys_out = iota + n;
  char* rp_tmp = (char*)(keys_out + n);
  rp_tmp = (char*)(((uintptr_t)rp_tmp + 255) & ~(uintptr_t)255);
  size_t rp_bytes = temp_bytes - (size_t)(rp_tmp - (char*)temp);
  hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, iota, n);
  VERS_HIP_TRY(hipGetLastError());
  VERS_HIP_TRY(rocprim::radix_sort_pairs((void*)rp_tmp, rp_bytes, assign, keys_out, (const uint32_t*)iota, sorted_ids,
                                         (size_t)n, 0u, bits, st));
  return VERS_OK;
}

}  // namespace vers
