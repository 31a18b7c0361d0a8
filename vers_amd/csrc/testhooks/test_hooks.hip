// test_hooks.hip -- libvers_hip_test.so (include/vers_hip_test.h): the TEST and one-GPU-emulation hooks, kept OUT of the product
// library.  Linked against libvers_hip.so: the hooks reach into a handle through the library's internal headers and call its internal
// functions (default symbol visibility), so a handle made by libvers_hip.so is what they take.
#include "../ivf_handle.hpp"
#include "../prescan.hip.h"
#include "../wide.hip.h"
#include "../../../include/vers_hip_test.h"


// =================================================================================================
// MEASUREMENT HOOK: a vers_gather_t whose exchange is a STAND-IN WITH RCCL's FOOTPRINT (no 8-GPU node has been available in
// any round; with one rank ncclAllGather degenerates to a copy kernel, which says nothing about what the real device kernel
// needs).  RCCL's all-gather on gfx950 is ONE kernel of `channels` workgroups (profiles/r05_rccl_kernel_meta.txt, read from the
// code objects with llvm-readelf --notes):
//   /opt/rocm 7.2 librccl (2.27.7)  ncclDevKernel_Generic_*   248-256 VGPRs, 37,664 B LDS, 512 threads
//   torch's bundled librccl (2.26.6) rcclGenericKernel<1|2|4>  244-248 VGPRs + 17-32 AGPRs, 19,744 B LDS, 256 threads, 352 B scratch
// i.e. a workgroup that needs (nearly) a whole CU's registers and cannot sit beside a list-scan block (448 of a SIMD's 512).
// The stand-in: `workgroups` blocks of `threads` threads holding 256 VGPRs (+ 32 AGPRs for the 256-thread shape) and `lds`
// bytes, which copy this rank's partial into every rank's slot of the gathered buffer (the bytes an all-gather writes: the
// merge then reads `world` well-formed partials) and stay resident for `spin_us` (the time the peers' bytes would take over xGMI).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void rccl_standin_kernel(const uint64_t* send, uint64_t* recv, uint64_t words, uint32_t world, uint32_t spin_ticks) {
  extern __shared__ __attribute__((aligned(16))) uint64_t sl[];
  const unsigned long long t0 = wall_clock64();  // 100 MHz
  asm volatile("v_mov_b32 v255, 0" ::: "v255");           // the register footprint of RCCL's generic kernel
  if constexpr (THREADS <= 256) asm volatile("v_accvgpr_write_b32 a31, 0" ::: "a31");
  sl[threadIdx.x] = t0;
  for (uint32_t r = 0; r < world; ++r)
    for (uint64_t i = (uint64_t)blockIdx.x * THREADS + threadIdx.x; i < words; i += (uint64_t)gridDim.x * THREADS) recv[(uint64_t)r * words + i] = send[i];
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (sl[threadIdx.x] == 1ull) recv[0] = 0;  // (keeps the LDS allocation alive; never true)
}
struct StandinGather {
  uint32_t world, workgroups, spin_us, threads, lds;
};
static int32_t standin_all_gather(void* ctx, const void* send_dev, void* recv_dev, uint64_t bytes, void* stream) {
  const StandinGather* sg = (const StandinGather*)ctx;
  const uint64_t words = bytes / 8;
  if (sg->threads > 256) {
    hipLaunchKernelGGL(rccl_standin_kernel<512>, dim3(sg->workgroups), dim3(512), sg->lds, (hipStream_t)stream, (const uint64_t*)send_dev, (uint64_t*)recv_dev, words,
                       sg->world, sg->spin_us * 100u);
  } else {
    hipLaunchKernelGGL(rccl_standin_kernel<256>, dim3(sg->workgroups), dim3(256), sg->lds, (hipStream_t)stream, (const uint64_t*)send_dev, (uint64_t*)recv_dev, words,
                       sg->world, sg->spin_us * 100u);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" int32_t vers_test_standin_gather(vers_gather_t* out, uint32_t rank, uint32_t world, uint32_t workgroups, uint32_t spin_us, uint32_t threads, uint32_t lds_bytes) {
  if (!out || world == 0 || rank >= world || workgroups == 0 || (threads != 256 && threads != 512) || lds_bytes < threads * 8u || lds_bytes > 64u * 1024u)
    return fail(VERS_ERR_INVALID, "vers_test_standin_gather: bad arguments (threads 256 | 512, lds_bytes in [8 * threads, 64 KiB])");
  StandinGather* sg = new StandinGather{world, workgroups, spin_us, threads, lds_bytes};  // (lives as long as the process: a measurement hook)
  out->ctx = sg;
  out->rank = rank;
  out->world = world;
  out->all_gather_async = standin_all_gather;
  return VERS_OK;
}

namespace vers {
__global__ __launch_bounds__(kWave) void wave_net_test_kernel(const uint64_t* in, uint64_t* out) {
  const int lane = threadIdx.x;
  const uint64_t a = in[lane], b = in[kWave + lane];
  out[0 * kWave + lane] = lane_xor64<1>(a, lane);
  out[1 * kWave + lane] = lane_xor64<2>(a, lane);
  out[2 * kWave + lane] = lane_xor64<4>(a, lane);
  out[3 * kWave + lane] = lane_xor64<8>(a, lane);
  out[4 * kWave + lane] = lane_xor64<16>(a, lane);
  out[5 * kWave + lane] = lane_xor64<32>(a, lane);
  out[6 * kWave + lane] = lane_rev64(a, lane);
  uint64_t s0 = a, s1 = a, s2 = b;
  wave_bitonic_sort64(s0, lane);
  wave_rank_sort64(s1, lane);
  wave_bitonic_sort64(s2, lane);
  out[7 * kWave + lane] = s0;
  out[8 * kWave + lane] = s1;
  wave_merge_sorted64(s0, s2, lane);
  out[9 * kWave + lane] = s0;
}
}  // namespace vers

namespace vers {
__global__ __launch_bounds__(kWave) void wide_net_test_kernel(const uint64_t* in, uint64_t* out) {
  const int lane = threadIdx.x;
  uint64_t a[kWideR], b[kWideR];
#pragma unroll
  for (int r = 0; r < kWideR; ++r) { a[r] = in[r * kWave + lane]; b[r] = in[kWideKeys + r * kWave + lane]; }
  wide_sort(a, lane);
  wide_sort<true>(b, lane);
#pragma unroll
  for (int r = 0; r < kWideR; ++r) { out[r * kWave + lane] = a[r]; out[kWideKeys + r * kWave + lane] = b[r]; }
  wide_merge_sorted(a, b, lane);
#pragma unroll
  for (int r = 0; r < kWideR; ++r) out[2 * kWideKeys + r * kWave + lane] = a[r];
  const uint64_t e0 = wide_get(a, 0), e77 = wide_get(a, 77), e255 = wide_get(a, 255);  // (whole wave)
  if (lane == 0) { out[3 * kWideKeys] = e0; out[3 * kWideKeys + 1] = e77; out[3 * kWideKeys + 2] = e255; }
}
}  // namespace vers

extern "C" int32_t vers_test_wide_net(int32_t device, const uint64_t* in, uint64_t* out) {
  if (!in || !out) return fail(VERS_ERR_INVALID, "vers_test_wide_net: null argument");
  DeviceGuard g(device);
  const size_t n_in = 2 * kWideKeys, n_out = 3 * kWideKeys + 3;
  uint64_t *d_in = nullptr, *d_out = nullptr;
  VERS_HIP_TRY(hipMalloc(&d_in, n_in * sizeof(uint64_t)));
  if (hipMalloc(&d_out, n_out * sizeof(uint64_t)) != hipSuccess) { (void)hipFree(d_in); return fail(VERS_ERR_HIP, "vers_test_wide_net: out of device memory"); }
  hipError_t e = hipMemcpy(d_in, in, n_in * sizeof(uint64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(wide_net_test_kernel, dim3(1), dim3(kWave), 0, 0, d_in, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, n_out * sizeof(uint64_t), hipMemcpyDeviceToHost);
  (void)hipFree(d_in); (void)hipFree(d_out);
  if (e != hipSuccess) return fail(VERS_ERR_HIP, std::string("vers_test_wide_net: ") + hipGetErrorString(e));
  return VERS_OK;
}

extern "C" int32_t vers_test_wave_net(int32_t device, const uint64_t* in, uint64_t* out) {
  if (!in || !out) return fail(VERS_ERR_INVALID, "vers_test_wave_net: null argument");
  DeviceGuard g(device);
  uint64_t *d_in = nullptr, *d_out = nullptr;
  VERS_HIP_TRY(hipMalloc(&d_in, 128 * sizeof(uint64_t)));
  if (hipMalloc(&d_out, 640 * sizeof(uint64_t)) != hipSuccess) { (void)hipFree(d_in); return fail(VERS_ERR_HIP, "vers_test_wave_net: out of device memory"); }
  hipError_t e = hipMemcpy(d_in, in, 128 * sizeof(uint64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(wave_net_test_kernel, dim3(1), dim3(kWave), 0, 0, d_in, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, 640 * sizeof(uint64_t), hipMemcpyDeviceToHost);
  (void)hipFree(d_in); (void)hipFree(d_out);
  if (e != hipSuccess) return fail(VERS_ERR_HIP, std::string("vers_test_wave_net: ") + hipGetErrorString(e));
  return VERS_OK;
}


extern "C" {

int32_t vers_ivf_test_poison_slack(vers_ivf_t* h, float value) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipDeviceSynchronize());
  if (h->cap_rows == 0) return VERS_OK;
  if (int32_t rc = poison_slack(h, value, nullptr)) return rc;
  if (int32_t rc = refresh_norms(h, 0, h->cap_rows, nullptr)) return rc;
  VERS_HIP_TRY(hipDeviceSynchronize());
  return VERS_OK;
}

int32_t vers_ivf_test_last_vals(vers_ivf_t* h, uint32_t q, uint64_t* out_vec_ids, float* out_vals, double* out_bound, uint32_t cap, uint32_t* out_n,
                                double* out_info) {
  if (!h || !out_n || (cap && (!out_vec_ids || !out_vals || !out_bound))) return fail(VERS_ERR_INVALID, "bad arguments");
  std::shared_lock<std::shared_mutex> lk(h->index);
  UseLastWs use_ws(h);
  if (!use_ws.ok || !W->last_pre.valid) return fail(VERS_ERR_INVALID, "vers_ivf_test_last_vals: the most recent search did not run the matrix-core list scan");
  const auto lp = W->last_pre;
  if (q >= lp.b) return fail(VERS_ERR_INVALID, "vers_ivf_test_last_vals: no such query in the last batch");
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipDeviceSynchronize());
  const uint32_t P = lp.P, S = lp.S_max, kp = lp.kp;
  const uint64_t n_pj = (uint64_t)lp.b * P;
  std::vector<uint64_t> keys((size_t)P * S * kp);
  std::vector<uint32_t> pl(P), pp(P), pn(P);
  const uint32_t* pj = W->pj.as<uint32_t>();
  VERS_HIP_TRY(hipMemcpy(keys.data(), W->partials.as<uint64_t>() + (uint64_t)q * P * S * kp, keys.size() * 8, hipMemcpyDeviceToHost));
  VERS_HIP_TRY(hipMemcpy(pl.data(), pj + (uint64_t)q * P, P * 4, hipMemcpyDeviceToHost));
  VERS_HIP_TRY(hipMemcpy(pp.data(), pj + n_pj + (uint64_t)q * P, P * 4, hipMemcpyDeviceToHost));
  VERS_HIP_TRY(hipMemcpy(pn.data(), pj + 3 * n_pj + lp.b + (uint64_t)q * P, P * 4, hipMemcpyDeviceToHost));
  std::vector<float> qrow(h->ldq);
  VERS_HIP_TRY(hipMemcpy(qrow.data(), lp.qp + (uint64_t)q * h->ldq, (size_t)h->ldq * 4, hipMemcpyDeviceToHost));
  uint32_t misc[4] = {0, 0, 0, 0};
  VERS_HIP_TRY(hipMemcpy(misc, h->pre_misc.p, 16, hipMemcpyDeviceToHost));
  double qn = 0.0;
  for (uint32_t j = 0; j < h->ldq; ++j) qn += (double)qrow[j] * (double)qrow[j];
  float xmax2, r2;
  memcpy(&xmax2, &misc[0], 4); memcpy(&r2, &misc[2], 4);
  double rq2 = 0.0;  // hi-only query blocks: the query's squared fp16 residual as the exact finish sums it (pre_bound)
  if (lp.shadow == 2) {
    const float qscale = h->metric ? -1.0f : -2.0f;
    for (uint32_t j = 0; j < h->ldq; ++j) {
      const float y = qscale * qrow[j], dl = y - (float)(_Float16)y;
      rq2 += (double)dl * (double)dl;
    }
  }
  const PreBound pb = pre_bound(qn, (double)xmax2, lp.shadow ? (double)r2 : 0.0, h->ld, h->metric, lp.shadow, rq2);
  if (out_info) { out_info[0] = qn; out_info[1] = xmax2; out_info[2] = lp.shadow ? r2 : 0.0; out_info[3] = pb.global; out_info[4] = pb.common; out_info[5] = kp; out_info[6] = lp.shadow; out_info[7] = h->metric; }
  uint32_t n = 0;
  for (uint32_t j = 0; j < P; ++j) {
    if (pl[j] == kNoList) continue;
    uint32_t off_j = 0;
    VERS_HIP_TRY(hipMemcpy(&off_j, h->slot_off.as<uint32_t>() + pl[j], 4, hipMemcpyDeviceToHost));
    for (uint32_t sq = 0; sq < pn[j] && sq < S; ++sq)
      for (uint32_t i = 0; i < kp; ++i) {
        const uint64_t key = keys[((size_t)j * S + sq) * kp + i];
        if (key == kKeyMax) continue;
        if (n < cap) {
          const uint32_t row = off_j + ((uint32_t)key - pp[j]);
          uint32_t vid = 0;
          VERS_HIP_TRY(hipMemcpy(&vid, h->row_ids.as<uint32_t>() + row, 4, hipMemcpyDeviceToHost));
          const uint32_t vb = order_bits_to_f32_bits((uint32_t)(key >> 32));
          float v; memcpy(&v, &vb, 4);
          out_vec_ids[n] = vid; out_vals[n] = v; out_bound[n] = pb.of((double)v);
        }
        ++n;
      }
  }
  *out_n = n;
  return VERS_OK;
}


}  // extern "C"
