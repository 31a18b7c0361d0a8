/* vers_hip_test.h -- TEST and one-GPU-emulation hooks of libvers_hip.so's internals.  NOT part of the drop-in boundary: these
 * entry points live in a second library, libvers_hip_test.so (vers_amd/csrc/testhooks/), which links against libvers_hip.so
 * and takes the handles that library makes.  A host that binds vers never sees this header. */
#ifndef VERS_HIP_TEST_H
#define VERS_HIP_TEST_H
#include "vers_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* TEST HOOK: overwrites every storage row that holds no vector (slack behind the lists, tile padding: uninitialised
 * device memory in production) with `value` (inf, NaN, 1e30 ...) and rebuilds the derived arrays.  Results and certificate
 * statistics must not depend on what those rows hold (tests/test_prescan_gpu.py). */
int32_t vers_ivf_test_poison_slack(vers_ivf_t* h, float value);
/* TEST HOOK: the raw pre-filter values of query q of the most recent batched nprobe search on this handle -- every (row, val)
 * the matrix-core list scan left in its partial lists (up to kp per probed list quad), the row as its vec_id, `val` exactly as
 * the certificate saw it (|x|^2 - 2 <x~, q>, or -<x~, q> for the cosine distance) and the bound the certificate charges that
 * candidate.  The test computes the reference's distance of each row and checks | val + |q|^2 - D_ref | <= bound.
 * out_info[8]: |q|^2, max |x|^2, shadow residual R^2, the bound for rows outside the list, its candidate-independent part,
 * kp, shadow in use, metric.  *out_n = values available (may exceed cap). */
int32_t vers_ivf_test_last_vals(vers_ivf_t* h, uint32_t q, uint64_t* out_vec_ids, float* out_vals, double* out_bound, uint32_t cap,
                                uint32_t* out_n, double* out_info8);
/* TEST HOOK: one wave of the matrix-core instruction a pre-filter uses, accumulated over K exactly as the kernels do, on
 * caller-chosen operands (tests/test_mfma_model_gpu.py measures the accumulation error the certificates' bounds assume).
 * kind 0 v_mfma_f32_32x32x16_f16, 1 v_mfma_f32_32x32x16_bf16 (A, B: 16-bit patterns), 2 v_mfma_f32_32x32x2_f32,
 * 3 v_mfma_f32_16x16x1_4b_f32 (f32).  A [rows][K], B [K][cols] row-major; rows x cols = 32 x 32 (kind 3: 64 x 16); host pointers. */
int32_t vers_test_mfma(int32_t device, uint32_t kind, const void* A, const void* B, uint32_t K, float* out_C);
/* MEASUREMENT HOOK: fills *out with an exchange that is a STAND-IN WITH RCCL's FOOTPRINT for one-GPU emulations of a W-GPU
 * search (scripts/emulate_shard.py): all_gather_async launches ONE kernel of `workgroups` blocks x `threads` (256 | 512) threads
 * that hold 256 VGPRs (+ 32 AGPRs at 256 threads) and lds_bytes of LDS -- the resources of RCCL's device kernel on gfx950
 * (profiles/r05_rccl_kernel_meta.txt) --, copy the rank's partial into every rank's slot and stay resident for spin_us.  Not an
 * exchange: results of a sharded search through it are this rank's partial merged with itself. */
int32_t vers_test_standin_gather(vers_gather_t* out, uint32_t rank, uint32_t world, uint32_t workgroups, uint32_t spin_us,
                                 uint32_t threads, uint32_t lds_bytes);
/* TEST HOOK: the wave-level lane networks of the kernels (scan.hip.h) on 128 host keys in[0..127], one wave on `device`:
 * out[64 j + l], j = 0..5 = lane l ^ (1 << j) of in[0..63]; out[384 + l] = lane 63 - l; out[448 ..] = in[0..63] sorted ascending
 * (bitonic network); out[512 ..] = the same by rank counting; out[576 ..] = the 64 smallest of in[0..127], ascending (two sorted
 * halves merged).  640 words out. */
int32_t vers_test_wave_net(int32_t device, const uint64_t* in, uint64_t* out);
/* TEST HOOK: the four-keys-per-lane networks of the wide candidate lists (wide.hip.h) on 512 host keys, one wave on `device`:
 * out[0..255] = in[0..255] sorted ascending, out[256..511] = in[256..511] sorted (the rolled rank sorts), out[512..767] = the 256 smallest
 * of the union, ascending; out[768..770] = elements 0, 77, 255 of that list as wide_get reads them.  771 words out. */
int32_t vers_test_wide_net(int32_t device, const uint64_t* in, uint64_t* out);
#ifdef __cplusplus
}
#endif
#endif /* VERS_HIP_TEST_H */
