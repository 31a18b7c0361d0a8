/* vers_comm_rccl.h -- OPTIONAL companion of vers_hip.h (libvers_rccl.so, links librccl): the exchanges of the multi-GPU
 * paths over an RCCL communicator, so that a host in any language needs nothing but an ncclComm_t (or a way to pass 128
 * bytes from rank 0 to the others) to run them.  One process per GPU; RCCL moves the bytes over xGMI.
 *
 *   search   vers_rccl_gather()  -> the vers_gather_t of vers_ivf_search_sharded_dev: ONE ncclAllGather per batch, queued on
 *                                   the batch's own stream -- no host synchronisation, no second stream, no event hops
 *                                   (what `Index::search_approximate` over 8 GPUs costs beyond the local scan);
 *   build    vers_rccl_comm()    -> the vers_comm_t of vers_ivf_build_sharded_dev: five synchronous callbacks (all_gather,
 *                                   send, recv, broadcast, all_to_all_v) on the communicator's own stream.
 *
 * The reference has no counterpart (it is a single-process CPU library: ivfflat.rs:153-198 walks its lists in one thread);
 * this replaces what INTEGRATION.md section 5 otherwise asks a Rust host to write itself.
 * All functions return the int32 status codes of vers_hip.h (VERS_ERR_COMM for an RCCL failure; vers_rccl_last_error()). */
#ifndef VERS_COMM_RCCL_H
#define VERS_COMM_RCCL_H

#include "vers_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vers_rccl vers_rccl_t;

#define VERS_RCCL_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */

/* Thread-local description of the most recent failure of a vers_rccl_* call on this thread. */
const char* vers_rccl_last_error(void);
/* Rank 0 makes the communicator's id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by whatever channel the
 * host has (a file, a TCP store, MPI, torch.distributed's store). */
int32_t vers_rccl_unique_id(void* out_id128);
/* Collective over all ranks: ncclCommInitRank on `device`.  The handle owns the communicator and one stream (for the
 * synchronous callbacks of vers_rccl_comm). */
int32_t vers_rccl_create(const void* id128, uint32_t rank, uint32_t world, int32_t device, vers_rccl_t** out);
/* The same around a communicator the host already has (`nccl_comm` is an ncclComm_t); it is NOT destroyed with the handle.
 * `device` is range-checked and must be the device the communicator was made on (ncclCommCuDevice). */
int32_t vers_rccl_adopt(void* nccl_comm, int32_t device, vers_rccl_t** out);
/* ncclCommDestroy (owned communicators) after the handle's stream has drained.  A communicator marked DEAD (a bounded wait
 * expired, an exchange failed) is aborted BEFORE the stream is touched -- owned or adopted -- because draining the stream would
 * wait for an RCCL kernel that spins for a peer that is gone. */
int32_t vers_rccl_destroy(vers_rccl_t* c);
/* ncclCommAbort: what the host calls when ANY rank returned non-zero from a sharded build / search (a rank that left early has
 * left its peers inside a collective; the communicator cannot be used again).  The handle stays valid for vers_rccl_destroy;
 * every later exchange through it fails at once with VERS_ERR_COMM. */
int32_t vers_rccl_abort(vers_rccl_t* c);
/* RCCL this library was compiled against (NCCL_VERSION_CODE), the one the process actually runs (ncclGetVersion) and the path
 * of the shared object that provides it (under PyTorch: torch's bundled librccl, mapped first).  A different major version
 * is refused by vers_rccl_create / _adopt / _unique_id. */
int32_t vers_rccl_versions(int32_t* out_build_code, int32_t* out_runtime_code, char* out_path, uint64_t path_cap);
/* Fills *out for vers_ivf_search_sharded_dev / vers_ivf_search_exhaustive_sharded_dev.  `out->ctx` points into the handle:
 * keep the handle alive while searches that use it are in flight. */
int32_t vers_rccl_gather(vers_rccl_t* c, vers_gather_t* out);
/* Fills *out for vers_ivf_build_sharded_dev (each callback queues its RCCL call on the handle's stream and waits for it --
 * a BOUNDED wait: hipStreamQuery polled against VERS_RCCL_TIMEOUT_S (default 600) and the communicator's asynchronous error
 * state; a dead peer makes the callback return VERS_ERR_COMM instead of hanging the rank inside hipStreamSynchronize). */
int32_t vers_rccl_comm(vers_rccl_t* c, vers_comm_t* out);

#ifdef __cplusplus
}
#endif
#endif
