"""The one exchange step of the sharded search path: a single all-gather of per-rank partial top-k
(RCCL over xGMI on the GPUs: torch.distributed backend "nccl"; gloo on CPU in the tests).

part[0] = keys, part[1] = vec ids, each [b, top_k] int64/uint64 bit patterns (kKeyMax padded).  The
gathered [world, 2, b, top_k] buffer feeds vers_topk_merge_dev with rank_stride = 2*b*top_k.
b=1024, top_k=10: 160 KiB per rank -- latency-bound on the 7 x 153 GB/s xGMI links, one collective
per batch, not a per-list or per-query exchange."""
from __future__ import annotations

import torch
import torch.distributed as dist


def all_gather_partials(part: "torch.Tensor", group=None) -> "torch.Tensor":
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(part.shape), dtype=part.dtype, device=part.device)
    try:
        dist.all_gather_into_tensor(out, part.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        chunks = [out[r] for r in range(world)]
        dist.all_gather(chunks, part.contiguous(), group=group)
    return out


def all_gather_chunks_inplace(dev_ptr: int, dtype: str, n_padded: int, chunk: int, rank: int, device: int, group=None) -> None:
    """The exchange step of the sharded build (vers_ivf_set_build_shard): the device array at `dev_ptr`
    ([n_padded] 4-byte words, this rank's results in [rank*chunk, (rank+1)*chunk)) is all-gathered in place.
    The library hands over a raw device pointer, so the data is staged through two torch tensors with
    device-to-device copies (n*4 bytes each way -- noise next to an assign pass)."""
    from .capi import check, lib
    import ctypes as C
    dev = torch.device("cuda", device)
    tdt = {"int32": torch.int32, "float32": torch.float32}[dtype]
    mine = torch.empty(chunk, dtype=tdt, device=dev)
    full = torch.empty(n_padded, dtype=tdt, device=dev)
    check(lib().vers_dev_copy(C.c_void_p(mine.data_ptr()), C.c_void_p(dev_ptr + 4 * rank * chunk), 4 * chunk))
    if dist.get_backend(group) == "nccl":  # RCCL over xGMI
        dist.all_gather_into_tensor(full, mine, group=group)
    else:  # gloo (tests on a box with fewer GPUs than ranks): staged through host memory
        world = dist.get_world_size(group)
        host = [torch.empty(chunk, dtype=tdt) for _ in range(world)]
        dist.all_gather(host, mine.cpu(), group=group)
        full.copy_(torch.cat(host).to(dev))
    torch.cuda.synchronize(dev)
    check(lib().vers_dev_copy(C.c_void_p(dev_ptr), C.c_void_p(full.data_ptr()), 4 * n_padded))
