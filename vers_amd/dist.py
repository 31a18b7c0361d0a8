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
