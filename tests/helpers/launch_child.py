"""Child of tests/test_launch.py: what one rank of `bench.py --gpus N` does around its work -- joins the process group
named by the launcher's environment, and rank 0 prints ONE JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(7)
dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": float(t.item()), "local_rank": int(os.environ["LOCAL_RANK"])}), flush=True)
else:
    print("noise from a non-zero rank")  # must not reach the parent's stdout
dist.destroy_process_group()
