"""Can RCCL run two ranks on ONE GPU here?  (It would let the nccl branches of vers_amd/dist.py and bench.py execute with
world > 1 on the single-GPU boxes a builder gets.)  usage: python scripts/probe/rccl_two_ranks_one_gpu.py"""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        x = torch.full((4,), float(rank), device="cuda")
        out = torch.empty(world * 4, device="cuda")
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_gather over RCCL with both ranks on cuda:0 -> {out.tolist()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:300]}", flush=True)
        sys.exit(3)


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    try:
        mp.spawn(worker, args=(2, port), nprocs=2, join=True)
    except Exception as e:  # noqa: BLE001
        print("spawn ended with:", str(e)[:200])
