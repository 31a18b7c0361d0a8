"""TorchComm on DEVICE buffers through the RCCL backend ("nccl"), world 1 on the one GPU of the test box: the raw
pointers the library hands over are wrapped zero-copy (__cuda_array_interface__) and the collectives that exist for a
single rank (all_gather, broadcast, all_to_all_v) run in place on them.  (send / recv need a peer: covered with gloo
in test_comm_cpu.py and test_dist_build_gpu.py.)"""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_torch_comm_rccl_world1_device_buffers():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from vers_amd.dist import TorchComm
        cm = TorchComm(device=0)
        st = cm.struct
        a = torch.arange(1000, dtype=torch.float32, device="cuda"); b = torch.zeros(1000, dtype=torch.float32, device="cuda")
        assert st.all_gather(None, a.data_ptr(), b.data_ptr(), 4000) == 0
        assert torch.equal(a, b)
        assert st.broadcast(None, b.data_ptr(), 4000, 0) == 0
        c = torch.zeros(1000, dtype=torch.float32, device="cuda")
        sb = (C.c_uint64 * 1)(4000); so = (C.c_uint64 * 1)(0)
        assert st.all_to_all_v(None, a.data_ptr(), sb, so, c.data_ptr(), sb, so) == 0
        assert torch.equal(a, c)
        # the wrapped tensor aliases the buffer (no copy)
        w = cm._wrap(a.data_ptr(), 4000)
        w.view(torch.float32)[3] = -7.0
        torch.cuda.synchronize()
        assert float(a[3]) == -7.0
    finally:
        dist.destroy_process_group()
