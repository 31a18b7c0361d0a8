"""One process per GPU, started from a plain `python bench.py --gpus N` command line.

The parent NEVER touches the GPU (no HIP call, no torch.cuda call): it only starts N children of the same script with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- the environment `torch.distributed.run` would give
them -- waits, relays rank 0's stdout (the JSON line) and returns non-zero when any child failed.  Nothing is ever
exec'd from a process that has initialised the GPU.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher() -> bool:
    """True inside a rank started by torch.distributed.run or by spawn_ranks."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_ranks(script: str, argv: list, n: int, extra_env: dict | None = None, poll_s: float = 0.2) -> int:
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if extra_env:
            env.update(extra_env)
        # rank 0 owns stdout (the one JSON line); the other ranks' stdout goes to stderr so nothing is printed twice
        out = subprocess.PIPE if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=out, text=(r == 0)))

    def relay():  # rank 0's JSON lines to our stdout; anything else it prints there (library banners) to stderr
        for line in procs[0].stdout:
            dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            dst.write(line)
            dst.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[launch] rank {r} exited with status {code}: stopping the other ranks", file=sys.stderr, flush=True)
                for o in live:  # exact PIDs of our own children only
                    procs[o].terminate()
        if live:
            time.sleep(poll_s)
    th.join(timeout=10)
    if rc:
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc
