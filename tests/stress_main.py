"""Child process of tests/test_stress_gpu.py: several host threads, each rotating its batches over its own streams, all
on ONE index, until the deadline; every result is compared with the serial answer bit for bit.
usage: python -m tests.stress_main SECONDS THREADS STREAMS_PER_THREAD"""
import sys
import threading
import time

import numpy as np
import torch

from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd.index import IVFFlatIndex


def main(seconds: float, n_threads: int, n_streams: int) -> int:
    n, d, k, b, top_k, nprobe = 30000, 96, 48, 128, 10, 8
    X = dg.dist_c(0x5731, n, d, 4 * k, dg.default_sigma(d))
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x5731, 1, k, n))
    Q = dg.dist_c(0x5732, 4 * b, d, 4 * k, dg.default_sigma(d))
    want = {(i, p): ix.search_batch(Q[i * b:(i + 1) * b], top_k, p) for i in range(4) for p in (nprobe, 0)}
    want1 = [ix.search_approximate(Q[i], top_k) for i in range(16)]
    Qd = torch.from_numpy(Q).cuda()
    torch.cuda.synchronize()
    deadline = time.time() + seconds
    errors, counts = [], [0] * (n_threads + 1)

    def dev_worker(t):
        try:
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            reps = 2 * n_streams
            ids = torch.zeros(reps, b, top_k, dtype=torch.int64, device="cuda"); dist = torch.zeros(reps, b, top_k, device="cuda")
            cnt = torch.zeros(reps, b, dtype=torch.int32, device="cuda")
            while time.time() < deadline and not errors:
                plan = []
                for r in range(reps):  # back to back, no synchronisation in between; both modes in the mix
                    i, p = (r + t) % 4, (nprobe if (r + t) % 3 else 0)
                    ix.search_dev(Qd[i * b:].data_ptr(), d, b, top_k, p, ids[r].data_ptr(), dist[r].data_ptr(), cnt[r].data_ptr(),
                                  streams[r % n_streams].cuda_stream)
                    plan.append((i, p))
                for s in streams:
                    ix.poll(s.cuda_stream)
                gi, gd, gc = ids.cpu().numpy().astype(np.uint64), dist.cpu().numpy(), cnt.cpu().numpy()
                for r, key in enumerate(plan):
                    w = want[key]
                    ok = np.array_equal(gc[r], w[2])
                    for q in range(b):
                        c = int(w[2][q])
                        ok = ok and np.array_equal(gi[r, q, :c], w[0][q, :c]) and np.array_equal(gd[r, q, :c].view(np.uint32), w[1][q, :c].view(np.uint32))
                    if not ok:
                        raise AssertionError(f"thread {t} batch {key}: result differs from the serial answer")
                counts[t] += reps
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    def host_worker():
        try:
            while time.time() < deadline and not errors:
                for i in range(16):  # Index::search_approximate, one query per call, host pointers
                    got = ix.search_approximate(Q[i], top_k)
                    if [g[0] for g in got] != [w[0] for w in want1[i]] or \
                            not np.array_equal(np.array([g[1] for g in got], dtype=np.float32).view(np.uint32),
                                               np.array([w[1] for w in want1[i]], dtype=np.float32).view(np.uint32)):
                        raise AssertionError(f"host call {i}: result differs from the serial answer")
                counts[n_threads] += 16
        except Exception as e:  # noqa: BLE001
            errors.append(("host", repr(e)))

    threads = [threading.Thread(target=dev_worker, args=(t,)) for t in range(n_threads)] + [threading.Thread(target=host_worker)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        print("STRESS FAILED", errors[:3], flush=True)
        return 1
    st = ix.prescan_stats()
    print(f"STRESS OK batches={sum(counts[:n_threads])} host_calls={counts[n_threads]} threads={n_threads} streams_per_thread={n_streams} "
          f"rescanned_queries={st['fallback_queries']} prescan_batches={st['batches']}", flush=True)
    ix.close()
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])))
