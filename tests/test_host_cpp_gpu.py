"""The C++ host mirror (vers_amd/host/ivfflat.hpp) against the C ABI from COMPILED code: g++ builds
tests/cpp/host_demo.cpp, which runs build_index -> add -> save_index -> load_index -> search_approximate on
256-byte-aligned Vector<N> rows and compares with oracle results bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import build as vbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_demo(tmp_path):
    exe = os.path.join(tmp_path, "host_demo")
    lib = vbuild.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_demo.cpp"),
                           "-L" + os.path.dirname(lib), "-lvers_hip", "-Wl,-rpath," + os.path.dirname(lib)])
    return exe


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    assert os.path.exists(compile_demo(str(tmp_path)))


@pytest.mark.gpu
@pytest.mark.parametrize("d", [40, 300])  # 300: Vector<300> has pitch 1280 B for 1200 B of data (80 B of padding per row)
def test_cpp_host_demo_matches_oracle(tmp_path, d):
    exe = compile_demo(str(tmp_path))
    n, k, iters, top_k, n_q = 900, 9, 4, 10, 5
    X = dg.dist_c(0x71, n, d, 12, dg.default_sigma(d))
    init = mg.init_draws(2, 1, k, n)
    b = co.build_index(X, k, 1, iters, init)
    extra = dg.dist_u(0x72, 1, d)[0]
    c = co.add_cluster(b["centroids"], extra)
    ids = [list(l) for l in b["ids"]]; ids[c].append(n)
    values = np.concatenate([X, extra[None]], axis=0)
    Q = dg.dist_c(0x73, n_q, d, 12, dg.default_sigma(d)); Q[1] = extra
    blob = struct.pack("<5Q", n, k, iters, top_k, n_q) + X.tobytes() + init.astype("<u8").tobytes() + extra.tobytes() + Q.tobytes()
    blob += b["assignments"].astype("<u8").tobytes()
    blob += np.ascontiguousarray(b["centroids"], dtype="<f4").tobytes()
    for q in Q:
        oi, od = co.search_approximate(values, b["centroids"], ids, q, top_k)
        blob += struct.pack("<Q", len(oi))
        for i, dd in zip(oi, od):
            blob += struct.pack("<QI4x", int(i), int(np.float32(dd).view(np.uint32)))
    ei, _ = co.search_exhaustive(values, Q[0], 5)
    blob += ei.astype("<u8").tobytes()
    fx = os.path.join(tmp_path, "fixture.bin")
    open(fx, "wb").write(blob)
    r = subprocess.run([exe, fx, os.path.join(tmp_path, "ivfflat.index"), str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches=0" in r.stdout, r.stdout + r.stderr
