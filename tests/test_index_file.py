"""The index file of Index::save_index / load_index (base.rs:31-58; field order ivfflat.rs:9-15): a 3-vector,
2-cluster IVFFlatIndex<2> whose bytes are written out BY HAND below from the bincode 1.3 rules (little endian, u64
lengths, fields in order without tags, [f32; N] raw without a length, no alignment padding), checked against the
Python mirror's writer / reader and against the compiled C++ mirror in both directions.  No GPU involved.
(bincode / serde_arrays are third-party crates that are not vendored in the reference: this pins OUR statement of
their published format -- 'parity unpinned' against a real vers-written file, see DESIGN.md.)"""
import os
import struct
import subprocess

import numpy as np
import pytest

from vers_amd import build as vbuild
from vers_amd.index import read_index_file, write_index_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VALUES = np.array([[1.0, -2.0], [0.5, 0.25], [-0.0, 3.0]], dtype=np.float32)
CENTROIDS = np.array([[0.75, -0.875], [0.0, 3.0]], dtype=np.float32)
ASSIGN = np.array([0, 0, 1], dtype=np.uint64)
IDS = [[0, 1], [2]]


def f32(x):  # the IEEE-754 single bit pattern, little endian
    return struct.pack("<f", x)


def u64(x):
    return struct.pack("<Q", x)


EXPECTED = b"".join([
    u64(2),                                           # num_centroids: usize
    u64(3),                                           # values: Vec<Vector<2>> -- length
    bytes.fromhex("0000803f"), bytes.fromhex("000000c0"),   #   1.0, -2.0   (raw [f32; 2], no length, no padding to 256 B)
    bytes.fromhex("0000003f"), bytes.fromhex("0000803e"),   #   0.5, 0.25
    bytes.fromhex("00000080"), bytes.fromhex("00004040"),   #   -0.0 (sign bit kept), 3.0
    u64(2),                                           # centroids -- length
    bytes.fromhex("0000403f"), bytes.fromhex("000060bf"),   #   0.75, -0.875
    bytes.fromhex("00000000"), bytes.fromhex("00004040"),   #   0.0, 3.0
    u64(3), u64(0), u64(0), u64(1),                   # assignments: Vec<usize>
    u64(2),                                           # ids: Vec<Vec<usize>> -- outer length
    u64(2), u64(0), u64(1),                           #   ids[0]
    u64(1), u64(2),                                   #   ids[1]
])


def test_expected_bytes_are_what_the_hand_rules_say():
    assert len(EXPECTED) == 8 + (8 + 24) + (8 + 16) + (8 + 24) + 8 + (8 + 16) + (8 + 8)
    assert EXPECTED[16:20] == f32(1.0) and EXPECTED[32:36] == f32(-0.0) and EXPECTED[48:52] == f32(0.75)


def test_python_writer_and_reader(tmp_path):
    p = os.path.join(tmp_path, "ivfflat.index")
    write_index_file(p, 2, VALUES, CENTROIDS, ASSIGN, IDS)
    assert open(p, "rb").read() == EXPECTED
    f = read_index_file(p, 2)
    assert f["num_centroids"] == 2 and f["ids"] == IDS
    assert np.array_equal(f["values"].view(np.uint32), VALUES.view(np.uint32))
    assert np.array_equal(f["centroids"].view(np.uint32), CENTROIDS.view(np.uint32))
    assert np.array_equal(f["assignments"], ASSIGN)
    open(p, "wb").write(EXPECTED[:-3])                # truncated file -> the reference's io::Error text
    with pytest.raises(IOError, match="Deserialization error"):
        read_index_file(p, 2)


def test_empty_index_file(tmp_path):
    """build_index with zero attempts keeps nothing (ivfflat.rs:109-110): empty centroids / assignments, k empty lists."""
    p = os.path.join(tmp_path, "empty.index")
    write_index_file(p, 3, VALUES, np.zeros((0, 2), np.float32), np.zeros(0, np.uint64), [[], [], []])
    raw = open(p, "rb").read()
    assert raw == u64(3) + EXPECTED[8:40] + u64(0) + u64(0) + u64(3) + u64(0) * 3
    f = read_index_file(p, 2)
    assert f["centroids"].shape == (0, 2) and f["ids"] == [[], [], []]


def test_cpp_mirror_writes_and_reads_the_same_bytes(tmp_path):
    exe = os.path.join(tmp_path, "index_file_demo")
    lib = vbuild.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "index_file_demo.cpp"),
                           "-L" + os.path.dirname(lib), "-lvers_hip", "-Wl,-rpath," + os.path.dirname(lib)])
    cpp_out = os.path.join(tmp_path, "cpp.index")
    subprocess.check_call([exe, "write", cpp_out])
    assert open(cpp_out, "rb").read() == EXPECTED                  # C++-saved == hand-written bytes
    f = read_index_file(cpp_out, 2)                                # ... and Python-loaded
    assert f["ids"] == IDS and np.array_equal(f["values"].view(np.uint32), VALUES.view(np.uint32))
    py_out = os.path.join(tmp_path, "py.index")
    write_index_file(py_out, 2, VALUES, CENTROIDS, ASSIGN, IDS)
    re_out = os.path.join(tmp_path, "resaved.index")
    subprocess.check_call([exe, "resave", py_out, re_out])         # Python-saved -> C++-loaded -> C++-saved
    assert open(re_out, "rb").read() == EXPECTED
