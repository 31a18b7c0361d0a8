"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol the
header declares, and the Python binding table matches the header (no compute: no GPU here)."""
import ctypes
import os
import re

from vers_amd import build as vbuild
from vers_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(header="vers_hip.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vers_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    so = vbuild.build()
    lib = ctypes.CDLL(so)
    names = header_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vers_hip.h but not exported"


def test_binding_table_matches_header():
    assert sorted(capi.SIGNATURES) == header_functions()


def test_product_boundary_carries_no_test_hooks():
    """The drop-in boundary (include/vers_hip.h, libvers_hip.so) declares and exports nothing named *test*: the test / emulation
    hooks are a second library (libvers_hip_test.so, include/vers_hip_test.h) that links against the product one."""
    import subprocess
    from vers_amd import testhooks
    assert not [n for n in header_functions() if "test" in n]
    so = vbuild.build()
    dyn = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported_c = [l.split()[-1] for l in dyn.splitlines() if l.split()[-1].startswith("vers_")]
    assert exported_c and not [n for n in exported_c if "test" in n], [n for n in exported_c if "test" in n]
    assert sorted(exported_c) == header_functions(), sorted(set(exported_c) ^ set(header_functions()))   # and nothing undeclared either
    names = header_functions("vers_hip_test.h")
    names = [n for n in names if "test" in n]   # (the header includes vers_hip.h's names by reference only in comments)
    assert sorted(testhooks.SIGNATURES) == names and len(names) == 6
    tso = vbuild.build_test_hooks()
    needed = subprocess.run(["readelf", "-d", tso], capture_output=True, text=True).stdout
    assert "libvers_hip.so" in needed and "$ORIGIN" in needed
    lib = ctypes.CDLL(vbuild.LIB, mode=ctypes.RTLD_GLOBAL)
    tlib = ctypes.CDLL(tso)
    for n in names:
        assert hasattr(tlib, n) and not hasattr(lib, n), n


def test_status_codes_match_header():
    txt = open(os.path.join(ROOT, "include", "vers_hip.h")).read()
    for name, val in [("VERS_OK", capi.OK), ("VERS_ERR_INVALID", capi.ERR_INVALID), ("VERS_ERR_NAN", capi.ERR_NAN),
                      ("VERS_ERR_INSUFFICIENT", capi.ERR_INSUFFICIENT), ("VERS_ERR_HIP", capi.ERR_HIP),
                      ("VERS_ERR_EMPTY", capi.ERR_EMPTY), ("VERS_MAX_TOPK", capi.MAX_TOPK)]:
        assert re.search(rf"#define {name} {val}\b", txt), name


def test_rccl_adapter_builds_exports_every_declared_symbol_and_matches_its_binding():
    """libvers_rccl.so (include/vers_comm_rccl.h): the optional RCCL adapter.  Loads without a GPU (it only links librccl),
    exports what its header declares, and libvers_hip.so itself does NOT depend on RCCL."""
    import subprocess
    from vers_amd import rccl
    import os
    import pytest
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(vbuild._hipcc())), "lib")
    if not any(f.startswith("librccl.so") for f in os.listdir(rocm_lib)):
        pytest.skip("no librccl under the hipcc prefix: the optional adapter is not built on this box")
    so = vbuild.build_rccl()
    names = header_functions("vers_comm_rccl.h")
    assert "vers_rccl_gather" in names and "vers_rccl_comm" in names
    assert sorted(rccl.SIGNATURES) == names
    lib = ctypes.CDLL(so)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vers_comm_rccl.h but not exported"
    needed = subprocess.run(["readelf", "-d", vbuild.LIB], capture_output=True, text=True).stdout
    assert "rccl" not in needed, "libvers_hip.so must not link RCCL: the adapter is a separate, optional library"
    assert "rccl" in subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
