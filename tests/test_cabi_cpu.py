"""CPU-side checks of the C-ABI library: it loads, exports every symbol the
header declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from slam_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    build.build()
    return api.lib()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "slam_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(L):
    names = header_functions()
    assert len(names) >= 45
    for n in names:
        assert hasattr(L, n), "libslam_mi355x.so does not export %s" % n
    assert sorted(api.EXPORTS) == names


def test_version_and_defaults(L):
    assert b"gfx950" in L.slam_version()
    p = api.icp_default_params()
    # icp.cpp:27
    assert (p.max_iter, p.min_delta, p.mode) == (20, 1e-6, api.ICP_P2P)
    g = api.grid_default_params()
    # mls.h:161,165,188,189
    assert (g.max_range, g.occupancy_increment, g.occupancy_decrement, g.min_cluster_points) == \
        (75.0, 1.0, 0.3, 10)


def test_argument_errors_do_not_need_a_device(L):
    h = C.c_void_p()
    m = np.zeros((2, 2))
    # icp.cpp:38-43: fewer than 5 model points
    rc = L.slam_icp_create(m.ctypes.data_as(C.c_void_p), 2, m.ctypes.data_as(C.c_void_p), 2, None,
                           C.byref(h))
    assert rc == api.E_TOO_FEW_MODEL and b"at least 5 model points" in L.slam_last_error()
    assert L.slam_grid_create(0, 10, 0.1, None, C.byref(h)) == api.E_INVALID


def _no_gpu():
    return api.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present")
def test_compute_fails_loudly_without_gpu(L):
    with pytest.raises(api.SlamError) as e:
        api.Icp(np.random.randn(10, 2), np.random.randn(10, 2))
    assert e.value.code == api.E_NO_DEVICE and "no CPU path" in str(e.value)
    with pytest.raises(api.SlamError) as e:
        api.Grid(100, 100, 0.1)
    assert e.value.code == api.E_NO_DEVICE
    with pytest.raises(api.SlamError):
        api.DeviceArray((16,), np.float32)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                for line in txt.splitlines():
                    if re.match(r"\s*(#\s*include|import|from)\b", line):
                        assert "oracle" not in line, "%s: %s" % (f, line)
                assert "CDLL" not in txt or f == "api.py", f


def test_python_binding_declares_pointer_signatures():
    """ctypes passes an undeclared Python int as a 32-bit C int: a device pointer handed to an entry point
    without argtypes would be truncated (a GPU fault, not an error code).  Only calls that take no pointer
    from Python, or a byref() struct, may go undeclared."""
    from slam_amd import api
    L = api.lib()
    undeclared = {n for n in api.EXPORTS if getattr(L, n).argtypes is None}
    assert undeclared <= {"slam_last_error", "slam_version", "slam_device_count", "slam_set_device",
                          "slam_device_synchronize", "slam_icp_default_params", "slam_grid_default_params",
                          "slam_ccicp_create"}


def test_python_structs_mirror_the_header(tmp_path):
    """slam_amd/api.py restates the parameter structs of include/slam_mi355x.h for ctypes: a field added on one side only
    would shift everything behind it.  A C program prints sizeof and every field's offset; ctypes must agree."""
    import ctypes as C
    import subprocess
    from slam_amd import api
    structs = {"slam_icp_params": api.IcpParams, "slam_grid_params": api.GridParams, "slam_mapper_params": api.MapperParams,
               "slam_gseg_params": api.GsegParams, "slam_icp_result": api.IcpResult}
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "slam_mi355x.h"', "int main(void) {"]
    for name, cls in structs.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (name, name))
        for f, _ in cls._fields_:
            lines.append('printf(" %s=%%zu", offsetof(%s, %s));' % (f, name, f))
        lines.append('printf("\\n");')
    lines += ["return 0;", "}"]
    src, exe = tmp_path / "sizes.c", tmp_path / "sizes"
    src.write_text("\n".join(lines))
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    for line in out.strip().splitlines():
        parts = line.split()
        cls = structs[parts[0]]
        assert int(parts[1]) == C.sizeof(cls), (parts[0], parts[1], C.sizeof(cls))
        for p in parts[2:]:
            f, off = p.split("=")
            assert getattr(cls, f).offset == int(off), (parts[0], f, off, getattr(cls, f).offset)
