"""The C-ABI library loads without a GPU and exports every function that include/*.h declares."""
import ctypes
import os
import re

import pytest

from raytracing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = _lib.lib()
    names = declared_functions("rt_amd.h") + declared_functions("rt_host.h")
    assert len(names) >= 20
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/ but not exported by librt_amd.so"
    assert sorted(names) == sorted(_lib.DEVICE_SYMBOLS + _lib.HOST_SYMBOLS)


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_lib.RtInode) == 60
    assert _lib.HIT_DTYPE.itemsize == 32 and _lib.TRI_REF_DTYPE.itemsize == 8
    assert _lib.lib().rt_abi_version() == 7


def test_errors_are_reported_not_crashed():
    L = _lib.lib()
    rc = L.rt_scene_create(None, 0, None)
    assert rc == -1 and b"null" in L.rt_last_error()
    rc = L.rt_intersect(None, None, None, ctypes.c_uint64(0), ctypes.c_float(0), 0, None)
    assert rc == -1


def test_no_fallback_without_library(tmp_path, monkeypatch):
    """The Python layer has no CPU path: a missing library is an ImportError, not a silent fallback."""
    import importlib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    try:
        _lib.lib()
        raised = False
    except ImportError:
        raised = True
    assert raised
    importlib.reload(_lib)


def test_cpp_example_fails_loudly_without_a_gpu(tmp_path):
    """The C++ host layer has no CPU fallback either: without a HIP device the program stops with the
    library's error message."""
    import os
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "render_builtin")
    if not os.path.exists(exe):
        pytest.skip("examples/render_builtin not built")
    run = subprocess.run([exe, "64", "48", "1", "1", str(tmp_path / "x.ppm")], capture_output=True, text=True, timeout=120)
    assert run.returncode == 1 and "no HIP device" in run.stderr
    assert not (tmp_path / "x.ppm").exists()
