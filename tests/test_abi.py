"""The C-ABI libraries load and export what the headers declare (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from conftest import REPO


def header_functions(path, prefix):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_tracer_library_exports_every_declared_symbol(rt):
    declared = header_functions(os.path.join(REPO, "include", "rt_amd.h"), "rt_")
    assert len(declared) >= 25
    lib = C.CDLL(rt.TRACER_LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"librt_amd.so does not export {name}"
    # the Python binding covers exactly the declared set
    from importlib import import_module
    tracer = import_module("ray-tracer_amd.tracer")
    assert tracer.declared_symbols() == declared


def test_host_library_exports_every_declared_symbol(rt):
    declared = header_functions(os.path.join(REPO, "include", "rt_host.h"), "rth_")
    lib = C.CDLL(rt.HOST_LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"librt_host.so does not export {name}"


def test_struct_sizes_match_ctypes(rt):
    from importlib import import_module
    tracer = import_module("ray-tracer_amd.tracer")
    lib = tracer.tracer_lib()
    expected = [rt.RtScene, rt.RtVolume, rt.RtMesh, rt.RtMaterial, rt.RtLight, rt.RtCamera, rt.RtSettings,
                tracer.RtCounters, tracer.RtTraceParams]
    for which, struct in enumerate(expected):
        assert lib.rt_abi_struct_size(which) == C.sizeof(struct), struct.__name__
    assert lib.rt_abi_struct_size(99) == 0


def test_no_gpu_means_loud_failure_not_fallback(rt):
    """Without a GPU every computing entry point fails with RT_ERR_DEVICE."""
    from importlib import import_module
    tracer = import_module("ray-tracer_amd.tracer")
    lib = tracer.tracer_lib()
    if lib.rt_device_count() > 0:
        pytest.skip("a GPU is present")
    scene = rt.Scene(os.path.join(REPO, "tests", "scenes", "furnace.json"))
    with pytest.raises(tracer.TracerError) as info:
        rt.Tracer(scene)
    assert info.value.code == 2
    import numpy as np
    with pytest.raises(tracer.TracerError):
        tracer.rows("rt_intersect_spheres", np.zeros((1, 10)), 8)
