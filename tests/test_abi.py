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


def test_trace_kernels_keep_two_waves_per_simd():
    """The trace kernels sit at the 256-register edge: one register more and the compiler falls
    back to one wave per SIMD, which halves the throughput without failing anything else.  The
    build keeps the compiler's resource remarks; every trace variant that a frame runs (no
    per-photon records: those variants serve the parity tests) must report two waves per SIMD, and
    every variant no scratch and no spilled vector registers."""
    import re
    path = os.path.join(REPO, "ray-tracer_amd", "csrc", "build", "rt_kernels.resources")
    if not os.path.exists(path):
        pytest.skip("no resource remarks: build ray-tracer_amd/csrc with make first")
    text = open(path).read()
    blocks = re.split(r"remark: Function Name: ", text)[1:]
    seen = 0
    for block in blocks:
        name = block.split()[0]
        if "traceKernel" not in name:
            continue
        seen += 1
        fields = dict(re.findall(r"remark:\s+([A-Za-z /\[\]]+?): (\w+) \[", block))
        records = re.search(r"traceKernelILi\dELb1E", name) is not None
        if not records:
            assert fields["Occupancy [waves/SIMD]"] == "2", (name, fields)
        assert fields["ScratchSize [bytes/lane]"] == "0", (name, fields)
        assert fields["VGPRs Spill"] == "0", (name, fields)
    assert seen == 24
