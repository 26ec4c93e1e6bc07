"""ray-tracer_amd — MI355X-native photon transport pass behind the reference's
Renderer / SceneLoader interfaces.

The directory name carries a hyphen (it is the reference's name), so import it with
    importlib.import_module("ray-tracer_amd")
Python here is glue only: it loads the two in-tree shared libraries

    host/librt_host.so   C++ scene loader / frame renderer   (include/rt_host.h)
    csrc/librt_amd.so    HIP tracer behind the C-ABI          (include/rt_amd.h)

and exposes thin ctypes wrappers.  Nothing in this package computes photon paths
on the CPU; if the HIP library is missing or no GPU is present the tracer wrappers
raise instead of falling back.
"""
import ctypes as _C
import os as _os

from .scene_types import (RtCamera, RtLight, RtMaterial, RtMesh, RtScene,  # noqa: F401
                          RtSettings, RtVolume)

_HERE = _os.path.dirname(_os.path.abspath(__file__))
HOST_LIB_PATH = _os.path.join(_HERE, "host", "librt_host.so")
TRACER_LIB_PATH = _os.path.join(_HERE, "csrc", "librt_amd.so")
REPO_ROOT = _os.path.dirname(_HERE)

_host = None


def host_lib():
    """The host-side library (scene loading, frame rendering)."""
    global _host
    if _host is None:
        if not _os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is not built; run __graft_entry__.build()")
        lib = _C.CDLL(HOST_LIB_PATH, mode=_C.RTLD_GLOBAL)
        lib.rth_last_error.restype = _C.c_char_p
        lib.rth_scene_load.argtypes = [_C.c_char_p, _C.POINTER(_C.c_void_p)]
        lib.rth_scene_get.restype = _C.POINTER(RtScene)
        lib.rth_scene_get.argtypes = [_C.c_void_p]
        lib.rth_scene_settings.restype = _C.POINTER(RtSettings)
        lib.rth_scene_settings.argtypes = [_C.c_void_p]
        lib.rth_scene_write.argtypes = [_C.c_void_p, _C.c_char_p]
        lib.rth_scene_free.argtypes = [_C.c_void_p]
        lib.rth_write_png16.argtypes = [_C.c_char_p, _C.c_void_p, _C.c_int32, _C.c_int32, _C.c_char_p]
        lib.rth_material_sample.argtypes = [_C.c_void_p, _C.c_int32, _C.POINTER(_C.c_double), _C.c_uint32,
                                            _C.POINTER(_C.c_double)]
        lib.rth_material_evaluate.argtypes = [_C.c_void_p, _C.c_int32, _C.POINTER(_C.c_double), _C.POINTER(_C.c_double)]
        _host = lib
    return _host


class Scene:
    """A loaded + flattened scene (owner of an rth_scene handle)."""

    def __init__(self, json_path):
        lib = host_lib()
        handle = _C.c_void_p()
        if lib.rth_scene_load(str(json_path).encode(), _C.byref(handle)) != 0:
            raise RuntimeError(lib.rth_last_error().decode())
        self._handle = handle
        self.path = str(json_path)

    @property
    def ptr(self):
        """POINTER(RtScene) for passing to the C-ABI / the oracle."""
        return host_lib().rth_scene_get(self._handle)

    @property
    def desc(self):
        return self.ptr.contents

    @property
    def settings(self):
        return host_lib().rth_scene_settings(self._handle).contents

    def material_sample(self, material, incident, normal, seed):
        """Material::sample of the scene's material `material` with draws from RandomGenerator(seed):
        (valid, direction[3], weight[3], pdf, draws consumed)."""
        rows = (_C.c_double * 6)(*incident, *normal)
        out = (_C.c_double * 9)()
        if host_lib().rth_material_sample(self._handle, material, rows, seed, out) != 0:
            raise RuntimeError(host_lib().rth_last_error().decode())
        return list(out)

    def material_evaluate(self, material, wi, wo, normal):
        """Material::evaluate / pdf / isDelta: (f[3], pdf, isDelta)."""
        rows = (_C.c_double * 9)(*wi, *wo, *normal)
        out = (_C.c_double * 5)()
        if host_lib().rth_material_evaluate(self._handle, material, rows, out) != 0:
            raise RuntimeError(host_lib().rth_last_error().decode())
        return list(out)

    def write(self, path):
        if host_lib().rth_scene_write(self._handle, str(path).encode()) != 0:
            raise RuntimeError(host_lib().rth_last_error().decode())

    def close(self):
        if self._handle:
            host_lib().rth_scene_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from .tracer import Tracer, tracer_lib  # noqa: E402,F401
