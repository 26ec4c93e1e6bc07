"""ctypes wrapper of the tracer's C-ABI (include/rt_amd.h) — filled in with the HIP library."""


def tracer_lib():
    raise RuntimeError("librt_amd.so wrapper not available yet")


class Tracer:
    pass
