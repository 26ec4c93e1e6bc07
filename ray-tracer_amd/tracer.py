"""ctypes wrapper of the tracer's C-ABI (include/rt_amd.h).

Loads csrc/librt_amd.so and raises if it is missing — there is no CPU path to fall
back to.  Arrays cross the boundary as numpy buffers (host pointers) or as raw
device pointers taken from torch tensors (camera buffers).
"""
import ctypes as C
import os

import numpy as np

SINK_NONE, SINK_PROBE, SINK_SPLAT = 0, 1, 2
RT_OK = 0

COUNTER_FIELDS = (
    "photons", "casts", "bounces", "deposits", "kept", "culled", "store_attempted", "store_size",
    "store_dropped", "splat_total", "splat_radius_clamped", "splat_luminance_clamped", "density_cells",
    "density_deposits", "density_dropped",
)


class RtCounters(C.Structure):
    _fields_ = [(name, C.c_uint64) for name in COUNTER_FIELDS] + [("deposited_power", C.c_double * 3)]


TRACE_REUSE_ORDER = 1  # rt_trace_params.flags


class RtTraceParams(C.Structure):
    _fields_ = [
        ("sink_mode", C.c_int32), ("flags", C.c_int32), ("first_photon", C.c_uint64),
        ("photon_count", C.c_uint64), ("record_photons", C.c_uint64),
    ]


# every symbol include/rt_amd.h declares, with its ctypes signature
_SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_struct_size": (C.c_uint64, [C.c_int32]),
    "rt_device_count": (C.c_int32, []),
    "rt_check_tree": (C.c_int32, [C.POINTER(C.c_double), C.c_int64, C.c_int32, C.c_double, C.POINTER(C.c_int64)]),
    "rt_create": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_set_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "rt_set_probes": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint64]),
    "rt_set_probes_from_cameras": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rt_set_store_limit": (C.c_int32, [C.c_void_p, C.c_uint64]),
    "rt_deposit_emitters": (C.c_int32, [C.c_void_p, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rt_set_splat": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_uint64]),
    "rt_bind_camera_buffer": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "rt_reset": (C.c_int32, [C.c_void_p]),
    "rt_trace": (C.c_int32, [C.c_void_p, C.POINTER(RtTraceParams)]),
    "rt_trace_static": (C.c_int32, [C.c_void_p, C.POINTER(RtTraceParams)]),
    "rt_trace_static_max_photons": (C.c_uint64, []),
    "rt_debug_device_allocations": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rt_debug_device_memory": (C.c_int32, [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rt_synchronize": (C.c_int32, [C.c_void_p]),
    "rt_allreduce": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_comm_init_all": (C.c_int32, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "rt_comm_destroy": (C.c_int32, [C.c_void_p]),
    "rt_read_timings": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rt_read_setup_timings": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rt_forget_gather_points": (C.c_int32, [C.c_void_p]),
    "rt_read_emission_order": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rt_read_counters": (C.c_int32, [C.c_void_p, C.POINTER(RtCounters)]),
    "rt_read_camera_buffer": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "rt_read_store": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rt_read_density": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                    C.POINTER(C.c_uint64)]),
    "rt_read_records": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "rt_collect_gather_points": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                             C.c_void_p]),
    "rt_store_append": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "rt_gather": (C.c_int32, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_void_p]),
    "rt_probe_keep_test": (C.c_int32, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rt_cast_rays": (C.c_int32, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "rt_emit": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rt_intersect_triangles": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "rt_intersect_bounds": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "rt_intersect_spheres": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "rt_intersect_quads": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "rt_intersect_planes": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "rt_bsdf_sample": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "rt_bsdf_evaluate": (C.c_int32, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
}

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIBRARY selects another build of the same library (e.g. the census build)
TRACER_LIB_PATH = os.environ.get("RT_AMD_LIBRARY") or os.path.join(_HERE, "csrc", "librt_amd.so")
_lib = None


def declared_symbols():
    return sorted(_SIGNATURES)


def tracer_lib():
    """The HIP tracer library; raises when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(TRACER_LIB_PATH):
            raise RuntimeError(f"{TRACER_LIB_PATH} is not built; run __graft_entry__.build() "
                               "(there is no CPU fallback for the photon pass)")
        # Another build named by RT_AMD_LIBRARY binds its own symbols first: the host library has
        # already brought the regular build into the process, whose symbols would otherwise win.
        override = bool(os.environ.get("RT_AMD_LIBRARY"))
        lib = C.CDLL(TRACER_LIB_PATH, mode=(os.RTLD_NOW | os.RTLD_DEEPBIND) if override else C.DEFAULT_MODE)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


class TracerError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt_amd error {code}: {message}")
        self.code = code


def _check(code):
    if code != RT_OK:
        raise TracerError(code, tracer_lib().rt_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def comm_init_all(devices):
    """One RCCL communicator per listed device, made in this process (rt_comm_init_all)."""
    n = len(devices)
    ids = (C.c_int32 * n)(*devices)
    comms = (C.c_void_p * n)()
    _check(tracer_lib().rt_comm_init_all(n, ids, comms))
    return [C.c_void_p(c) for c in comms]


def comm_destroy(comm):
    _check(tracer_lib().rt_comm_destroy(comm))


def device_allocations():
    """(hipMalloc calls, hipFree calls) the library has made in this process so far."""
    made, freed = C.c_uint64(), C.c_uint64()
    tracer_lib().rt_debug_device_allocations(C.byref(made), C.byref(freed))
    return int(made.value), int(freed.value)


def device_memory(device=0):
    """(free, total) bytes of a GPU, from hipMemGetInfo."""
    free, total = C.c_uint64(), C.c_uint64()
    _check(tracer_lib().rt_debug_device_memory(device, C.byref(free), C.byref(total)))
    return int(free.value), int(total.value)


def rows(fn_name, inputs, out_cols, out_dtype=np.float64, device=0):
    """Run a device function-level entry point over rows of `inputs`."""
    a = np.ascontiguousarray(inputs, dtype=np.float64)
    shape = (a.shape[0],) + ((out_cols,) if out_cols else ())
    out = np.zeros(shape, dtype=out_dtype)
    _check(getattr(tracer_lib(), fn_name)(device, a.shape[0], _ptr(a), _ptr(out)))
    return out


class Tracer:
    """One rt_tracer handle: a scene resident on one GPU."""

    def __init__(self, scene, device=0):
        self.scene = scene
        self.device = device
        handle = C.c_void_p()
        _check(tracer_lib().rt_create(C.cast(scene.ptr, C.c_void_p), device, C.byref(handle)))
        self._handle = handle
        desc = scene.desc
        self.cameras = [(desc.cameras[c].width, desc.cameras[c].height) for c in range(desc.camera_count)]
        self.light_count = int(desc.light_count)
        self.max_bounces = int(desc.settings.bounce_threshold) + 1

    def close(self):
        if getattr(self, "_handle", None):
            tracer_lib().rt_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- set-up -------------------------------------------------------------
    def set_stream(self, hip_stream):
        _check(tracer_lib().rt_set_stream(self._handle, C.c_void_p(hip_stream)))

    def set_probes(self, probes, cell_size, keep_radius, store_capacity):
        probes = np.ascontiguousarray(probes, dtype=np.float64).reshape(-1, 3)
        _check(tracer_lib().rt_set_probes(self._handle, _ptr(probes), probes.shape[0], cell_size, keep_radius,
                                          store_capacity))

    def set_probes_from_cameras(self, cell_size, keep_radius, records_per_probe=256, max_store_capacity=40_000_000):
        """Probes = the probe-pass records of the gathering cameras, kept on the device.  Returns
        (probe count, store capacity)."""
        probes, capacity = C.c_uint64(), C.c_uint64()
        _check(tracer_lib().rt_set_probes_from_cameras(self._handle, cell_size, keep_radius, records_per_probe,
                                                       max_store_capacity, C.byref(probes), C.byref(capacity)))
        return int(probes.value), int(capacity.value)

    def set_store_limit(self, max_records):
        """One rank's share of the store (0: its whole capacity); see rt_set_store_limit."""
        _check(tracer_lib().rt_set_store_limit(self._handle, C.c_uint64(max_records)))

    def deposit_emitters(self, deposit_spacing):
        """Emitter-face deposits straight into the store; returns (generated, kept)."""
        samples, kept = C.c_uint64(), C.c_uint64()
        _check(tracer_lib().rt_deposit_emitters(self._handle, deposit_spacing, C.byref(samples), C.byref(kept)))
        return int(samples.value), int(kept.value)

    def set_splat(self, min_splat_radius, density_cell_size=0.0, density_capacity=1 << 20):
        _check(tracer_lib().rt_set_splat(self._handle, min_splat_radius, density_cell_size, density_capacity))

    def bind_camera_buffer(self, camera, device_pointer):
        _check(tracer_lib().rt_bind_camera_buffer(self._handle, camera, C.c_void_p(device_pointer)))

    def reset(self):
        _check(tracer_lib().rt_reset(self._handle))

    # -- the photon pass ----------------------------------------------------
    def trace(self, sink_mode=SINK_NONE, first_photon=0, photon_count=0, record_photons=0, reuse_order=False):
        """reuse_order: the emission order of the last pass over the same range may be used again
        (RT_TRACE_REUSE_ORDER): same results, no sort."""
        params = RtTraceParams(sink_mode, TRACE_REUSE_ORDER if reuse_order else 0, first_photon, photon_count, record_photons)
        _check(tracer_lib().rt_trace(self._handle, C.byref(params)))

    def trace_static(self, sink_mode=SINK_NONE, first_photon=0, photon_count=0):
        """The pass through the entry point that allocates nothing (at most
        rt_trace_static_max_photons() ids per call)."""
        params = RtTraceParams(sink_mode, 0, first_photon, photon_count, 0)
        _check(tracer_lib().rt_trace_static(self._handle, C.byref(params)))

    def synchronize(self):
        _check(tracer_lib().rt_synchronize(self._handle))

    def timings(self):
        """(sort_ms, trace_ms) of the last pass, from HIP events on the tracer's stream."""
        sort_ms, trace_ms = C.c_float(), C.c_float()
        _check(tracer_lib().rt_read_timings(self._handle, C.byref(sort_ms), C.byref(trace_ms)))
        return float(sort_ms.value), float(trace_ms.value)

    def setup_timings(self):
        """Device ms of the camera-side set-up so far: probe pass, probe index (cells), its table of bricks."""
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _check(tracer_lib().rt_read_setup_timings(self._handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"probe_pass": float(a.value), "index_build": float(b.value), "filter_build": float(c.value)}

    def forget_gather_points(self):
        """The next collect / count of gather points runs the probe pass again."""
        _check(tracer_lib().rt_forget_gather_points(self._handle))

    def emission_order(self):
        """Photon ids of the last pass (numbered light by light from 0) in the order they were traced;
        empty when the pass was traced unsorted."""
        n = C.c_uint64(0)
        _check(tracer_lib().rt_read_emission_order(self._handle, None, 0, C.byref(n)))
        out = np.empty(int(n.value), dtype=np.uint32)
        if out.size:
            _check(tracer_lib().rt_read_emission_order(self._handle, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
        return out

    # -- read-back ----------------------------------------------------------
    def counters(self):
        c = RtCounters()
        _check(tracer_lib().rt_read_counters(self._handle, C.byref(c)))
        out = {name: int(getattr(c, name)) for name in COUNTER_FIELDS}
        out["deposited_power"] = np.array(list(c.deposited_power))
        return out

    def allreduce(self, comm, want_totals=True):
        """Sum the camera buffers over the ranks of an RCCL communicator (see comm_init_all); returns
        the counters summed over the ranks (None when want_totals is False: the call then passes NULL,
        which changes nothing about what is reduced)."""
        c = RtCounters()
        _check(tracer_lib().rt_allreduce(self._handle, comm, C.byref(c) if want_totals else None))
        if not want_totals:
            return None
        out = {name: int(getattr(c, name)) for name in COUNTER_FIELDS}
        out["deposited_power"] = np.array(list(c.deposited_power))
        return out

    def camera_buffer(self, camera):
        w, h = self.cameras[camera]
        out = np.zeros((h, w, 3), dtype=np.float32)
        _check(tracer_lib().rt_read_camera_buffer(self._handle, camera, _ptr(out)))
        return out

    def store(self, max_records=None):
        n = C.c_uint64()
        _check(tracer_lib().rt_read_store(self._handle, None, 0, C.byref(n)))
        count = int(n.value) if max_records is None else min(int(n.value), max_records)
        out = np.zeros((count, 13), dtype=np.float32)
        if count:
            _check(tracer_lib().rt_read_store(self._handle, _ptr(out), count, C.byref(n)))
        return out

    def density(self):
        n = C.c_uint64()
        _check(tracer_lib().rt_read_density(self._handle, None, None, None, 0, C.byref(n)))
        count = int(n.value)
        keys = np.zeros((count, 3), dtype=np.int64)
        power = np.zeros((count, 3), dtype=np.float32)
        counts = np.zeros((count,), dtype=np.uint64)
        if count:
            _check(tracer_lib().rt_read_density(self._handle, _ptr(keys), _ptr(power), _ptr(counts), count,
                                                C.byref(n)))
        return keys, power, counts

    def records(self):
        n = C.c_uint64()
        _check(tracer_lib().rt_read_records(self._handle, C.byref(n), None, None, None, None, None))
        m, b = int(n.value), self.max_bounces
        out = {
            "emitted": np.zeros((m, 6)), "casts_valid": np.zeros((m, 2), dtype=np.int32),
            "bounce_f64": np.zeros((m, b, 10)), "bounce_i32": np.zeros((m, b, 8), dtype=np.int32),
            "bounce_power": np.zeros((m, b, 3), dtype=np.float32),
        }
        if m:
            _check(tracer_lib().rt_read_records(self._handle, C.byref(n), _ptr(out["emitted"]),
                                                _ptr(out["casts_valid"]), _ptr(out["bounce_f64"]),
                                                _ptr(out["bounce_i32"]), _ptr(out["bounce_power"])))
        return out

    # -- the probe pass -----------------------------------------------------
    def collect_gather_points(self, camera=0):
        """(rows[n, 19], stats) of one camera's probe pass; see rt_collect_gather_points."""
        n = C.c_uint64()
        stats = np.zeros(3, dtype=np.uint64)
        _check(tracer_lib().rt_collect_gather_points(self._handle, camera, None, 0, C.byref(n), _ptr(stats)))
        out = np.zeros((int(n.value), 19))
        if n.value:
            _check(tracer_lib().rt_collect_gather_points(self._handle, camera, _ptr(out), n.value, C.byref(n),
                                                         _ptr(stats)))
        names = ("camera_rays", "delta_extensions", "misses")
        return out, dict(zip(names, (int(v) for v in stats)))

    def count_gather_points(self, camera=0):
        """Run one camera's probe pass and leave the records on the device; returns their number."""
        n = C.c_uint64()
        stats = np.zeros(3, dtype=np.uint64)
        _check(tracer_lib().rt_collect_gather_points(self._handle, camera, None, 0, C.byref(n), _ptr(stats)))
        return int(n.value)

    def collect_probe_positions(self):
        """Gather-point positions of every camera, in camera order (what the reference's
        Renderer feeds into ProbeIndex, src/Renderer.cpp:101-131)."""
        desc = self.scene.desc
        parts = [np.zeros((0, 3))]
        for c in range(len(self.cameras)):
            # debug cameras (bounce / light filter) run no gather and contribute no probes
            if desc.cameras[c].bounce_filter >= 0 or desc.cameras[c].light_filter >= 0:
                continue
            parts.append(self.collect_gather_points(c)[0][:, 2:5])
        return np.ascontiguousarray(np.concatenate(parts, axis=0))

    def store_append(self, records):
        records = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 13)
        _check(tracer_lib().rt_store_append(self._handle, _ptr(records), records.shape[0]))

    def gather(self, camera, gather_cell_size, min_gather_radius):
        """Index the store and add the camera's gathered radiance into its buffer; returns the stats."""
        stats = np.zeros(4, dtype=np.uint64)
        _check(tracer_lib().rt_gather(self._handle, camera, gather_cell_size, min_gather_radius, _ptr(stats)))
        return dict(zip(("pixels_hit", "pixels_gathered", "deposits_accum", "unindexed"), (int(v) for v in stats)))

    def probe_keep_test(self, points):
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        keep = np.zeros(points.shape[0], dtype=np.uint8)
        _check(tracer_lib().rt_probe_keep_test(self._handle, points.shape[0], _ptr(points), _ptr(keep)))
        return keep.astype(bool)

    # -- function-level -----------------------------------------------------
    def cast_rays(self, rays, self_hit=1e-4):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        f = np.zeros((n, 7))
        i = np.zeros((n, 4), dtype=np.int32)
        _check(tracer_lib().rt_cast_rays(self._handle, n, _ptr(rays), self_hit, _ptr(f), _ptr(i)))
        return f, i

    def emit(self, light, draws5, total_photons):
        draws5 = np.ascontiguousarray(draws5, dtype=np.float64).reshape(-1, 5)
        n = draws5.shape[0]
        rays = np.zeros((n, 6))
        colors = np.zeros((n, 3), dtype=np.float32)
        _check(tracer_lib().rt_emit(self._handle, light, n, _ptr(draws5), total_photons, _ptr(rays), _ptr(colors)))
        return rays, colors
