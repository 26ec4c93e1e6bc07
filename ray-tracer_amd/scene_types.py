"""ctypes mirrors of the plain-data structs in include/rt_scene.h.

Field order and types must match the header exactly; tests/test_abi.py checks the
sizes against the compiled libraries (rt_abi_struct_size).
"""
import ctypes as C


class RtVolume(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("material", C.c_int32), ("mesh", C.c_int32), ("animation", C.c_int32),
        ("position", C.c_double * 3), ("rotation", C.c_double * 4), ("param", C.c_double * 18),
    ]


class RtMesh(C.Structure):
    _fields_ = [("first_triangle", C.c_int64), ("triangle_count", C.c_int64), ("name", C.c_char * 64)]


class RtMaterial(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("reserved", C.c_int32), ("color", C.c_float * 3), ("reserved2", C.c_float),
        ("alpha", C.c_double), ("ior", C.c_double), ("name", C.c_char * 64),
    ]


class RtLight(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("shape", C.c_int32), ("position", C.c_double * 3), ("rotation", C.c_double * 4),
        ("color", C.c_float * 3), ("reserved", C.c_float), ("luminous_flux", C.c_double),
        ("intensity_candela", C.c_double), ("flux_override", C.c_double), ("angle", C.c_double),
        ("inner_radius", C.c_double), ("max_angle", C.c_double), ("linearity", C.c_double),
        ("radius", C.c_double), ("width", C.c_double), ("height", C.c_double), ("name", C.c_char * 64),
    ]


class RtCamera(C.Structure):
    _fields_ = [
        ("position", C.c_double * 3), ("rotation", C.c_double * 4), ("vertical_fov_degrees", C.c_double),
        ("width", C.c_int32), ("height", C.c_int32), ("projection", C.c_int32), ("bounce_filter", C.c_int32),
        ("light_filter", C.c_int32), ("animation", C.c_int32), ("orthographic_height", C.c_double),
        ("aperture_radius", C.c_double), ("focus_distance", C.c_double), ("f_number", C.c_double),
        ("shutter_time", C.c_double), ("iso", C.c_double), ("name", C.c_char * 64),
    ]


class RtSettings(C.Structure):
    _fields_ = [
        ("photons_per_light", C.c_uint64), ("bounce_store_capacity", C.c_uint64), ("fetch_size", C.c_uint64),
        ("worker_count", C.c_uint64), ("probe_sub_sample", C.c_uint64), ("image_width", C.c_int32),
        ("image_height", C.c_int32), ("bounce_threshold", C.c_int32), ("use_probe_gather", C.c_int32),
        ("deterministic", C.c_int32), ("seeded", C.c_int32), ("seed", C.c_uint32),
        ("camera_time_samples", C.c_int32), ("termination_threshold", C.c_double),
        ("hash_grid_cell_size", C.c_double), ("density_cell_scale", C.c_double),
        ("splat_min_radius_scale", C.c_double), ("splat_luminance_clamp", C.c_double),
        ("probe_keep_radius_scale", C.c_double), ("frame_time", C.c_double), ("shutter_time", C.c_double),
    ]


class RtKeyframe(C.Structure):
    _fields_ = [
        ("time", C.c_double), ("value", C.c_double * 3), ("in_tangent", C.c_double * 3),
        ("out_tangent", C.c_double * 3), ("use_auto_tangent", C.c_int32), ("reserved", C.c_int32),
    ]


class RtAnimation(C.Structure):
    _fields_ = [
        ("has_rotation", C.c_int32), ("reserved", C.c_int32), ("rotation_axis", C.c_double * 3),
        ("base_rotation", C.c_double * 4), ("constant_position", C.c_double * 3),
        ("first_position_key", C.c_int64), ("position_key_count", C.c_int64),
        ("first_rotation_key", C.c_int64), ("rotation_key_count", C.c_int64),
    ]


class RtScene(C.Structure):
    _fields_ = [
        ("volume_count", C.c_int64), ("mesh_count", C.c_int64), ("triangle_count", C.c_int64),
        ("material_count", C.c_int64), ("light_count", C.c_int64), ("camera_count", C.c_int64),
        ("volumes", C.POINTER(RtVolume)), ("meshes", C.POINTER(RtMesh)), ("tri_pos", C.POINTER(C.c_double)),
        ("tri_nrm", C.POINTER(C.c_double)), ("materials", C.POINTER(RtMaterial)), ("lights", C.POINTER(RtLight)),
        ("cameras", C.POINTER(RtCamera)), ("settings", RtSettings),
        ("animation_count", C.c_int64), ("keyframe_count", C.c_int64),
        ("animations", C.POINTER(RtAnimation)), ("keyframes", C.POINTER(RtKeyframe)),
    ]
