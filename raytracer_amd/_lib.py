"""ctypes binding of librt_mi355x.so (include/rt_mi355x.h).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RT_LIB", os.path.join(_HERE, "librt_mi355x.so"))  # RT_LIB: A/B builds of the same ABI
_LIB = None

REC_COLS, FINAL_COLS, ROW, ATOM = 48, 32, 32, 12
FLAG_FULL_OUTPUT, FLAG_SINGLECHILD, FLAG_FINAL_ROWS = 1, 2, 4
COMPAT_PY_INDEX, COMPAT_PY_ROOT2 = 1, 2

NOALLOC_MAX_RAYS, NOALLOC_MAX_SCATTERS = 262144, 4094

EXPORTS = ("rt_init", "rt_free", "rt_last_error", "rt_set_surfaces", "rt_set_rayleigh_lut", "rt_trace", "rt_trace_device", "rt_trace_device_noalloc",
           "rt_trace_point_source", "rt_generate_point_source", "rt_trace_camera", "rt_trace_camera_range", "rt_generate_camera_rays", "rt_result_table_device", "rt_set_record_sink", "rt_set_final_rows_device", "rt_match_endpoints", "rt_set_option", "rt_result_num_steps", "rt_result_pass_retries", "rt_result_table",
           "rt_result_step_counts", "rt_result_num_records", "rt_result_records", "rt_result_record_fields", "rt_split_record_fields", "rt_normalize_rays", "rt_order_records", "rt_result_final", "rt_result_timing",
           "rt_result_timing_by_kernel", "rt_device_info")


class Params(C.Structure):
    _fields_ = [("max_scatters", C.c_int32), ("flags", C.c_int32), ("compat", C.c_int32), ("unified_maxiter", C.c_int32),
                ("min_travel_length", C.c_double), ("follow_threshold", C.c_double * 2), ("tir_handling", C.c_double),
                ("seed", C.c_uint64), ("ray_id_base", C.c_int64), ("ray_id_stride", C.c_int64)]


class Camera(C.Structure):
    _fields_ = [("resolution", C.c_int32 * 2), ("lens_type", C.c_int32), ("n_distortion", C.c_int32),
                ("pixel_pitch", C.c_double * 2), ("pixel_center", C.c_double * 2), ("focal_length", C.c_double),
                ("pitch", C.c_double), ("yaw", C.c_double), ("roll", C.c_double), ("distortion", C.c_double * 8),
                ("position", C.c_double * 3), ("polref", C.c_double * 3), ("stokes", C.c_double * 4)]


class MatchParams(C.Structure):
    _fields_ = [("max_dist", C.c_double), ("angle_den", C.c_double), ("cam_incidence_den", C.c_double), ("led_incidence_den", C.c_double),
                ("dist_den", C.c_double), ("flags", C.c_int32), ("reserved", C.c_int32)]


MATCH_DEVICE_POINTERS = 1
LENS_TYPES = {"tan": 0, "theta": 1, "sin": 2}
RECORD_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                          C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int64, C.c_int64)


class RayTracerLibraryError(RuntimeError):
    pass


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise RayTracerLibraryError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(hipcc --offload-arch=gfx950).  raytracer_amd has no CPU path." % SO_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64, and when this library has already pulled in the
    # system's, torch later finds "No HIP GPUs".  Loading torch first (if it is installed) makes both use torch's.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(SO_PATH)
    P = C.POINTER
    L.rt_init.argtypes = [C.c_int, P(C.c_void_p)]
    L.rt_free.argtypes = [C.c_void_p]; L.rt_free.restype = None
    L.rt_last_error.argtypes = [C.c_void_p]; L.rt_last_error.restype = C.c_char_p
    L.rt_set_surfaces.argtypes = [C.c_void_p, P(C.c_double), C.c_int32, P(C.c_double), C.c_int32, P(C.c_uint32), C.c_int32]
    L.rt_set_rayleigh_lut.argtypes = [C.c_void_p, P(C.c_double), C.c_int32]
    L.rt_trace.argtypes = [C.c_void_p, P(C.c_double), P(C.c_double), C.c_int64, P(Params)]
    L.rt_trace_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, P(Params)]
    L.rt_trace_device_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, P(Params), C.c_void_p, C.c_void_p]
    L.rt_trace_point_source.argtypes = [C.c_void_p, P(C.c_double), C.c_int64, P(Params)]
    L.rt_generate_point_source.argtypes = [C.c_void_p, P(C.c_double), C.c_int64, C.c_uint64, C.c_int64, P(C.c_double), P(C.c_double)]
    L.rt_trace_camera.argtypes = [C.c_void_p, P(Camera), P(Params)]
    try:
        L.rt_trace_camera_range.argtypes = [C.c_void_p, P(Camera), C.c_int64, C.c_int64, P(Params)]
        L.rt_result_table_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    except AttributeError:
        if "RT_LIB" not in os.environ:   # (an older A/B build named by RT_LIB may lack the newest entry points; the product's own library may not)
            raise
    L.rt_generate_camera_rays.argtypes = [C.c_void_p, P(Camera), C.c_int64, C.c_int64, P(C.c_double), P(C.c_double), P(C.c_int32)]
    L.rt_set_record_sink.argtypes = [C.c_void_p, RECORD_SINK, C.c_void_p]
    L.rt_match_endpoints.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, P(C.c_int32), C.c_int32, P(MatchParams),
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_set_final_rows_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.rt_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.rt_result_num_steps.argtypes = [C.c_void_p, P(C.c_int32)]
    try:
        L.rt_result_pass_retries.argtypes = [C.c_void_p, P(C.c_int32)]
    except AttributeError:
        if not os.environ.get("RT_LIB"):
            raise
    L.rt_result_table.argtypes = [C.c_void_p, P(C.c_double)]
    L.rt_result_step_counts.argtypes = [C.c_void_p, P(C.c_int64), P(C.c_int64)]
    L.rt_result_num_records.argtypes = [C.c_void_p, P(C.c_int64)]
    L.rt_result_records.argtypes = [C.c_void_p, P(C.c_double), P(C.c_int32), P(C.c_int64), P(C.c_int64), P(C.c_int32)]
    L.rt_result_record_fields.argtypes = [C.c_void_p, P(C.c_int64), C.c_int64, C.c_int32, P(C.c_int32), P(C.c_int32), P(P(C.c_double))]
    L.rt_split_record_fields.argtypes = [P(C.c_double), C.c_int64, P(C.c_int64), C.c_int64, C.c_int32, P(C.c_int32), P(C.c_int32), P(P(C.c_double))]
    try:
        L.rt_normalize_rays.argtypes = [P(C.c_double), C.c_int64, P(C.c_double)]
        L.rt_order_records.argtypes = [P(C.c_int32), P(C.c_int64), P(C.c_int64), P(C.c_int32), C.c_int64, C.c_int32, P(C.c_int64), P(C.c_int64)]
    except AttributeError:
        if "RT_LIB" not in os.environ:
            raise
    L.rt_result_final.argtypes = [C.c_void_p, P(C.c_double)]
    L.rt_result_timing.argtypes = [C.c_void_p, P(C.c_double), P(C.c_int32)]
    L.rt_result_timing_by_kernel.argtypes = [C.c_void_p, P(C.c_double), P(C.c_int32)]
    L.rt_device_info.argtypes = [C.c_void_p, P(C.c_int32), P(C.c_int64), C.c_char_p, C.c_int32]
    _LIB = L
    return L
