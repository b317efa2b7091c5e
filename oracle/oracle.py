"""ctypes front end of the CPU oracle (oracle/rt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package raytracer_amd never does.  It deliberately does not import raytracer_amd either: surface objects are
turned into the flat table here, independently, and `inbounds_function` lambdas are *called* (through a C
callback, once per surface per step on the reference's N-by-3-by-M array) instead of being compiled, so
that the product's bounds compiler is itself checked against the lambdas.

Mirrors python/RayTracer2.py:185 (signature, defaults, return triple).
"""
import ctypes as C
import os
import subprocess
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ROW, ATOM, REC, FIN = 32, 12, 48, 32
SHAPES = {"plane": 0, "sphere": 1, "cylinder": 2, "quadsurface": 3, "torus": 4}
STYPES = {"normal": 0, "diffuse": 1, "unified": 2, "retro": 3}
COMPAT = {"python": 1, "python-strict": 3, "matlab": 0}


class Params(C.Structure):
    _fields_ = [("max_scatters", C.c_int32), ("flags", C.c_int32), ("compat", C.c_int32), ("unified_maxiter", C.c_int32),
                ("min_travel_length", C.c_double), ("follow_threshold", C.c_double * 2), ("tir_handling", C.c_double),
                ("seed", C.c_uint64), ("ray_id_base", C.c_int64), ("ray_id_stride", C.c_int64)]


CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int, C.POINTER(C.c_uint8))


def build():
    so = os.path.join(_HERE, "librt_oracle.so")
    src = os.path.join(_HERE, "rt_oracle.c")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ora_result_step_count.restype = C.c_int64
        _LIB.ora_path_hash.restype = C.c_uint32
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def geom_params(shape, param_list):
    """Flatten surface.param_list (python/RayToShape.py:18-29) to the 13-double block of include/rt_mi355x.h."""
    g = np.zeros(13)
    pl = param_list
    if shape == "plane":
        g[0:3] = np.asarray(pl[0], float).ravel(); g[3:6] = np.asarray(pl[1], float).ravel()
    elif shape == "sphere":
        g[0:3] = np.asarray(pl[0], float).ravel(); g[3] = float(pl[1])
    elif shape == "cylinder":
        g[0:3] = np.asarray(pl[0], float).ravel(); g[3:6] = np.asarray(pl[1], float).ravel(); g[6] = float(pl[2])
    elif shape == "quadsurface":
        g[0:9] = np.asarray(pl[0], float).reshape(9); g[9:12] = np.asarray(pl[1], float).ravel(); g[12] = float(pl[2])
    elif shape == "torus":
        g[0:3] = np.asarray(pl[0], float).ravel(); g[3:6] = np.asarray(pl[1], float).ravel(); g[6] = float(pl[2]); g[7] = float(pl[3])
    else:
        raise Exception('Geometry has unrecognized shape')
    return g


def build_table(surfacelist, nobulk=False):
    S = len(surfacelist)
    t = np.zeros((S, ROW))
    for i, s in enumerate(surfacelist):
        t[i, 0] = SHAPES[s.shape] if s.shape in SHAPES else -1
        if t[i, 0] < 0:
            raise Exception('Geometry has unrecognized shape')
        t[i, 1] = STYPES.get(s.surface_type, 0)
        t[i, 2:15] = geom_params(s.shape, s.param_list)
        t[i, 15] = s.n_outside; t[i, 16] = s.n_inside; t[i, 17] = s.absorption
        for j, name in enumerate(("abslength_outside", "abslength_inside", "rayleigh_outside", "rayleigh_inside")):
            v = getattr(s, name, None)
            t[i, 18 + j] = np.inf if (nobulk or v is None or np.size(v) == 0) else float(v)
        up = getattr(s, "unifiedparams", None)
        t[i, 22:27] = [0, 1, 0, 1, 0] if (up is None or np.size(up) == 0) else np.asarray(up, float)
    return t


def trace(ray_startingpoints, rays, surfacelist, max_scat=10, min_travel_len=np.spacing(1.0), follow_thresh=(0.0, 0.0),
          tir_handling=None, full_output=True, singlechild=True, output_raytable=False, *, compat="python", seed=0,
          ray_id_base=0, ray_id_stride=1, program=None, final=False, unified_maxiter=0):
    """Run the oracle.  `program=None`: call the surfaces' inbounds lambdas; otherwise (table, atoms, code) from a
    bounds compiler (used to time the oracle without Python in the loop, and to test the compiler)."""
    L = lib()
    sp = np.ascontiguousarray(ray_startingpoints, dtype=np.float64)
    ry = np.ascontiguousarray(rays, dtype=np.float64)
    if len(surfacelist) == 0 or sp.ndim != 2 or ry.ndim != 2 or sp.shape[1] != 3 or ry.shape[1] != 10 or sp.shape[0] != ry.shape[0]:
        raise Exception('Improper input to RayTracer2')
    n = sp.shape[0]
    ft = np.atleast_1d(np.asarray(follow_thresh, float))
    thr = (float(ft[0]), float(ft[0])) if ft.size == 1 else (float(ft[0]), float(ft[1]))
    K = int(np.atleast_1d(max_scat)[0])
    mtl = float(np.atleast_1d(min_travel_len)[0])
    cflags = COMPAT[compat]
    if tir_handling is None or np.size(tir_handling) == 0:
        ref = thr[0] if (cflags & 1) else thr[1]  # RayTracer2.py:252 vs RayTracer2.m:279
        tir_handling = 1.0 if ref > np.max(ry[:, 6]) else -1.0
    prm = Params()
    prm.max_scatters = K
    prm.flags = (1 if (full_output or output_raytable) else 0) | (2 if singlechild else 0) | (4 if final else 0)
    prm.compat = cflags & 3
    prm.unified_maxiter = unified_maxiter
    prm.min_travel_length = mtl
    prm.follow_threshold[0], prm.follow_threshold[1] = thr
    prm.tir_handling = float(tir_handling)
    prm.seed = seed
    prm.ray_id_base = ray_id_base
    prm.ray_id_stride = ray_id_stride

    if program is None:
        table = build_table(surfacelist, nobulk=(compat == "python-strict"))
        atoms = np.zeros(ATOM); code = np.zeros(1, np.uint32)

        def _cb(ctx, surf, p, nn, M, out):
            try:
                pa = np.ctypeslib.as_array(p, shape=(nn, 3, M))
                with np.errstate(all="ignore"):
                    r = np.asarray(surfacelist[surf].inbounds_function(pa))
                r = np.reshape(r, (nn, M)).astype(np.uint8)
                np.ctypeslib.as_array(out, shape=(nn, M))[...] = r
                return 0
            except Exception as e:  # noqa
                import traceback; traceback.print_exc()
                return 1
        cb = CB(_cb)
    else:
        table, atoms, code = [np.ascontiguousarray(x) for x in program]
        table = table.astype(np.float64); atoms = atoms.astype(np.float64); code = code.astype(np.uint32)
        cb = C.cast(None, CB)
    S = table.shape[0]
    res = C.c_void_p()
    rc = L.ora_trace(_dp(sp), _dp(ry), C.c_int64(n), _dp(table), C.c_int(S), _dp(atoms), code.ctypes.data_as(C.POINTER(C.c_uint32)),
                     C.byref(prm), cb, None, None, C.c_int(0), C.byref(res))
    if rc != 0:
        raise RuntimeError("ora_trace failed: %d" % rc)
    try:
        ns = L.ora_result_num_steps(res)
        table_out = np.zeros((ns, 5, S, 2))
        L.ora_result_table(res, _dp(table_out))
        live = np.zeros(K, np.int64); scat = np.zeros(K, np.int64)
        L.ora_result_counts(res, live.ctypes.data_as(C.POINTER(C.c_int64)), scat.ctypes.data_as(C.POINTER(C.c_int64)))
        ris = []
        if full_output or output_raytable:
            for k in range(ns):
                m = L.ora_result_step_count(res, k)
                if scat[k] == 0:
                    continue  # RayTracer2.py:405-407: nothing appended in a step where nothing scatters
                rec = np.zeros((m, REC))
                L.ora_result_step_copy(res, k, _dp(rec))
                ris.append(records_to_interface(rec))
        fin = None
        if final:
            fin = np.zeros((n, FIN))
            L.ora_result_final(res, _dp(fin))
    finally:
        L.ora_result_free(res)
    raytable = []
    if output_raytable:  # RayTracer2.m:249-253,543-548
        raytable = np.zeros((K + 1, n, 13))
        raytable[0, :, 0:3] = sp; raytable[0, :, 3:13] = ry
        with np.errstate(all="ignore"):  # RayTracer2.m:244-245: layer 1 holds the rays as normalised
            raytable[0, :, 3:6] /= np.abs(np.sqrt(np.sum(raytable[0, :, 3:6] ** 2, axis=1, keepdims=True)))
            raytable[0, :, 6:9] /= np.abs(np.sqrt(np.sum(raytable[0, :, 6:9] ** 2, axis=1, keepdims=True)))
        base = 0 if (cflags & 1) else 1
        for k, ri in enumerate(ris):
            cut = ri.has_reflected == 0
            ix = (np.abs(ri.ray_index[cut]).astype(np.int64) - base - ray_id_base) // max(int(ray_id_stride), 1)
            raytable[k + 1, ix, 0:3] = ri.intersection_point[cut]
            raytable[k + 1, ix, 3:13] = ri.refracted_ray[cut]
    out = types.SimpleNamespace(ray_interfaces=ris if full_output else [], absorption_table=table_out, raytable=raytable,
                                n_live=live, n_scat=scat, final=fin, num_steps=ns)
    return out


def records_to_interface(rec):
    """[n][48] record block -> object with the attributes of python/rayInterfaces.py:4-20 (+ extras)."""
    ri = types.SimpleNamespace()
    ri.incoming_ray = rec[:, 0:10].copy(); ri.refracted_ray = rec[:, 10:20].copy(); ri.reflected_ray = rec[:, 20:30].copy()
    ri.intersection_point = rec[:, 30:33].copy(); ri.surface_normal = rec[:, 33:36].copy()
    ri.ray_index = rec[:, 36].copy(); ri.surface_index = rec[:, 37].copy(); ri.distance_traveled = rec[:, 38].copy()
    ri.n_incident = rec[:, 39].copy(); ri.n_transmitted = rec[:, 40].copy()
    ri.bulkabs_incident = rec[:, 41].copy(); ri.bulkabs_transmitted = rec[:, 42].copy()
    ri.rayleigh_incident = rec[:, 43].copy(); ri.rayleigh_transmitted = rec[:, 44].copy()
    ri.orientation = rec[:, 45].copy(); ri.surface_number = rec[:, 46].copy(); ri.is_rayleigh = rec[:, 47].copy()
    ri.has_reflected = (rec[:, 36] < 0).astype(np.int8)
    return ri


# ---- unit entry points -----------------------------------------------------------------------------------
def ray_to_shape(shape, param_list, sp, d, py_root2=False):
    L = lib()
    sp = np.ascontiguousarray(sp, float); d = np.ascontiguousarray(d, float)
    n = sp.shape[0]; M = {"plane": 1, "torus": 4}.get(shape, 2)
    g = geom_params(shape, param_list)
    p = np.zeros((n, M, 3)); nr = np.zeros((n, M, 3)); l = np.zeros((n, M)); o = np.zeros((n, M)); ir = np.zeros((n, M), np.int32)
    L.ora_ray_to_shape(SHAPES[shape], _dp(g), _dp(sp), _dp(d), C.c_int64(n), int(py_root2), _dp(p), _dp(nr), _dp(l), _dp(o),
                       ir.ctypes.data_as(C.POINTER(C.c_int32)))
    return p, nr, l, o, ir


def fresnel(rays, normals, n1, n2, tir=-1.0):
    L = lib()
    rays = np.ascontiguousarray(rays, float); normals = np.ascontiguousarray(normals, float); n = rays.shape[0]
    n1 = np.ascontiguousarray(np.broadcast_to(np.asarray(n1, float), (n,))); n2 = np.ascontiguousarray(np.broadcast_to(np.asarray(n2, float), (n,)))
    a = np.zeros((n, 10)); b = np.zeros((n, 10))
    L.ora_fresnel(_dp(rays), _dp(normals), _dp(n1), _dp(n2), C.c_double(tir), C.c_int64(n), _dp(a), _dp(b))
    return a, b


def unified(rays, normals, n1, n2, params, seed=0, maxiter=10000):
    L = lib()
    rays = np.ascontiguousarray(rays, float); normals = np.ascontiguousarray(normals, float); n = rays.shape[0]
    n1 = np.ascontiguousarray(np.broadcast_to(np.asarray(n1, float), (n,))); n2 = np.ascontiguousarray(np.broadcast_to(np.asarray(n2, float), (n,)))
    prm = np.ascontiguousarray(params, float)
    out = np.zeros((n, 10))
    L.ora_unified(_dp(rays), _dp(normals), _dp(n1), _dp(n2), _dp(prm), C.c_int64(n), C.c_uint64(seed), C.c_int(maxiter), _dp(out))
    return out


def rayleigh(rays, seed=0):
    L = lib()
    rays = np.ascontiguousarray(rays, float); out = np.zeros_like(rays)
    L.ora_rayleigh(_dp(rays), C.c_int64(rays.shape[0]), C.c_uint64(seed), _dp(out))
    return out


def rayleigh_lut(prec=100000):
    L = lib(); out = np.zeros(prec + 1); L.ora_rayleigh_lut(C.c_int(prec), _dp(out)); return out


def uniforms(seed, ray_id, path_hash, step, n):
    L = lib(); out = np.zeros(n)
    L.ora_uniforms(C.c_uint64(seed), C.c_int64(ray_id), C.c_uint32(path_hash), C.c_int(step), C.c_int(n), _dp(out)); return out


def eval_bounds(code, off, length, atoms, p):
    L = lib(); p = np.ascontiguousarray(p, float); out = np.zeros(p.shape[0], np.uint8)
    code = np.ascontiguousarray(code, np.uint32); atoms = np.ascontiguousarray(atoms, float)
    L.ora_eval_bounds(code.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int(off), C.c_int(length), _dp(atoms), _dp(p), C.c_int64(p.shape[0]),
                      out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)


def poly_roots(coef):
    L = lib(); c = np.ascontiguousarray(coef, float); deg = c.size - 1
    re = np.zeros(4); ir = np.zeros(4, np.int32)
    L.ora_poly_roots(_dp(c), C.c_int(deg), _dp(re), ir.ctypes.data_as(C.POINTER(C.c_int32)))
    return re[:deg], ir[:deg].astype(bool)


def generate_rays_from_camera(resolution, pixel_pitch, pixel_center, focal_length, pitch=0.0, yaw=0.0, roll=0.0, radial_distortion=(), lens_type="tan"):
    """numpy restatement of MATLAB/Examples/ImagingExamples/GenerateRaysFromCamera.m:42-81, line by line (the numpy port
    python/GenerateRaysFromCamera.py does not run: SURVEY.md section 2).  Returns (ray_direction N x 3, pixels N x 2),
    rows in the column-major order of i_pix(:), j_pix(:).  Parity unpinned by any executable reference."""
    res = [int(resolution[0]), int(resolution[1])]
    pp = np.atleast_1d(np.asarray(pixel_pitch, float)).ravel()
    if pp.size == 1:
        pp = np.array([pp[0], pp[0]])                                        # :38-40
    f = float(np.atleast_1d(focal_length).ravel()[0])
    i_pix, j_pix = np.meshgrid(np.arange(1, res[0] + 1), np.arange(1, res[1] + 1), indexing="ij")   # :44-45
    i_pix = i_pix.ravel(order="F").astype(float); j_pix = j_pix.ravel(order="F").astype(float)   # (:)
    pixels = np.column_stack([i_pix, j_pix]).astype(np.int64)                # :46
    loc = np.column_stack([(pixel_center[0] - i_pix) * pp[0], np.zeros(i_pix.size) - f, -(pixel_center[1] - j_pix) * pp[1]])   # :48-50
    d2 = np.sum(loc[:, [0, 2]] ** 2, axis=1)                                 # :52
    rd = np.atleast_1d(np.asarray(radial_distortion, float)).ravel()
    eff = f * (1 + sum(rd[k] * (f ** -2 * d2) ** (k + 1) for k in range(rd.size)))   # :54-56
    with np.errstate(all="ignore"):
        if lens_type == "theta":
            theta = np.sqrt(d2) / eff                                        # :59-60
        elif lens_type == "sin":
            theta = np.arcsin(np.sqrt(d2) / eff)                             # :61-62
        else:
            theta = np.arctan(np.sqrt(d2) / eff)                             # :63-66
    phi = np.arctan2(-loc[:, 2], -loc[:, 0])                                 # :69
    rdir = np.column_stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)])   # :70
    M1 = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])         # :75
    M2 = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])  # :76
    M3 = np.array([[np.cos(roll), 0, np.sin(roll)], [0, 1, 0], [-np.sin(roll), 0, np.cos(roll)]])      # :77
    M = M1 @ M2 @ M3                                                         # :79
    return (M @ rdir.T).T, pixels                                            # :81


class FinalRays(object):
    """python/finalRays.py:2-9 plus the derived angles of BubbleImageCreator.py:163-165"""


def final_rays(raytracer_output, n_rays, index_base=0):
    """Restatement of python/BubbleImageCreator.py:134-165 (the loop as its MATLAB comments mean it: the numpy lines do not
    run - `range(np.size(x,0,-1))`, `np.zeros(n, 10)`, `not array`, two-output np.isin).  Walk the scatter steps from the
    last to the first; a ray id not found yet whose (positive) id is a member of that step's ray_index takes that step's
    record, at the lowest row holding it (ismember).  Ray ids start at index_base.  Rays never found keep zeros.
    cos_incident uses columns 1:3 of the incoming ray (the numpy `0:2` cannot broadcast)."""
    fr = FinalRays()
    fr.incoming_ray = np.zeros((n_rays, 10)); fr.intersection_point = np.zeros((n_rays, 3)); fr.surface_normal = np.zeros((n_rays, 3))
    fr.ray_index = np.zeros(n_rays); fr.surface_index = np.zeros(n_rays)
    found = np.zeros(n_rays, bool)
    ids = np.arange(n_rays) + index_base
    for n in range(len(raytracer_output) - 1, -1, -1):                       # :149
        if found.all():
            break                                                            # :150-151
        ri = np.asarray(raytracer_output[n].ray_index).ravel()
        first = {}
        for row, v in enumerate(ri):                                         # ismember: lowest row
            if v not in first:
                first[v] = row
        present = np.array([float(i) in first for i in ids], bool) if n_rays else np.zeros(0, bool)   # :153
        these = ~found & present                                             # :154
        found |= these                                                       # :156
        rows = np.array([first[float(i)] for i in ids[these]], dtype=int)
        fr.incoming_ray[these] = np.asarray(raytracer_output[n].incoming_ray)[rows]          # :157
        fr.intersection_point[these] = np.asarray(raytracer_output[n].intersection_point)[rows]   # :158
        fr.surface_normal[these] = np.asarray(raytracer_output[n].surface_normal)[rows]      # :159
        fr.ray_index[these] = ri[rows]                                       # :160
        fr.surface_index[these] = np.asarray(raytracer_output[n].surface_index).ravel()[rows]    # :161
    fr.found = found
    fr.cos_incident = -np.sum(fr.incoming_ray[:, 0:3] * fr.surface_normal, axis=1)   # :163
    with np.errstate(invalid="ignore"):
        fr.sin_incident = np.sqrt(1 - fr.cos_incident ** 2)                  # :164
    fr.theta_incident = np.arccos(np.clip(fr.cos_incident, -1.0, 1.0))       # :165 (clamped: see ora_match_endpoints)
    return fr


def match_endpoints(cam, led, surfaces, max_dist=5.0, angle_den=2 * 10 * np.pi / 180, cam_den=2 * 45 * np.pi / 180, led_den=2 * 45 * np.pi / 180, dist_den=2 * 2 ** 2):
    """ora_match_endpoints on two FinalRays: (pixel_intensity, led_cam, led_dist, led_add).  `surfaces` in the convention
    of surface_index (sign ignored).  Rays never found (zeros, surface_index 0) only take part if 0 is listed - callers
    mask them by setting surface_index to NaN first when the geometry is 0-based."""
    L = lib()
    n_cam, n_led = cam.intersection_point.shape[0], led.intersection_point.shape[0]
    arr = lambda a: np.ascontiguousarray(a, float)  # noqa: E731
    c_ip, c_ir, c_th, c_six = arr(cam.intersection_point), arr(cam.incoming_ray), arr(cam.theta_incident), arr(cam.surface_index)
    l_ip, l_ir, l_th, l_six = arr(led.intersection_point), arr(led.incoming_ray), arr(led.theta_incident), arr(led.surface_index)
    sf = np.ascontiguousarray(surfaces, np.int32)
    pix = np.zeros(n_cam); lc = np.zeros(n_led, np.int64); ld = np.zeros(n_led); la = np.zeros(n_led)
    L.ora_match_endpoints(_dp(c_ip), _dp(c_ir), _dp(c_th), _dp(c_six), C.c_int64(n_cam), _dp(l_ip), _dp(l_ir), _dp(l_th), _dp(l_six), C.c_int64(n_led),
                          sf.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(sf.size), C.c_double(max_dist), C.c_double(angle_den), C.c_double(cam_den),
                          C.c_double(led_den), C.c_double(dist_den), _dp(pix), lc.ctypes.data_as(C.POINTER(C.c_int64)), _dp(ld), _dp(la))
    return pix, lc, ld, la
