"""Drop-in for the reference's python/RayTracer2.py: same module name, same function name, same positional
signature, same return triple -- the propagation loop itself runs in HIP kernels on an MI355X
(raytracer_amd/csrc/rt_kernels.hip behind include/rt_mi355x.h).

    [ray_interfaces, absorption_table, raytable] = RayTracer2(ray_startingpoints, rays, surfacelist, max_scat,
            min_travel_len, follow_thresh, tir_handling, full_output, singlechild, output_raytable)

Differences a caller can see, all deliberate (SURVEY.md quirk ledger):
  * no stdout noise (Q13); scalar arguments are accepted wherever the reference insists on indexables (Q19);
  * paths that crash or silently misbehave in the numpy port follow the MATLAB text instead: second sphere/cylinder
    root (Q16), bulk absorption and Rayleigh scattering (Q3), diffuse/retro surfaces (Q7/Q8), torus and quadsurface
    (Q10), UNIFIED with sigma_alpha > 0 (Q11), tir_handling >= 0 (Q6), output_raytable (Q9).  `compat='python-strict'`
    switches Q3 and Q16 back on for bit-level comparisons with the numpy port;
  * index conventions stay those of the numpy port by default (`compat='python'`: 0-based `n*orientation`, surface 0
    tallied on both orientations, Q1/Q2/Q5); `compat='matlab'` gives the 1-based MATLAB ones.  Either way the
    records carry the extra attributes `orientation`, `surface_number`, `is_rayleigh`;
  * randomness is a counter-based Philox stream keyed by (seed, ray id, path, step): results do not depend on the
    number of GPUs or on chunking.  `seed=None` draws the seed from numpy's global stream, so `np.random.seed()`
    still makes runs repeatable.
Keyword-only extras: compat, seed, device, return_final (True: rows as a numpy array; "device": as a torch tensor in
HBM), ray_id_base, ray_id_stride (ray i has the id base + i * stride: a rank's interleaved share), device_table (the
absorption_table as a torch tensor [max_scat][5][S][2] in HBM, for a device-side all-reduce), ctx.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import table as _table
from .rayInterfaces import rayInterfaces

NUMPY_NORMALISE = False   # (tools/bench_dropin.py --numpy-host-work: the caller's rays normalised by the numpy expression, as in round 2)
COMPAT = {"python": _lib.COMPAT_PY_INDEX, "python-strict": _lib.COMPAT_PY_INDEX | _lib.COMPAT_PY_ROOT2, "matlab": 0}
_CTX = {}


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context(object):
    """One rt_ctx (one GPU, one stream) plus the geometry currently uploaded to it."""

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        if self.L.rt_init(int(device), C.byref(h)) != 0:
            raise _lib.RayTracerLibraryError(self.L.rt_last_error(None).decode())
        self.h = h
        self.device = device
        self._geom_key = None
        self._lut_set = False
        self.n_surf = 0
        self.options = {}

    def close(self):
        if self.h:
            self.L.rt_free(self.h); self.h = None

    def _check(self, rc):
        if rc != 0:
            msg = self.L.rt_last_error(self.h).decode()
            if msg in ("Improper input to RayTracer2", "Geometry has unrecognized shape"):
                raise Exception(msg)
            err = _lib.RayTracerLibraryError(msg); err.code = rc
            raise err

    def set_option(self, key, value):
        self._check(self.L.rt_set_option(self.h, key.encode(), float(value)))
        self.options[key] = float(value)   # what the caller asked for last (RayTracer2 restores "nodice_chunk" from here)

    def device_info(self):
        ncu = C.c_int32(); hbm = C.c_int64(); name = C.create_string_buffer(64)
        self._check(self.L.rt_device_info(self.h, C.byref(ncu), C.byref(hbm), name, 64))
        return {"n_cu": ncu.value, "hbm_bytes": hbm.value, "arch": name.value.decode()}

    def set_program(self, tab, atoms, code):
        tab = np.ascontiguousarray(tab, np.float64); atoms = np.ascontiguousarray(atoms, np.float64); code = np.ascontiguousarray(code, np.uint32)
        self._check(self.L.rt_set_surfaces(self.h, _dp(tab), tab.shape[0], _dp(atoms), atoms.shape[0], code.ctypes.data_as(C.POINTER(C.c_uint32)), code.shape[0]))
        self.n_surf = tab.shape[0]
        self.table = tab
        if np.any(np.isfinite(tab[:, 20:22])) and not self._lut_set:
            lut = np.ascontiguousarray(_table.rayleigh_lut())
            self._check(self.L.rt_set_rayleigh_lut(self.h, _dp(lut), lut.shape[0])); self._lut_set = True

    def set_surfaces(self, surfacelist, nobulk=False):
        """upload the geometry unless it is the one already there.  The key holds the surfaces' own values and, for every
        inbounds_function, its code and whatever it closes over or reads from its globals (a lambda whose closed-over bound
        changes keeps its id); the surfaces and functions are kept referenced, so that no id in the key can be recycled."""
        key = (tuple(id(s) for s in surfacelist), nobulk, tuple(_surface_fingerprint(s) for s in surfacelist))
        if key != self._geom_key:
            self.set_program(*_table.compile_surfaces(surfacelist, nobulk=nobulk))
            self._geom_key = key
            self._geom_refs = (list(surfacelist), [getattr(s, "inbounds_function", None) for s in surfacelist])

    def params(self, K, flags, compat, min_travel, thr, tir, seed, ray_id_base, unified_maxiter=0, ray_id_stride=1):
        p = _lib.Params()
        p.ray_id_stride = int(ray_id_stride)
        p.max_scatters = int(K); p.flags = int(flags); p.compat = int(compat); p.unified_maxiter = int(unified_maxiter)
        p.min_travel_length = float(min_travel); p.follow_threshold[0] = float(thr[0]); p.follow_threshold[1] = float(thr[1])
        p.tir_handling = float(tir); p.seed = int(seed) & 0xFFFFFFFFFFFFFFFF; p.ray_id_base = int(ray_id_base)
        return p

    # ---- results
    def results(self, want_records, want_final, n_rays):
        ns = C.c_int32(); self._check(self.L.rt_result_num_steps(self.h, C.byref(ns))); ns = ns.value
        tab = np.zeros((ns, 5, self.n_surf, 2))
        if ns:
            self._check(self.L.rt_result_table(self.h, _dp(tab)))
        out = {"num_steps": ns, "table": tab}
        if want_records:
            n = C.c_int64(); self._check(self.L.rt_result_num_records(self.h, C.byref(n))); n = n.value
            meta_only = isinstance(want_records, str) and want_records == "meta"  # the rows stay in the library: interfaces() splits them into fields
            rec = None if meta_only else np.empty((n, _lib.REC_COLS))
            step = np.empty(n, np.int32); ray = np.empty(n, np.int64); par = np.empty(n, np.int64); fl = np.empty(n, np.int32)
            if n:
                self._check(self.L.rt_result_records(self.h, None if meta_only else _dp(rec), step.ctypes.data_as(C.POINTER(C.c_int32)), ray.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     par.ctypes.data_as(C.POINTER(C.c_int64)), fl.ctypes.data_as(C.POINTER(C.c_int32))))
            out.update(rec=rec, step=step, ray=ray, parent=par, flags=fl)
        if want_final:
            fin = np.zeros((n_rays, _lib.FINAL_COLS)); self._check(self.L.rt_result_final(self.h, _dp(fin))); out["final"] = fin
        ms = C.c_double(); nl = C.c_int32(); self._check(self.L.rt_result_timing(self.h, C.byref(ms), C.byref(nl)))
        out["kernel_ms"] = ms.value; out["n_launches"] = nl.value
        return out

    def interfaces(self, groups, rows=None, n_rows=0):
        """one rayInterfaces (python/rayInterfaces.py:4-20) per non-empty index array of `groups` (order_records): the
        library splits the records of the last trace - or the row table `rows` (a ctypes double pointer, n_rows rows) a
        record sink was handed - into the fields, rows in the order given"""
        nf = len(_FIELDS)
        col0 = (C.c_int32 * nf)(*[f[1] for f in _FIELDS]); ncols = (C.c_int32 * nf)(*[f[2] for f in _FIELDS])
        out = []
        for g in groups:
            if not g.size:
                continue
            g = np.ascontiguousarray(g, np.int64)
            ri = rayInterfaces()
            arrs = [np.empty((g.size, f[2]) if f[2] > 1 else (g.size,)) for f in _FIELDS]
            ptrs = (C.POINTER(C.c_double) * nf)(*[_dp(a) for a in arrs])
            gp = g.ctypes.data_as(C.POINTER(C.c_int64))
            if rows is None:
                self._check(self.L.rt_result_record_fields(self.h, gp, g.size, nf, col0, ncols, ptrs))
            elif self.L.rt_split_record_fields(rows, int(n_rows), gp, g.size, nf, col0, ncols, ptrs) != 0:
                raise _lib.RayTracerLibraryError("rt_split_record_fields: bad request")
            for f, a in zip(_FIELDS, arrs):
                setattr(ri, f[0], a)
            out.append(ri)
        return out

    def final_rows_device(self, n_rays):
        """a torch tensor [n_rays][32] on this context's GPU that the next traces with final rows write into (see
        rt_set_final_rows_device); release with final_rows_host()"""
        import torch
        t = torch.zeros((int(n_rays), _lib.FINAL_COLS), dtype=torch.float64, device="cuda:%d" % int(self.device))
        torch.cuda.synchronize(t.device)
        self._check(self.L.rt_set_final_rows_device(self.h, C.c_void_p(t.data_ptr()), int(n_rays)))
        return t

    def final_rows_host(self):
        self._check(self.L.rt_set_final_rows_device(self.h, None, 0))

    def table_device(self, K):
        """the absorption_table of the last trace as a torch tensor [K][5][n_surf][2] on this context's GPU (all K rows; rows
        of steps in which nothing scattered are zero) - what the all-reduce of a multi-GPU run takes, without a trip
        through host memory (rt_result_table_device)"""
        import torch
        t = torch.empty((int(K), 5, self.n_surf, 2), dtype=torch.float64, device="cuda:%d" % int(self.device))
        self._check(self.L.rt_result_table_device(self.h, C.c_void_p(t.data_ptr()), None))
        ns = C.c_int32(); self._check(self.L.rt_result_num_steps(self.h, C.byref(ns)))
        t.num_steps = ns.value   # the reference's num_scatters: the rows behind it are zero
        return t

    def step_counts(self, K):
        live = np.zeros(K, np.int64); scat = np.zeros(K, np.int64)
        self._check(self.L.rt_result_step_counts(self.h, live.ctypes.data_as(C.POINTER(C.c_int64)), scat.ctypes.data_as(C.POINTER(C.c_int64))))
        return live, scat


def _value_print(v, depth=0):
    """hashable digest of a value a bounds lambda may depend on"""
    if isinstance(v, (int, float, complex, bool, str, bytes, type(None))):
        return v
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, np.ndarray):
        return ("nd", v.shape, v.dtype.str, v.tobytes()) if v.size <= 4096 else ("nd-big", id(v), v.shape)
    if isinstance(v, (list, tuple)) and depth < 3:
        return tuple(_value_print(x, depth + 1) for x in v)
    if callable(v) and depth < 3:
        return _function_print(v, depth + 1)
    return ("obj", id(v))


def _function_print(fn, depth=0):
    code = getattr(fn, "__code__", None)
    if code is None:
        return ("callable", id(fn))
    cells = tuple(_value_print(c.cell_contents, depth) if _cell_set(c) else None for c in (fn.__closure__ or ()))
    g = getattr(fn, "__globals__", {})
    names = tuple((n, _value_print(g[n], depth)) for n in code.co_names if n in g and not isinstance(g[n], type(np)))
    consts = tuple(_value_print(c, depth) if not hasattr(c, "co_code") else c.co_code for c in code.co_consts)
    return (code.co_code, consts, cells, names, _value_print(getattr(fn, "__defaults__", None), depth))


def _cell_set(c):
    try:
        c.cell_contents
        return True
    except ValueError:
        return False


def _surface_fingerprint(s):
    pl = getattr(s, "param_list", None)
    try:
        plf = tuple(float(x) for p in pl for x in np.asarray(p, float).ravel())
    except Exception:
        plf = None
    up = getattr(s, "unifiedparams", None)
    return (getattr(s, "shape", None), plf, getattr(s, "n_outside", None), getattr(s, "n_inside", None), getattr(s, "surface_type", None),
            getattr(s, "absorption", None), None if up is None else tuple(np.asarray(up, float).ravel()),
            tuple(None if getattr(s, n, None) is None else float(np.asarray(getattr(s, n)).ravel()[0]) if np.size(getattr(s, n)) else None for n in _table.BULK_PROPS),
            _function_print(getattr(s, "inbounds_function", None)) if getattr(s, "inbounds_function", None) is not None else None)


def trace_camera(cam, surfacelist, max_scat=10, min_travel_len=np.spacing(np.float64(1)), follow_thresh=np.array([0, 0], dtype=np.float64),
                 tir_handling=[], full_output=False, singlechild=True, *, compat="python", seed=None, device=0, ctx=None, ray_id_base=0,
                 unified_maxiter=0, device_rows=False, first_pixel=0, n_pixels=None, ray_id_stride=1):
    """The camera run of python/BubbleImageCreator.py:129 without host arrays: rays are generated in HBM from the
    rt_camera `cam` (GenerateRaysFromCamera.make_camera), traced, and reduced to the final row per pixel
    (BubbleImageCreator.py:149-161).  Returns (absorption_table, final_rows[n_pixels][32], ray_interfaces or []).
    first_pixel / n_pixels / ray_id_stride: a share of the sensor - pixels first, first + stride, ... (ray i has the id
    ray_id_base + i * stride) - for runs sharded over GPUs (multigpu.sharded_trace_camera)."""
    if np.size(surfacelist) == 0:
        raise Exception('Improper input to RayTracer2')
    npix = int(cam.resolution[0]) * int(cam.resolution[1])
    whole = n_pixels is None and first_pixel == 0 and ray_id_stride == 1
    n = npix if n_pixels is None else int(n_pixels)
    st = np.array([[cam.stokes[0]]])
    K, mtl, thr, tir = parse_args(np.hstack([np.zeros((1, 6)), st, np.zeros((1, 3))]), max_scat, min_travel_len, follow_thresh, tir_handling, compat)
    for s in surfacelist:
        if not hasattr(s, "unifiedparams") or np.size(s.unifiedparams) == 0:
            s.unifiedparams = [0, 1, 0, 1, 0]
        for name in _table.BULK_PROPS:
            if not hasattr(s, name) or np.size(getattr(s, name)) == 0:
                setattr(s, name, np.inf)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(np.random.randint(0, 2 ** 31 - 1))
    c = ctx if ctx is not None else get_context(device)
    c.set_surfaces(surfacelist, nobulk=(compat == "python-strict"))
    flags = (_lib.FLAG_FULL_OUTPUT if full_output else 0) | (_lib.FLAG_SINGLECHILD if singlechild else 0) | _lib.FLAG_FINAL_ROWS
    prm = c.params(K, flags, COMPAT[compat], mtl, thr, tir, seed, ray_id_base, unified_maxiter, ray_id_stride)
    rows = c.final_rows_device(n) if device_rows else None
    try:
        if whole:
            c._check(c.L.rt_trace_camera(c.h, C.byref(cam), C.byref(prm)))
        else:
            c._check(c.L.rt_trace_camera_range(c.h, C.byref(cam), int(first_pixel), n, C.byref(prm)))
    finally:
        if device_rows:
            c.final_rows_host()
    res = c.results("meta" if full_output else False, not device_rows, n)
    if device_rows:
        res["final"] = rows
    ris = []
    if full_output:
        groups = order_groups(res["step"], res["ray"], res["parent"], res["flags"], res["num_steps"])
        ris = c.interfaces(groups)
    return res["table"], res["final"], ris


def trace_point_source(origin, n, surfacelist, max_scat=10, min_travel_len=np.spacing(np.float64(1)), follow_thresh=np.array([0, 0], dtype=np.float64),
                       tir_handling=[], singlechild=True, *, compat="python", seed=None, device=0, ctx=None, ray_id_base=0, ray_id_stride=1,
                       unified_maxiter=0, device_table=False):
    """The light-collection run of MATLAB/Examples/RunExamples/RunSBCGeometry.m:45-77 without host arrays: n isotropic
    unpolarised unit-intensity rays from `origin` are generated in HBM (ray i has the id ray_id_base + i * ray_id_stride and
    is the same ray whichever call, GPU or chunk makes it) and traced for their tallies.  Returns the absorption_table, or
    with device_table=True a torch tensor [max_scat][5][S][2] in HBM (Context.table_device)."""
    if np.size(surfacelist) == 0 or int(n) < 1:
        raise Exception('Improper input to RayTracer2')
    K, mtl, thr, tir = parse_args(np.array([[0, 0, 0, 0, 0, 0, 1.0, 0, 0, 0]]), max_scat, min_travel_len, follow_thresh, tir_handling, compat)
    for s in surfacelist:
        if not hasattr(s, "unifiedparams") or np.size(s.unifiedparams) == 0:
            s.unifiedparams = [0, 1, 0, 1, 0]
        for name in _table.BULK_PROPS:
            if not hasattr(s, name) or np.size(getattr(s, name)) == 0:
                setattr(s, name, np.inf)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(np.random.randint(0, 2 ** 31 - 1))
    c = ctx if ctx is not None else get_context(device)
    c.set_surfaces(surfacelist, nobulk=(compat == "python-strict"))
    prm = c.params(K, _lib.FLAG_SINGLECHILD if singlechild else 0, COMPAT[compat], mtl, thr, tir, seed, ray_id_base, unified_maxiter, ray_id_stride)
    org = (C.c_double * 3)(*[float(x) for x in origin])
    c._check(c.L.rt_trace_point_source(c.h, org, int(n), C.byref(prm)))
    return c.table_device(K) if device_table else c.results(False, False, int(n))["table"]


class npz_shards(object):
    """record_sink that writes one .npz per chunk of rays into `directory`: for every step k of the chunk the arrays of
    python/rayInterfaces.py:4-20 as `k<step>_<field>`, in the reference's row order within the chunk, plus `first_ray_id`
    and `n_rays`.  `paths` lists the files written."""

    def __init__(self, directory, prefix="ray_interfaces"):
        import os
        self.directory = directory; self.prefix = prefix; self.paths = []
        os.makedirs(directory, exist_ok=True)

    def __call__(self, chunk, first_ray_id, n_rays, ray_interfaces):
        import os
        out = {"first_ray_id": np.int64(first_ray_id), "n_rays": np.int64(n_rays), "n_steps": np.int64(len(ray_interfaces))}
        for k, ri in enumerate(ray_interfaces):
            for f, v in vars(ri).items():
                out["k%d_%s" % (k, f)] = v
        path = os.path.join(self.directory, "%s_%05d.npz" % (self.prefix, chunk))
        np.savez(path, **out); self.paths.append(path)


def _install_sink(c, record_sink, K):
    """C callback around a Python sink(chunk_index, first_ray_id, n_rays, [rayInterfaces per step])"""
    state = {"chunk": 0, "error": None}

    def cb(user, n, rec, step, ray, parent, flags, first, nrays):
        try:
            n = int(n)
            st = np.ctypeslib.as_array(step, shape=(n,))
            ry = np.ctypeslib.as_array(ray, shape=(n,)); pa = np.ctypeslib.as_array(parent, shape=(n,)); fl = np.ctypeslib.as_array(flags, shape=(n,))
            groups = order_groups(st, ry, pa, fl, K)
            record_sink(state["chunk"], int(first), int(nrays), c.interfaces(groups, rec, n))
            state["chunk"] += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            state["error"] = e
            return 1
    fn = _lib.RECORD_SINK(cb)
    c._check(c.L.rt_set_record_sink(c.h, fn, None))
    return fn, state


def get_context(device=0):
    if device not in _CTX:
        _CTX[device] = Context(device)
    return _CTX[device]


def order_groups(step, ray, parent, flags, num_steps):
    """order_records() by the library (rt_order_records: tables instead of sorts, on several host threads); the numpy
    version below where the library declines (metadata that is not the reference's tree)"""
    L = _lib.load()
    n = int(step.shape[0])
    if n and num_steps and hasattr(L, "rt_order_records"):
        st = np.ascontiguousarray(step, np.int32); ry = np.ascontiguousarray(ray, np.int64); pa = np.ascontiguousarray(parent, np.int64); fl = np.ascontiguousarray(flags, np.int32)
        order = np.empty(n, np.int64); counts = np.zeros(int(num_steps), np.int64)
        I32, I64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        if L.rt_order_records(st.ctypes.data_as(I32), ry.ctypes.data_as(I64), pa.ctypes.data_as(I64), fl.ctypes.data_as(I32), n, int(num_steps),
                              order.ctypes.data_as(I64), counts.ctypes.data_as(I64)) == 0:
            ends = np.cumsum(counts)
            return [order[e - c:e] for c, e in zip(counts, ends)]
    return order_records(step, ray, parent, flags, num_steps)


def order_records(step, ray, parent, flags, num_steps):
    """Row order of the reference for every step: step 0 in input order; step k+1 = [refracted children in the order
    of their parents ; reflected children in the order of their parents] (python/RayTracer2.py:531-540)."""
    n = step.shape[0]
    pos = np.zeros(n, np.int64)
    out = []
    for k in range(num_steps):
        idx = np.nonzero(step == k)[0]
        if idx.size == 0:
            out.append(idx); continue
        if k == 0:
            o = np.argsort(ray[idx], kind="stable")
        else:
            born = (flags[idx] & 4) != 0
            o = np.lexsort((ray[idx], pos[parent[idx]], born))
        idx = idx[o]
        pos[idx] = np.arange(idx.size)
        out.append(idx)
    return out


# fields of python/rayInterfaces.py:4-20 as (name, first record column, columns): the layout of RT_REC_COLS in include/rt_mi355x.h
_FIELDS = (("incoming_ray", 0, 10), ("refracted_ray", 10, 10), ("reflected_ray", 20, 10), ("intersection_point", 30, 3), ("surface_normal", 33, 3),
           ("ray_index", 36, 1), ("surface_index", 37, 1), ("distance_traveled", 38, 1), ("n_incident", 39, 1), ("n_transmitted", 40, 1),
           ("bulkabs_incident", 41, 1), ("bulkabs_transmitted", 42, 1), ("rayleigh_incident", 43, 1), ("rayleigh_transmitted", 44, 1),
           ("orientation", 45, 1), ("surface_number", 46, 1), ("is_rayleigh", 47, 1))


def parse_args(rays, max_scat, min_travel_len, follow_thresh, tir_handling, compat, max_s0=None):
    """argument handling of python/RayTracer2.py:193-212, 250-255 (scalars accepted everywhere); max_s0: max(rays[:, 6]) if the caller has it"""
    ft = np.atleast_1d(np.asarray(follow_thresh, dtype=np.float64)).ravel()
    thr = (float(ft[0]), float(ft[0])) if ft.size == 1 else (float(ft[0]), float(ft[1]))
    K = int(np.atleast_1d(max_scat).ravel()[0])
    mtl = float(np.atleast_1d(min_travel_len).ravel()[0])
    if tir_handling is None or np.size(tir_handling) == 0:
        ref = thr[0] if (COMPAT[compat] & _lib.COMPAT_PY_INDEX) else thr[1]  # RayTracer2.py:252 vs RayTracer2.m:279
        tir = 1.0 if ref > (np.max(rays[:, 6]) if max_s0 is None else max_s0) else -1.0
    else:
        tir = float(np.atleast_1d(tir_handling).ravel()[0])
    return K, mtl, thr, tir


def RayTracer2(ray_startingpoints, rays, surfacelist=[], max_scat=10, min_travel_len=np.spacing(np.float64(1)),
               follow_thresh=np.array([0, 0], dtype=np.float64), tir_handling=[], full_output=True, singlechild=True,
               output_raytable=False, *, compat="python", seed=None, device=0, return_final=False, ray_id_base=0, ctx=None,
               unified_maxiter=0, record_sink=None, ray_id_stride=1, device_table=False):
    # error check inputs (python/RayTracer2.py:216-217), before any device work
    if np.size(surfacelist) == 0 or np.ndim(ray_startingpoints) != 2 or np.ndim(rays) != 2 or \
            np.shape(ray_startingpoints)[1] != 3 or np.shape(rays)[1] != 10 or np.shape(ray_startingpoints)[0] != np.shape(rays)[0]:
        raise Exception('Improper input to RayTracer2')
    if compat not in COMPAT:
        raise ValueError("compat must be one of %r" % (sorted(COMPAT),))
    sp = np.ascontiguousarray(ray_startingpoints, dtype=np.float64)
    ry = np.ascontiguousarray(rays, dtype=np.float64)
    n = ry.shape[0]
    if n == 0:
        raise Exception('Improper input to RayTracer2')
    # the reference normalises the caller's array in place (python/RayTracer2.py:221-222); keep that visible effect
    max_s0 = None
    if isinstance(rays, np.ndarray) and rays.dtype == np.float64:
        L_ = _lib.load()
        if rays.flags.c_contiguous and rays.flags.writeable and hasattr(L_, "rt_normalize_rays") and not NUMPY_NORMALISE:
            m_ = C.c_double()   # the same operations on several host threads (the numpy expression takes longer than the trace), and max(rays[:, 6]) on the way
            if L_.rt_normalize_rays(_dp(rays), n, C.byref(m_)) == 0:
                max_s0 = m_.value
        else:
            with np.errstate(all="ignore"):
                rays[:, 0:3] = rays[:, 0:3] / np.abs(np.sqrt(np.sum(rays[:, 0:3] ** 2, axis=1, keepdims=True)))
                rays[:, 3:6] = rays[:, 3:6] / np.abs(np.sqrt(np.sum(rays[:, 3:6] ** 2, axis=1, keepdims=True)))
    K, mtl, thr, tir = parse_args(ry, max_scat, min_travel_len, follow_thresh, tir_handling, compat, max_s0)
    # missing optional attributes get the reference's defaults on the caller's objects (RayTracer2.py:234-248)
    for s in surfacelist:
        if not hasattr(s, "unifiedparams") or np.size(s.unifiedparams) == 0:
            s.unifiedparams = [0, 1, 0, 1, 0]
        for name in _table.BULK_PROPS:
            if not hasattr(s, name) or np.size(getattr(s, name)) == 0:
                setattr(s, name, np.inf)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(np.random.randint(0, 2 ** 31 - 1))
    c = ctx if ctx is not None else get_context(device)
    c.set_surfaces(surfacelist, nobulk=(compat == "python-strict"))
    want_rec = bool(full_output or output_raytable)
    flags = (_lib.FLAG_FULL_OUTPUT if want_rec else 0) | (_lib.FLAG_SINGLECHILD if singlechild else 0) | (_lib.FLAG_FINAL_ROWS if return_final else 0)
    prm = c.params(K, flags, COMPAT[compat], mtl, thr, tir, seed, ray_id_base, unified_maxiter, ray_id_stride)
    chunk = None
    chunk_before = c.options.get("nodice_chunk")   # the caller's own setting, if any (the context is shared and cached)
    chunk_changed = False
    sink = None
    dev_rows = c.final_rows_device(n) if (isinstance(return_final, str) and return_final == "device") else None
    if record_sink is not None:  # streaming: records leave chunk by chunk; ray_interfaces comes back empty
        if not full_output or output_raytable or not singlechild:
            raise ValueError("record_sink needs full_output=True, singlechild=True and output_raytable=False")
        sink = _install_sink(c, record_sink, K)
    try:
        for attempt in range(8):
            try:
                c._check(c.L.rt_trace(c.h, _dp(sp), _dp(ry), n, C.byref(prm)))
                break
            except _lib.RayTracerLibraryError as e:
                if sink is not None and sink[1]["error"] is not None:
                    raise sink[1]["error"]
                # ray-splitting mode cannot bound its queue/record growth in advance: on overflow retry with smaller passes
                if getattr(e, "code", 0) != -3 or singlechild or attempt == 7:
                    raise
                chunk = max(256, int(chunk if chunk is not None else min(n, chunk_before or (1 << 20))) // 4)
                c.set_option("nodice_chunk", chunk); chunk_changed = True
    finally:
        if chunk_changed:   # the smaller passes were for this call only: back to what the caller had set (or the library's default)
            c.set_option("nodice_chunk", chunk_before if chunk_before is not None else (1 << 20))
            if chunk_before is None:
                del c.options["nodice_chunk"]
        if sink is not None:
            c.L.rt_set_record_sink(c.h, C.cast(None, _lib.RECORD_SINK), None)
        if dev_rows is not None:
            c.final_rows_host()
    res = c.results("meta" if want_rec else False, bool(return_final) and dev_rows is None, n)
    if dev_rows is not None:
        res["final"] = dev_rows
    ns = res["num_steps"]
    base = 0 if (COMPAT[compat] & _lib.COMPAT_PY_INDEX) else 1

    ray_interfaces = []
    raytable = []
    if want_rec:
        groups = order_groups(res["step"], res["ray"], res["parent"], res["flags"], ns)
        ris = c.interfaces(groups)
        if output_raytable:  # MATLAB/RayTracer2.m:249-253, 543-548 (numpy port broken)
            raytable = np.zeros((K + 1, n, 13))
            raytable[0, :, 0:3] = sp; raytable[0, :, 3:13] = ry
            raytable[0, :, 3:6] /= np.linalg.norm(raytable[0, :, 3:6], axis=1, keepdims=True)
            raytable[0, :, 6:9] /= np.linalg.norm(raytable[0, :, 6:9], axis=1, keepdims=True)
            for g, ri in zip([g for g in groups if g.size], ris):
                k = int(res["step"][g[0]])
                cut = (res["flags"][g] & 1) == 0  # never reflected <=> positive ray_index
                ix = (res["ray"][g][cut] - ray_id_base) // max(int(ray_id_stride), 1)
                raytable[k + 1, ix, 0:3] = ri.intersection_point[cut]
                raytable[k + 1, ix, 3:13] = ri.refracted_ray[cut]
        if full_output:
            ray_interfaces = ris
    out = [np.array(ray_interfaces, dtype=object), c.table_device(K) if device_table else res["table"], raytable]
    if return_final:
        out.append(res["final"])
    return out
