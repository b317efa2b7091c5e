"""ctypes wrapper of tests/hostsim (host build of the device header) -- test aid, see hostsim.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle import oracle as O
from raytracer_amd import table as T

HERE = os.path.dirname(os.path.abspath(__file__))
_L = None


def lib():
    global _L
    if _L is None:
        so = os.path.join(HERE, "hostsim", "libhostsim.so")
        src = os.path.join(HERE, "hostsim", "hostsim.cpp")
        hdr = os.path.join(HERE, "..", "raytracer_amd", "csrc", "rt_device.hpp")
        if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
            subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-unknown-pragmas", "-o", so, src])
        _L = C.CDLL(so)
        _L.hs_step_count.restype = C.c_int64
    return _L


def trace(sp, rays, surfacelist, K=10, min_travel=np.spacing(1.0), thr=(0.0, 0.0), tir=-1.0, full_output=True, singlechild=True,
          compat="python", seed=0, ray_id_base=0, final=False):
    L = lib()
    sp = np.ascontiguousarray(sp, float); rays = np.ascontiguousarray(rays, float); n = sp.shape[0]
    table, atoms, code = T.compile_surfaces(surfacelist, nobulk=(compat == "python-strict"))
    lut = T.rayleigh_lut()
    prm = O.Params()
    prm.max_scatters = K; prm.flags = (1 if full_output else 0) | (2 if singlechild else 0) | (4 if final else 0)
    prm.compat = O.COMPAT[compat] & 3; prm.min_travel_length = min_travel
    prm.follow_threshold[0], prm.follow_threshold[1] = thr
    prm.tir_handling = tir; prm.seed = seed; prm.ray_id_base = ray_id_base
    res = C.c_void_p()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L.hs_trace(dp(sp), dp(rays), C.c_int64(n), dp(table), C.c_int(table.shape[0]), dp(atoms), code.ctypes.data_as(C.POINTER(C.c_uint32)),
               dp(lut), C.c_int(len(lut)), C.byref(prm), C.byref(res))
    ns = L.hs_num_steps(res); S = table.shape[0]
    tab = np.zeros((ns, 5, S, 2)); L.hs_table(res, dp(tab))
    live = np.zeros(K, np.int64); scat = np.zeros(K, np.int64)
    L.hs_counts(res, live.ctypes.data_as(C.POINTER(C.c_int64)), scat.ctypes.data_as(C.POINTER(C.c_int64)))
    ris = []
    for k in range(ns):
        m = L.hs_step_count(res, k)
        if scat[k] == 0 or not full_output:
            continue
        rec = np.zeros((m, 48)); L.hs_step_copy(res, k, dp(rec)); ris.append(O.records_to_interface(rec))
    fin = None
    if final:
        fin = np.zeros((n, 32)); L.hs_final(res, dp(fin))
    L.hs_free(res)
    import types
    return types.SimpleNamespace(ray_interfaces=ris, absorption_table=tab, n_live=live, n_scat=scat, final=fin, num_steps=ns)
