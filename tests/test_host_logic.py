"""CPU tests of the host side: bounds compiler, surface table, argument handling, C-ABI export list, and the device
header compiled for the host (tests/hostsim) against the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

import cases
import hostsim_util as H
from compare_util import compare_results
from oracle import oracle as O
from raytracer_amd import _lib, bounds as B, table as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_fn(fn, scale=20.0, n=20000, seed=0):
    atoms, code = B.compile_bounds(fn, scale=scale)
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-scale, scale, (n, 3))
    want = np.reshape(fn(np.repeat(pts[:, :, None], 2, axis=2)), (n, -1))[:, 0]
    got = B.eval_program(atoms, code, pts)
    assert np.array_equal(want, got)
    arr = np.zeros((max(len(atoms), 1), 12))
    for i, a in enumerate(atoms):
        arr[i] = a
    assert np.array_equal(O.eval_bounds(np.array(code, np.uint32), 0, len(code), arr, pts), want)  # oracle's evaluator agrees
    return atoms, code


def test_bounds_reference_style_lambdas():
    # python/TestGeometry.py:48,61,74 ; python/UnifiedTest.py:19
    a, c = _check_fn(lambda p: np.reshape((p[:, 2, :] > 0) * (p[:, 2, :] < 5), (p.shape[0], -1)))
    assert [int(x[0]) for x in a] == [B.K_COORD, B.K_COORD]
    _check_fn(lambda p: np.reshape((p[:, 2, :] >= 5) * (p[:, 2, :] < 10), (p.shape[0], -1)))
    _check_fn(lambda p: np.reshape((p[:, 0] ** 2 + p[:, 1] ** 2) < 100, (p.shape[0], -1)))
    _check_fn(lambda p: np.reshape((p[:, 2, :] >= 0) * (p[:, 2, :] <= 10), (p.shape[0], -1)))


def test_bounds_matlab_style_predicates():
    # CreateArBCGeometry.m:328-330 (axial band), :410 (disk about a point), :542-543 (distance from an axis), :298-316 (atan2 bands)
    c3 = np.array([0, -10., 60.])[None, :, None]; a3 = np.array([0, -0.1045, 0.9945])[None, :, None]
    ax = lambda p: np.sum((p - c3) * a3, axis=1)  # noqa: E731
    _check_fn(lambda p: np.reshape((ax(p) > 0) & (ax(p) <= 1.77), (p.shape[0], -1)), scale=80)
    _check_fn(lambda p: np.reshape(np.sum((p - c3) ** 2, axis=1) <= 4.0 ** 2, (p.shape[0], -1)), scale=80)
    _check_fn(lambda p: np.reshape((p[:, 2, :] > 20) & ((np.sum((p - c3) ** 2, axis=1) - ax(p) ** 2) > 2.2 ** 2), (p.shape[0], -1)), scale=80)
    _check_fn(lambda p: np.reshape((np.arctan2(p[:, 1, :], p[:, 0, :]) > 0.3) | ~(p[:, 2, :] < 1), (p.shape[0], -1)))
    _check_fn(lambda p: np.reshape(np.logical_and(np.sqrt(p[:, 0, :] ** 2 + p[:, 1, :] ** 2) < 7, np.abs(p[:, 2, :]) <= 3), (p.shape[0], -1)))
    _check_fn(lambda p: np.reshape(np.logical_or(np.abs(p[:, 2, :] - 1) > 3, p[:, 0, :] * p[:, 1, :] >= 2), (p.shape[0], -1)))


def test_bounds_refuses_what_it_cannot_compile():
    with pytest.raises(B.BoundsCompileError):
        B.compile_bounds(lambda p: np.reshape(p[:, 2, :] ** 2 * p[:, 0, :] > 1, (p.shape[0], -1)))   # cubic
    with pytest.raises(B.BoundsCompileError):
        B.compile_bounds(lambda p: np.reshape(np.sin(p[:, 2, :]) > 0, (p.shape[0], -1)))              # unsupported ufunc
    with pytest.raises(B.BoundsCompileError):
        B.compile_bounds(lambda p: (p[:, 2, :] > 0) and (p[:, 2, :] < 5))                              # python `and` on arrays (createGeometry.py:293)


def test_table_errors_match_reference_messages():
    s = cases.sphere_plane()
    s[0].shape = "cone"
    with pytest.raises(Exception, match="Geometry has unrecognized shape"):      # RayToShape.py:29
        T.compile_surfaces(s)
    s = cases.sphere_plane(); s[1].param_list = [np.array([0, 0, 1.]), np.array([0, 0, 0.])]
    with pytest.raises(Exception, match="Invalid plane for RayToPlane"):        # RayToPlane.py:30
        T.compile_surfaces(s)
    s = cases.unifiedtest(params=(0, 0, .6, .5, .1))
    with pytest.raises(Exception, match="impropper reflection parameters"):     # UnifiedReflectorModel.py:113-114
        T.compile_surfaces(s)


def test_table_layout_and_defaults():
    tab, atoms, code = T.compile_surfaces(cases.testgeometry())
    assert tab.shape == (5, 32) and list(tab[:, 0]) == [2, 2, 0, 0, 0] and list(tab[:, 1]) == [2, 2, 0, 0, 0]
    assert np.all(np.isinf(tab[:, 18:22]))                       # RayTracer2.m:263-269 defaults
    assert list(tab[2, 22:27]) == [0, 1, 0, 1, 0]               # RayTracer2.m:271-274
    assert np.allclose(tab[0, 5:8], [0, 0, 1]) and tab[0, 8] == 10
    lut = T.rayleigh_lut(); ol = O.rayleigh_lut()
    assert lut.shape == (100001,) and np.allclose(lut, ol, rtol=0, atol=1e-12) and lut[0] == 0 and abs(lut[-1] - 2 * np.pi) < 1e-12


def test_argument_handling():
    from raytracer_amd.RayTracer2 import parse_args, order_records, RayTracer2
    rays = np.zeros((3, 10)); rays[:, 6] = 1
    assert parse_args(rays, 10, np.spacing(1.0), 0, [], "python") == (10, np.spacing(1.0), (0.0, 0.0), -1.0)       # Q19 accepted
    assert parse_args(rays, [18], [1e-5], [1e-5, 100], [], "python") == (18, 1e-5, (1e-5, 100.0), -1.0)           # BubbleImageCreator.py:129
    assert parse_args(rays, [18], [1e-5], [1e-5, 100], [], "matlab")[3] == 1.0                                      # RayTracer2.m:279
    assert parse_args(rays, 5, 1e-6, 2.0, [], "python")[3] == 1.0                                                   # RayTracer2.py:252
    with pytest.raises(Exception, match="Improper input to RayTracer2"):                                              # RayTracer2.py:216-217
        RayTracer2(np.zeros((3, 3)), np.zeros((3, 10)), [])
    with pytest.raises(Exception, match="Improper input to RayTracer2"):
        RayTracer2(np.zeros((3, 3)), np.zeros((4, 10)), cases.sphere_plane())
    # reference order: step 1 = [refracted children in parent order; reflected children in parent order]
    step = np.array([1, 0, 1, 0, 1, 1]); ray = np.array([7, 9, 9, 7, 7, 9]); parent = np.array([3, -1, 1, -1, 3, 1]); flags = np.array([4, 0, 0, 0, 0, 4])
    g = order_records(step, ray, parent, flags, 2)
    assert list(g[0]) == [3, 1] and list(g[1]) == [4, 2, 0, 5]


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(rt_\w+)\s*\(", hdr, re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    so = os.path.join(ROOT, "raytracer_amd", "librt_mi355x.so")
    assert os.path.exists(so), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    L = ctypes.CDLL(so)  # loads without a GPU; no compute call is made here
    for name in declared:
        assert hasattr(L, name), name


def test_product_never_imports_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M) and "librt_oracle" not in src and "rt_oracle.c\"" not in src, f


# ---- device header compiled for the host vs the oracle (two independent implementations of the same spec)
def test_hostsim_config1_nodice():
    z = np.load(os.path.join(ROOT, "tests", "golden", "g1_cfg1_nodice_256.npz"))
    a = O.trace(z["sp"], z["rays"], cases.sphere_plane(), [10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False, compat="python-strict")
    b = H.trace(z["sp"], z["rays"], cases.sphere_plane(), 10, 1e-9, (1e-6, 1e-6), -1.0, True, False, compat="python-strict")
    compare_results(a, b)


@pytest.mark.parametrize("kw", [{}, dict(params=(.2, .8, .2, .3, .1), n_panel_in=1.5, n_med=1.0)])
def test_hostsim_unified(kw):
    sp, rays = cases.unifiedtest_rays(4000)
    compare_results(O.trace(sp, rays, cases.unifiedtest(**kw), 10, seed=9), H.trace(sp, rays, cases.unifiedtest(**kw), 10, seed=9))


def test_hostsim_mixed_geometry_all_paths():
    sp, rays = cases.isotropic_rays(4000, 21, origin=(1, 2, 3), polarized=True)
    a = O.trace(sp, rays.copy(), cases.mixed_geometry(), 25, 1e-7, 1e-5, -1, True, True, False, seed=31, compat="matlab", final=True)
    b = H.trace(sp, rays.copy(), cases.mixed_geometry(), 25, 1e-7, (1e-5, 1e-5), -1.0, True, True, compat="matlab", seed=31, final=True)
    compare_results(a, b, growth=3.0)
    assert np.allclose(a.final, b.final, rtol=1e-6, atol=1e-6, equal_nan=True)
    sp, rays = cases.isotropic_rays(1000, 22, origin=(1, 2, 3))
    a = O.trace(sp, rays.copy(), cases.mixed_geometry(), 8, 1e-7, np.array([2e-2, 2e-2]), 0.5, True, False, False, seed=32, compat="matlab")
    b = H.trace(sp, rays.copy(), cases.mixed_geometry(), 8, 1e-7, (2e-2, 2e-2), 0.5, True, False, compat="matlab", seed=32)
    compare_results(a, b, growth=3.0)


def test_hostsim_arbc_geometry():
    from raytracer_amd import geometries as G
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs())
    assert len(sl) == 39  # CreateArBCGeometry.m has 39 surfaces
    sp, rays = cases.isotropic_rays(3000, 3, origin=(0, 0, 10))
    a = O.trace(sp, rays, sl, 30, [1e-6], 1e-4, -1, True, True, False, compat="matlab", seed=11)
    b = H.trace(sp, rays, sl, 30, 1e-6, (1e-4, 1e-4), -1.0, True, True, compat="matlab", seed=11)
    compare_results(a, b, growth=3.0)
    # the oracle driven by the compiled bounds program equals the oracle calling the lambdas
    c = O.trace(sp, rays, sl, 30, [1e-6], 1e-4, -1, True, True, False, compat="matlab", seed=11, program=T.compile_surfaces(sl))
    compare_results(a, c, tol=1e-12)


def test_geometry_key_sees_a_changed_closure_and_a_changed_global():
    """a lambda whose closed-over bound changes keeps its id; the stale bounds program must not be reused"""
    from raytracer_amd import RayTracer2 as R
    import cases
    bound = [5.0]
    s = cases.Surf(); s.shape = 'sphere'; s.param_list = [np.array([0, 0, 0.]), 10.]; s.n_outside = 1.0; s.n_inside = 1.5; s.surface_type = 'normal'; s.absorption = 0
    lim = 5.0
    s.inbounds_function = lambda p: np.reshape(p[:, 2, :] < lim, (p.shape[0], -1))
    k1 = R._surface_fingerprint(s)
    lim = 7.0                                            # same function object, same id, new closed-over value
    k2 = R._surface_fingerprint(s)
    assert k1 != k2
    lim = 5.0
    assert R._surface_fingerprint(s) == k1
    s.inbounds_function = lambda p: np.reshape(p[:, 2, :] < bound[0], (p.shape[0], -1))
    k3 = R._surface_fingerprint(s); bound[0] = 6.0
    assert R._surface_fingerprint(s) != k3


def _random_record_tree(rng, n_rays, K, split):
    """ordering metadata of a trace as the device leaves it: records of a random tree (dice mode: one child per record;
    ray splitting: up to one refracted and one reflected child), shuffled"""
    step, ray, parent, flags = [], [], [], []
    cur = []                                     # (record index, ray) of the previous step
    for r in rng.permutation(n_rays)[: max(1, int(0.9 * n_rays))]:
        step.append(0); ray.append(int(r) + 1000); parent.append(-1); flags.append(2); cur.append(len(step) - 1)
    for k in range(1, K):
        nxt = []
        for p in cur:
            kids = [b for b in (0, 1) if rng.random() < (0.45 if split else 0.0)] if split else ([int(rng.random() < 0.3)] if rng.random() < 0.8 else [])
            for born in kids:
                step.append(k); ray.append(ray[p]); parent.append(p); flags.append((4 if born else 0) | (1 if (born or flags[p] & 1) else 0)); nxt.append(len(step) - 1)
        cur = nxt
    n = len(step); perm = rng.permutation(n); inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    step = np.array(step, np.int32)[perm]; ray = np.array(ray, np.int64)[perm]; flags = np.array(flags, np.int32)[perm]
    parent = np.array(parent, np.int64)[perm]; parent = np.where(parent >= 0, inv[np.maximum(parent, 0)], -1)
    return step, ray, parent, flags


def test_library_orders_records_like_the_numpy_rule_and_normalises_like_numpy():
    """rt_order_records / rt_normalize_rays (plain host code of the library: no GPU needed) against the numpy statements of
    python/RayTracer2.py:531-540 and :221-222 they replace in the drop-in call"""
    import ctypes as C
    from raytracer_amd import _lib
    from raytracer_amd.RayTracer2 import order_records, order_groups
    rng = np.random.default_rng(5)
    for split, n_rays, K in ((False, 3000, 7), (True, 400, 9), (False, 1, 3), (True, 70000, 4)):
        step, ray, parent, flags = _random_record_tree(rng, n_rays, K, split)
        want = order_records(step, ray, parent, flags, K)
        got = order_groups(step, ray, parent, flags, K)
        assert len(got) == len(want) == K
        for a_, b_ in zip(got, want):
            assert np.array_equal(a_, b_)
    # metadata that is not such a tree: the library declines (-2) and order_groups falls back to the sort
    L = _lib.load()
    step = np.array([0, 0], np.int32); ray = np.array([5, 5], np.int64); parent = np.array([-1, -1], np.int64); flags = np.zeros(2, np.int32)
    order = np.zeros(2, np.int64); counts = np.zeros(1, np.int64)
    I32, I64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    assert L.rt_order_records(step.ctypes.data_as(I32), ray.ctypes.data_as(I64), parent.ctypes.data_as(I64), flags.ctypes.data_as(I32), 2, 1, order.ctypes.data_as(I64), counts.ctypes.data_as(I64)) == -2
    assert [list(g) for g in order_groups(step, ray, parent, flags, 1)] == [list(g) for g in order_records(step, ray, parent, flags, 1)]
    # normalisation: the same bits as the numpy expression, zero and non-finite rows included
    rays = rng.normal(size=(200001, 10)) * 10.0 ** rng.integers(-150, 150, size=(200001, 1))
    rays[7, 0:3] = 0; rays[9, 3:6] = np.inf; rays[11, 1] = np.nan
    want = rays.copy()
    with np.errstate(all="ignore"):
        want[:, 0:3] = want[:, 0:3] / np.abs(np.sqrt(np.sum(want[:, 0:3] ** 2, axis=1, keepdims=True)))
        want[:, 3:6] = want[:, 3:6] / np.abs(np.sqrt(np.sum(want[:, 3:6] ** 2, axis=1, keepdims=True)))
    mx = C.c_double(0.0)
    assert L.rt_normalize_rays(rays.ctypes.data_as(C.POINTER(C.c_double)), rays.shape[0], C.byref(mx)) == 0
    assert np.array_equal(rays, want, equal_nan=True) and mx.value == np.max(want[:, 6])
    rays[5, 6] = np.nan
    assert L.rt_normalize_rays(rays.ctypes.data_as(C.POINTER(C.c_double)), rays.shape[0], C.byref(mx)) == 0 and np.isnan(mx.value)   # numpy.max: NaN if any
