"""The BASELINE.json configurations at their full sizes, through properties that do not need the oracle to finish:
energy conservation (MATLAB/Examples/RunExamples/RunSBCGeometry.m:87-92: everything absorbed or escaped over all steps
plus what is still followed after the last one equals what went in), independence of how the rays are split over
calls / GPUs (ray ids key the random streams), and - for the UNIFIED reflector - the lobe fractions its parameters ask
for.  Tallies only (full_output off), rays generated on the device where the configuration has a generator."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _total(t):
    return t[:, 0:4].sum() + t[-1, 4].sum()


def _ctx(prog):
    from raytracer_amd.RayTracer2 import Context
    c = Context(0); c.set_program(*prog)
    return c


def _pad(t, rows):
    out = np.zeros((rows,) + t.shape[1:]); out[:t.shape[0]] = t
    return out


def test_config2_unifiedtest_1e7_rays():
    """UnifiedTest.py geometry, 1e7 identical rays, dice mode, K=10"""
    from raytracer_amd.RayTracer2 import RayTracer2
    n = 10_000_000
    sp, rays = cases.unifiedtest_rays(n)
    sl = cases.unifiedtest()
    from raytracer_amd.RayTracer2 import get_context
    _, t, _ = RayTracer2(sp, rays, sl, 10, full_output=False, seed=99, compat="matlab")   # 'python' tallies surface 0 twice
    live, scat = get_context(0).step_counts(10)
    assert abs(_total(t) - n) < 1e-6 * n
    # the panel is hit once by every ray; (spike 0.5 + back 0.1 + lobe 0.25 with sigma_alpha = 0 ...) the reflected share that
    # reaches the hemisphere is what the reference fixture measured at 2e5 rays
    z = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "g5_unifiedtest_2e5.npz"))
    nref = int(z["n"]); f_ref = len(z["hits"]) / nref
    assert scat[0] == n
    f = scat[1] / n                                              # rays with a second interaction
    assert abs(f - f_ref) < 4 * np.sqrt(f_ref * (1 - f_ref) * (1 / n + 1 / nref)) + 1e-9
    # two halves with their own ray id ranges: same tallies (atomics: order of addition only)
    h = n // 2
    _, ta, _ = RayTracer2(sp[:h], rays[:h], sl, 10, full_output=False, seed=99, compat="matlab")
    _, tb, _ = RayTracer2(sp[h:], rays[h:], sl, 10, full_output=False, seed=99, ray_id_base=h, compat="matlab")
    rows = max(t.shape[0], ta.shape[0], tb.shape[0])
    assert np.allclose(_pad(ta, rows) + _pad(tb, rows), _pad(t, rows), rtol=1e-9, atol=1e-6)


def test_config3_camera_1e8_rays():
    """BubbleImageCreator.py:129 on the jar geometry with 1e8 camera rays generated in HBM"""
    from raytracer_amd import _lib, geometries as G, table as T
    specs = G.ArBCSpecs(cam_resolution=np.array([8660, 11548]))
    cam = G.arbc_camera(specs)
    n = 8660 * 11548
    assert n > 1e8
    c = _ctx(T.compile_surfaces(G.create_arbc_geometry(specs)))
    try:
        def run(workspace_gb):
            c.set_option("workspace_gb", workspace_gb)
            prm = c.params(18, _lib.FLAG_SINGLECHILD, 0, 1e-5, (1e-5, 100.0), -1.0, 7, 0)
            c._check(c.L.rt_trace_camera(c.h, C.byref(cam), C.byref(prm)))
            return c.results(False, False, n)["table"]
        t = run(64)
        assert abs(_total(t) - n) < 1e-9 * n                     # unit Stokes vector per pixel
        assert t[0].sum() > 0.2 * n                               # the window and the can walls take their share at once
        t2 = run(2)                                               # same run in many chunks
        assert t2.shape == t.shape and np.allclose(t2, t, rtol=1e-9, atol=1e-6)
    finally:
        c.close()


def test_config4_ray_splitting_1e6_rays():
    """no-dice mode on the jar geometry: the ray count grows per bounce until the thresholds bite"""
    from raytracer_amd.RayTracer2 import RayTracer2
    from raytracer_amd import geometries as G
    n = 1_000_000
    sl = G.create_arbc_geometry(G.ArBCSpecs())
    sp, rays = cases.isotropic_rays(n, 11, origin=(0, 0, -5))
    _, t, _ = RayTracer2(sp, rays, sl, 18, 1e-5, [1e-5, 1e-5], [], False, False, seed=1)
    assert abs(_total(t) - n) < 1e-4 * n
    assert t[:, 4].sum() > 3 * n                                  # several generations were followed
    h = n // 2
    _, ta, _ = RayTracer2(sp[:h], rays[:h], sl, 18, 1e-5, [1e-5, 1e-5], [], False, False, seed=1)
    _, tb, _ = RayTracer2(sp[h:], rays[h:], sl, 18, 1e-5, [1e-5, 1e-5], [], False, False, seed=1, ray_id_base=h)
    rows = max(t.shape[0], ta.shape[0], tb.shape[0])
    assert np.allclose(_pad(ta, rows) + _pad(tb, rows), _pad(t, rows), rtol=1e-9, atol=1e-6)


def test_config5_full_mc_one_batch():
    """one bench batch of the full Monte-Carlo configuration: Rayleigh scattering, bulk absorption, UNIFIED reflector,
    point source, K=100 (bench.py's workload)"""
    import bench
    from raytracer_amd import _lib
    sl, prog, origin, kw = bench.build_workload("sbc_full_mc")
    c = _ctx(prog)
    try:
        n = 1 << 25
        org = (C.c_double * 3)(*origin)

        def run(m, base):
            prm = c.params(kw["K"], _lib.FLAG_SINGLECHILD, 0, kw["min_travel"], kw["thr"], kw["tir"], 2024, base)
            c._check(c.L.rt_trace_point_source(c.h, org, m, C.byref(prm)))
            return c.results(False, False, m)["table"], c.step_counts(kw["K"])
        c.set_option("lanes", 1)                                  # one lane first (the library's default is two for a run of this size)
        t, (live, scat) = run(n, 0)
        assert abs(_total(t) - n) < 1e-9 * n
        assert live[0] == n and np.all(np.diff(live) <= 0) and scat.sum() > 3 * n
        (ta, _), (tb, _) = run(n // 2, 0), run(n // 2, n // 2)    # what two GPUs would each do
        rows = max(t.shape[0], ta.shape[0], tb.shape[0])
        assert np.allclose(_pad(ta, rows) + _pad(tb, rows), _pad(t, rows), rtol=1e-9, atol=1e-6)
        c.set_option("lanes", 2)                                  # the same batch as two halves at the same time on two streams
        t2, (live2, scat2) = run(n, 0)
        assert t2.shape == t.shape and np.allclose(t2, t, rtol=1e-9, atol=1e-6) and np.array_equal(live2, live) and np.array_equal(scat2, scat)
        # ... and from host arrays (the second lane takes the second half of the caller's arrays)
        m = 1 << 24
        sp, rays = cases.isotropic_rays(m, 21, origin=origin)
        prm = c.params(kw["K"], _lib.FLAG_SINGLECHILD, 0, kw["min_travel"], kw["thr"], kw["tir"], 77, 0)
        P = C.POINTER(C.c_double)
        tabs = []
        for lanes in (1, 2):
            c.set_option("lanes", lanes)
            c._check(c.L.rt_trace(c.h, sp.ctypes.data_as(P), rays.ctypes.data_as(P), m, C.byref(prm)))
            tabs.append(c.results(False, False, m)["table"])
        assert tabs[0].shape == tabs[1].shape and np.allclose(tabs[0], tabs[1], rtol=1e-9, atol=1e-6) and abs(_total(tabs[1]) - m) < 1e-9 * m
    finally:
        c.close()
