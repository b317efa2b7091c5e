"""GPU parity on the edges of the input space (python/RayTracer2.py:216-222, 250-270, 404-407): one ray, rays that
miss everything, a zero direction, zero intensity, a NaN starting point, max_scat = 1 - each against the oracle on the
same inputs - and the error paths of the boundary, which must raise before any device work."""
import types

import numpy as np
import pytest

import cases
from compare_util import compare_results
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def gpu(*a, **k):
    from raytracer_amd.RayTracer2 import RayTracer2
    out = RayTracer2(*a, **k)
    return types.SimpleNamespace(ray_interfaces=list(out[0]), absorption_table=out[1], raytable=out[2])


def one_ray(d=(0, 0, 1), p=(3.0, 0, 0), s0=1.0):
    sp = np.array([p], float); rays = np.zeros((1, 10)); rays[0, 0:3] = d; rays[0, 3:6] = (1, 0, 0); rays[0, 6] = s0
    return sp, rays


def test_one_ray_no_dice():
    sp, rays = one_ray()
    args = ([10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False)
    got = gpu(sp, rays.copy(), cases.sphere_plane(), *args, compat="python-strict")
    ref = O.trace(sp, rays.copy(), cases.sphere_plane(), *args, compat="python-strict")
    assert len(ref.ray_interfaces) == 6
    compare_results(ref, got)


def test_rays_that_miss_everything():
    """nothing scatters: no ray_interfaces, the table keeps one all-zero row (RayTracer2.py:404-407, 542)"""
    sp = np.array([[0, 0, 50.0], [0, 0, 60.0]]); rays = np.zeros((2, 10)); rays[:, 2] = 1; rays[:, 3] = 1; rays[:, 6] = 1
    got = gpu(sp, rays.copy(), cases.sphere_plane(), 10, compat="matlab", seed=1)
    ref = O.trace(sp, rays.copy(), cases.sphere_plane(), 10, compat="matlab", seed=1)
    assert len(got.ray_interfaces) == len(ref.ray_interfaces) == 0
    assert got.absorption_table.shape == ref.absorption_table.shape and not got.absorption_table.any()


@pytest.mark.parametrize("mutate", ["zero_direction", "zero_intensity", "nan_start"])
def test_degenerate_rays_among_good_ones(mutate):
    """a ray with |d| = 0 is not normalised and hits nothing (RayToShape.py:11-14), zero intensity is followed or not by the
    thresholds alone, a NaN starting point fails every comparison: the other rays are not disturbed.  (min_travel_len
    1e-9: with the default of one ulp, whether a ray re-hits the surface it has just left 1e-15 further on is decided by
    the last bit of the intersection arithmetic, in the reference as here.)"""
    sp, rays = cases.isotropic_rays(2000, 3, origin=(3, 0, 0))
    if mutate == "zero_direction":
        rays[::7, 0:3] = 0
    elif mutate == "zero_intensity":
        rays[::2, 6] = 0
    else:
        sp[::5, 0] = np.nan
    with np.errstate(all="ignore"):
        got = gpu(sp, rays.copy(), cases.sphere_plane(), 6, [1e-9], compat="matlab", seed=1)
        ref = O.trace(sp, rays.copy(), cases.sphere_plane(), 6, [1e-9], compat="matlab", seed=1)
    assert [r.ray_index.size for r in got.ray_interfaces] == [r.ray_index.size for r in ref.ray_interfaces]
    compare_results(ref, got)


def test_max_scat_one():
    sp, rays = cases.isotropic_rays(1000, 3, origin=(3, 0, 0))
    got = gpu(sp, rays.copy(), cases.sphere_plane(), 1, [1e-9], compat="matlab", seed=1)
    ref = O.trace(sp, rays.copy(), cases.sphere_plane(), 1, [1e-9], compat="matlab", seed=1)
    assert len(got.ray_interfaces) == 1 and got.absorption_table.shape[0] == 1
    compare_results(ref, got)


def test_errors_come_before_device_work():
    """python/RayTracer2.py:216-217, UnifiedReflectorModel.py:112-114: same exceptions, nothing traced"""
    from raytracer_amd.RayTracer2 import RayTracer2
    sp, rays = one_ray()
    sl = cases.sphere_plane()
    for bad in ((sp, rays, []), (sp[0], rays, sl), (sp, rays[:, :9], sl), (sp[:, :2], rays, sl), (np.vstack([sp, sp]), rays, sl), (sp[:0], rays[:0], sl)):
        with pytest.raises(Exception, match="Improper input to RayTracer2"):
            RayTracer2(*bad)
    su = cases.unifiedtest(params=(.1, .5, .6, .5, .1))  # lobe + spike + back > 1
    with pytest.raises(Exception):
        RayTracer2(*cases.unifiedtest_rays(10), su)


def test_record_fields_refuses_bad_requests():
    import ctypes as C
    from raytracer_amd.RayTracer2 import RayTracer2, get_context
    sp, rays = cases.isotropic_rays(100, 3, origin=(3, 0, 0))
    RayTracer2(sp, rays.copy(), cases.sphere_plane(), 3, compat="matlab", seed=1)
    c = get_context(0)
    n = C.c_int64(); c.L.rt_result_num_records(c.h, C.byref(n))
    out = np.zeros((2, 3)); ptr = (C.POINTER(C.c_double) * 1)(out.ctypes.data_as(C.POINTER(C.c_double)))
    order = np.array([0, n.value], np.int64)  # one past the end
    one = (C.c_int32 * 1)
    assert c.L.rt_result_record_fields(c.h, order.ctypes.data_as(C.POINTER(C.c_int64)), 2, 1, one(30), one(3), ptr) == -1
    assert b"out of range" in c.L.rt_last_error(c.h)
    order[1] = 0
    assert c.L.rt_result_record_fields(c.h, order.ctypes.data_as(C.POINTER(C.c_int64)), 2, 1, one(46), one(3), ptr) == -1
    assert c.L.rt_result_record_fields(c.h, order.ctypes.data_as(C.POINTER(C.c_int64)), 2, 1, one(30), one(3), ptr) == 0
    assert np.array_equal(out[0], out[1]) and np.isfinite(out).all()
