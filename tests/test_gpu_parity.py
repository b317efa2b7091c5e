"""GPU parity tests: the HIP path (through the C ABI, via the reference-signature wrapper) against the CPU oracle on the
same seeded inputs, and against the golden fixtures generated from the numpy reference.
Deterministic mode: identical hit/ray sequences, 1e-9 relative on every field (north_star).  MC mode: the oracle shares
the Philox convention, so records are compared ray by ray too; fixtures from the reference are compared at 3 sigma."""
import os
import types

import numpy as np
import pytest

import cases
from compare_util import compare_results
from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def gpu(*a, **k):
    from raytracer_amd.RayTracer2 import RayTracer2
    out = RayTracer2(*a, **k)
    r = types.SimpleNamespace(ray_interfaces=list(out[0]), absorption_table=out[1], raytable=out[2])
    if len(out) > 3:
        r.final = out[3]
    return r


def test_golden_config1_nodice():
    z = np.load(os.path.join(G, "g1_cfg1_nodice_256.npz"))
    got = gpu(z["sp"], z["rays"].copy(), cases.sphere_plane(), [10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False, compat="python-strict")
    ref = types.SimpleNamespace(ray_interfaces=cases.unpack_interfaces(z), absorption_table=z["table"])
    compare_results(ref, got)


def test_golden_testgeometry_normal_nodice():
    z = np.load(os.path.join(G, "g3_testgeom_normal_nodice_200.npz"))
    got = gpu(z["sp"], z["rays"].copy(), cases.testgeometry(cyl_type="normal"), [8], [1e-9], np.array([1e-4, 1e-4]), [], True, False, False, compat="python-strict")
    ref = types.SimpleNamespace(ray_interfaces=cases.unpack_interfaces(z), absorption_table=z["table"])
    compare_results(ref, got)


def test_config1_nodice_1e5_vs_oracle_and_golden_summary():
    z = np.load(os.path.join(G, "g2_cfg1_nodice_1e5_summary.npz"))
    sp, rays = cases.isotropic_rays(100000, 0, origin=(3, 0, 0))
    args = ([10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False)
    got = gpu(sp, rays.copy(), cases.sphere_plane(), *args, compat="python-strict")
    assert [len(r.ray_index) for r in got.ray_interfaces] == list(z["nper"])
    assert np.allclose(z["table"], got.absorption_table, rtol=1e-9, atol=1e-9)
    ref = O.trace(sp, rays.copy(), cases.sphere_plane(), *args, compat="python-strict")
    compare_results(ref, got)


@pytest.mark.parametrize("kw", [{}, dict(params=(0, .8, .2, .3, .1), n_panel_in=1.5, n_med=1.0), dict(params=(.2, .8, .2, .3, .1), n_panel_in=1.5, n_med=1.0)])
def test_unified_dice_vs_oracle(kw):
    sp, rays = cases.unifiedtest_rays(20000)
    got = gpu(sp, rays.copy(), cases.unifiedtest(**kw), 10, seed=9)
    ref = O.trace(sp, rays.copy(), cases.unifiedtest(**kw), 10, seed=9)
    compare_results(ref, got)


def test_testgeometry_unified_dice_vs_oracle():
    sp, rays = cases.isotropic_rays(20000, 2, origin=(0, 0, 7))
    got = gpu(sp, rays.copy(), cases.testgeometry(), 10, [1e-9], seed=4)
    ref = O.trace(sp, rays.copy(), cases.testgeometry(), 10, [1e-9], seed=4)
    compare_results(ref, got)


def test_unifiedtest_histogram_vs_reference_fixture():
    from test_oracle_golden import _hemi_hist, _chi2_ok
    z = np.load(os.path.join(G, "g5_unifiedtest_2e5.npz"))
    n = 400000
    sp, rays = cases.unifiedtest_rays(n)
    got = gpu(sp, rays, cases.unifiedtest(), 10, seed=123)
    hits = got.ray_interfaces[1].intersection_point
    nref = int(z["n"])
    f_ref = len(z["hits"]) / nref; f = len(hits) / n
    assert abs(f - f_ref) < 3 * np.sqrt(f_ref * (1 - f_ref) * (1 / n + 1 / nref)) + 1e-9
    ok, chi2, dof = _chi2_ok(_hemi_hist(z["hits"].astype(float)), len(z["hits"]), _hemi_hist(hits), len(hits))
    assert ok, (chi2, dof)


def test_final_rows_and_conservation():
    sp, rays = cases.isotropic_rays(50000, 7, origin=(3, 0, 0))
    got = gpu(sp, rays.copy(), cases.sphere_plane(), 20, 1e-9, 1e-6, -1, False, True, False, seed=5, return_final=True, compat="matlab")
    ref = O.trace(sp, rays.copy(), cases.sphere_plane(), 20, 1e-9, 1e-6, -1, False, True, False, seed=5, final=True, compat="matlab")
    assert np.allclose(ref.absorption_table, got.absorption_table, rtol=1e-9, atol=1e-9)
    assert np.allclose(ref.final, got.final, rtol=1e-9, atol=1e-9, equal_nan=True)
    t = got.absorption_table  # MATLAB/Examples/RunExamples/RunSBCGeometry.m:87-92
    assert abs(t[:, 0:4].sum() + t[-1, 4].sum() - 50000) < 1e-6
