"""GPU parity on the paths the numpy port cannot run (torus, quadsurface, diffuse, retro, Rayleigh, bulk absorption,
UNIFIED with sigma_alpha > 0, tir_handling >= 0): HIP vs the CPU oracle (which follows the MATLAB text; "parity
unpinned" by any executable reference, see oracle/rt_oracle.c), plus the host-side mechanics (chunking, ordering)."""
import types

import numpy as np
import pytest

import cases
from compare_util import compare_results
from oracle import oracle as O
from test_gpu_parity import gpu

pytestmark = pytest.mark.gpu


def test_mixed_geometry_dice_vs_oracle():
    sp, rays = cases.isotropic_rays(30000, 21, origin=(1, 2, 3), polarized=True)
    args = (25, 1e-7, 1e-5, -1, True, True, False)
    got = gpu(sp, rays.copy(), cases.mixed_geometry(), *args, seed=31, compat="matlab")
    ref = O.trace(sp, rays.copy(), cases.mixed_geometry(), *args, seed=31, compat="matlab")
    compare_results(ref, got, growth=3.0, tol_max=1e-4)  # 25 bounces in a closed, curved scene: rounding is amplified
    assert sum(int(r.is_rayleigh.sum()) for r in got.ray_interfaces) > 100          # the Rayleigh path ran
    assert any(np.any(r.surface_number == 1) for r in got.ray_interfaces)            # torus hit
    assert any(np.any(r.surface_number == 2) for r in got.ray_interfaces)            # quadric hit


def test_mixed_geometry_nodice_tir_positive_vs_oracle():
    sp, rays = cases.isotropic_rays(3000, 22, origin=(1, 2, 3))
    args = (8, 1e-7, np.array([2e-2, 2e-2]), 0.5, True, False, False)
    got = gpu(sp, rays.copy(), cases.mixed_geometry(), *args, seed=32, compat="matlab")
    ref = O.trace(sp, rays.copy(), cases.mixed_geometry(), *args, seed=32, compat="matlab")
    compare_results(ref, got, growth=3.0)


@pytest.mark.parametrize("black_cones,K", [(True, 30), (False, 6)])
def test_arbc_geometry_dice_vs_oracle(black_cones, K):
    """ray-by-ray against the oracle.  With the reflective cones light bounces on for all K steps and last-bit differences
    grow by a factor of a few per bounce (a billiard), so the record-level comparison stops after 6 steps there; the
    long chains are compared as distributions in test_gpu_level.py::test_tally_only_runs_take_the_lean_step."""
    from raytracer_amd import geometries as G
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs(black_cones=black_cones))
    sp, rays = cases.isotropic_rays(20000, 3, origin=(0, 0, 10))
    args = (K, [1e-6], 1e-4, -1, True, True, False)
    got = gpu(sp, rays.copy(), sl, *args, compat="matlab", seed=11)
    ref = O.trace(sp, rays.copy(), sl, *args, compat="matlab", seed=11)
    compare_results(ref, got, growth=3.0)


def test_sbc_geometry_bulk_rayleigh_vs_oracle():
    from raytracer_amd import geometries as G
    sl = G.create_sbc_geometry()
    sp, rays = cases.isotropic_rays(20000, 4, origin=(0, 0.6, 1.3))
    args = (100, 1e-6, 1e-4, -1, True, True, True)  # the RunSBCGeometry.m:76-77 call, with records
    got = gpu(sp, rays.copy(), sl, *args, compat="matlab", seed=12)
    ref = O.trace(sp, rays.copy(), sl, *args, compat="matlab", seed=12)
    compare_results(ref, got, growth=2.0)
    assert np.allclose(ref.raytable, got.raytable, rtol=1e-6, atol=1e-6)
    t = got.absorption_table
    assert abs(t[:, 0:4].sum() + t[-1, 4].sum() - 20000) < 1  # RunSBCGeometry.m:87-92


def test_chunked_equals_unchunked():
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    sp, rays = cases.isotropic_rays(40000, 8, origin=(3, 0, 0))
    a = RayTracer2(sp, rays.copy(), cases.sphere_plane(), 10, 1e-9, 1e-6, -1, True, True, False, seed=3)
    c = Context(0); c.set_option("workspace_gb", 0.02)  # forces several chunks of ~5000 rays
    b = RayTracer2(sp, rays.copy(), cases.sphere_plane(), 10, 1e-9, 1e-6, -1, True, True, False, seed=3, ctx=c)
    c.close()
    compare_results(types.SimpleNamespace(ray_interfaces=list(a[0]), absorption_table=a[1]), types.SimpleNamespace(ray_interfaces=list(b[0]), absorption_table=b[1]), tol=1e-12)


@pytest.mark.parametrize("full_output", [True, False])
def test_upload_pipeline_equals_one_copy(full_output):
    """host inputs of a large run go through two pinned stages, "upload_chunk" rays at a time, one being filled by host
    threads while the other is copied (made small here so that 40 000 rays are 14 pieces): same records, final rows and
    tallies as the run that hands the pageable arrays to one hipMemcpy, with records and in the tally-only lean step"""
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    sp, rays = cases.isotropic_rays(40000, 8, origin=(3, 0, 0))
    args = (10, 1e-9, 1e-6, -1, full_output, True, False)
    out = []
    for chunk in (0, 3000):
        c = Context(0)
        try:
            c.set_option("upload_chunk", chunk); c.set_option("level_min_rays", 0)
            out.append(RayTracer2(sp, rays.copy(), cases.sphere_plane(), *args, seed=3, ctx=c, return_final=True))
        finally:
            c.close()
    a, b = out
    assert np.allclose(a[1], b[1], rtol=1e-12, atol=1e-9) and a[1].any()
    assert np.array_equal(a[3], b[3], equal_nan=True)  # final rows: one per ray, in ray order
    if full_output:
        compare_results(types.SimpleNamespace(ray_interfaces=list(a[0]), absorption_table=a[1]), types.SimpleNamespace(ray_interfaces=list(b[0]), absorption_table=b[1]), tol=1e-12)


def test_nodice_small_queue_retries():
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    sp, rays = cases.isotropic_rays(100000, 0, origin=(3, 0, 0))
    args = ([10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False)
    c = Context(0); c.set_option("nodice_chunk", 30000)
    b = RayTracer2(sp, rays.copy(), cases.sphere_plane(), *args, ctx=c)
    c.close()
    assert [len(r.ray_index) for r in b[0]] == [100000, 139957, 138120, 124864, 127220, 35043]


def test_nodice_overflowing_pass_is_repeated_inside_the_call():
    """a ray-splitting pass whose record buffer overflows (a workspace of 20 MB holds 65 536 records, a pass of 30 000 rays
    makes 200 000) is traced again in smaller passes from where it began, by the library (include/rt_mi355x.h,
    rt_result_pass_retries): same records and tallies as the run with room for everything, and no retry from ray 0"""
    import ctypes as C
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    sp, rays = cases.isotropic_rays(100000, 0, origin=(3, 0, 0))
    args = ([10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False)
    a = RayTracer2(sp, rays.copy(), cases.sphere_plane(), *args, seed=5)
    c = Context(0); c.set_option("nodice_chunk", 30000); c.set_option("workspace_gb", 0.02)
    try:
        b = RayTracer2(sp, rays.copy(), cases.sphere_plane(), *args, seed=5, ctx=c)
        n = C.c_int32(); c._check(c.L.rt_result_pass_retries(c.h, C.byref(n)))
    finally:
        c.close()
    assert n.value >= 1, n.value
    assert [len(r.ray_index) for r in a[0]] == [len(r.ray_index) for r in b[0]]
    compare_results(types.SimpleNamespace(ray_interfaces=list(a[0]), absorption_table=a[1]), types.SimpleNamespace(ray_interfaces=list(b[0]), absorption_table=b[1]), tol=1e-12)


def test_point_source_generation_and_trace():
    import ctypes as C
    from raytracer_amd import _lib
    from raytracer_amd.RayTracer2 import Context
    from raytracer_amd import geometries as G, table as T
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs(black_cones=True))   # short chains: comparable to rounding (see above)
    prog = T.compile_surfaces(sl)
    c = Context(0); c.set_program(*prog)
    n = 50000; org = (C.c_double * 3)(0, 0, 10)
    sp = np.zeros((n, 3)); rays = np.zeros((n, 10))
    c._check(c.L.rt_generate_point_source(c.h, org, n, 77, 1000, sp.ctypes.data_as(C.POINTER(C.c_double)), rays.ctypes.data_as(C.POINTER(C.c_double))))
    assert np.allclose(np.linalg.norm(rays[:, 0:3], axis=1), 1) and np.allclose(np.sum(rays[:, 0:3] * rays[:, 3:6], 1), 0, atol=1e-12)
    assert abs(rays[:, 2].mean()) < 4 / np.sqrt(3 * n)  # isotropic
    prm = c.params(100, _lib.FLAG_SINGLECHILD, 0, 1e-6, (1e-4, 1e-4), -1.0, 77, 1000)
    c._check(c.L.rt_trace_point_source(c.h, org, n, C.byref(prm)))
    res = c.results(False, False, n)
    ref = O.trace(sp, rays, sl, 100, 1e-6, 1e-4, -1, False, True, False, compat="matlab", seed=77, ray_id_base=1000)
    assert np.allclose(ref.absorption_table, res["table"], rtol=1e-8, atol=1e-8)
    c.close()
