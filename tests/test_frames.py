"""The frame program of the nearest-hit search (rt_device.hpp: build_program / find_hit) and the lean single-child
step (bounce_phys_single), through the host build of the device header, against the oracle - which knows nothing of
frames: it calls the surfaces' lambdas and the reference's per-surface formulas."""
import ctypes as C
import os
from collections import Counter

import numpy as np
import pytest

import cases
import hostsim_util as H
from compare_util import compare_results
from oracle import oracle as O
from raytracer_amd import geometries as G
from raytracer_amd import table as T

KINDS = ("FRAME", "PLANE", "SPHERE", "CYL", "QUAD", "GENERIC", "TORUS", "GROUP")


def program(sl, frames=True):
    tab, atoms, code = T.compile_surfaces(sl)
    dst = np.zeros((4 * len(sl) + 8, 16)); dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    n = H.lib().hs_program(dp(tab), tab.shape[0], dp(atoms), code.ctypes.data_as(C.POINTER(C.c_uint32)), int(frames), dp(dst), dst.shape[0])
    meta = dst[:n, 0].copy().view(np.uint64)
    return [dict(kind=KINDS[int(m) & 15], surf=(int(m) >> 4) & 0xFFF, natoms=(int(m) >> 16) & 3, residual=int(m) >> 52, words=dst[i].copy()) for i, m in enumerate(meta)]


def test_program_of_the_jar_geometry():
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs())
    pr = program(sl)
    c = Counter(r["kind"] for r in pr)
    # 'reflector/diffuser strip cone' (surface 31) is bounded by z > 120 & z < -20: nothing satisfies that, it gets no record
    assert c["FRAME"] == 2 and c["GENERIC"] == 0 and c["TORUS"] == 4 and c["CYL"] == 14 and c["PLANE"] == 11 and c["SPHERE"] == 4 and c["QUAD"] == 5
    assert sorted(r["surf"] for r in pr if r["kind"] not in ("FRAME", "GROUP")) == [s_ for s_ in range(39) if s_ != 31]  # every other surface exactly once
    assert sum(1 for r in pr if r["residual"]) <= 4
    assert all(r["kind"] == "GENERIC" for r in program(sl, frames=False)) and len(program(sl, frames=False)) == 39
    # bounding boxes of the frames' members (a wave skips a frame nobody can reach before its best hit): the z frame holds
    # everything up to the pressure vessel (radius 30, from z = -20 up), the camera frame the can around the viewport
    fr = [r for r in pr if r["kind"] == "FRAME"]
    assert all(r["natoms"] & 1 for r in fr)   # "has box" flag
    assert abs(np.sqrt(fr[0]["words"][7]) - 30) < 1e-6 and abs(fr[0]["words"][8] + 20) < 1e-6 and np.isinf(fr[0]["words"][9])
    # GROUP records: what lies far out in a frame goes behind a test that lets a wave skip it.  In the z frame: the shell
    # beyond radius 12 (reflector cylinder and its cone, then the pressure vessel walls nested inside it) and what lies
    # high up (the outer domes, then the vessel heads above z = 100); counts include nested GROUP records
    zf = pr[1:1 + fr[0]["surf"]]
    assert fr[0]["surf"] == len(zf) and pr[1 + fr[0]["surf"]]["kind"] == "FRAME"
    groups = [(i, r) for i, r in enumerate(zf) if r["kind"] == "GROUP"]
    def inside(i, r):
        return {m["surf"] for m in zf[i + 1:i + 1 + r["surf"]] if m["kind"] != "GROUP"}
    shells = [(i, r) for i, r in groups if r["words"][7] > 0]
    assert abs(shells[0][1]["words"][7] - 144) < 1e-3 and inside(*shells[0]) == {29, 30, 34, 35}
    assert abs(shells[1][1]["words"][7] - 841) < 1e-2 and inside(*shells[1]) == {34, 35}
    above = [(i, r) for i, r in groups if np.isfinite(r["words"][8])]
    assert inside(*above[0]) == {6, 7, 36, 37} and abs(above[-1][1]["words"][8] - 100) < 1e-6 and inside(*above[-1]) == {36, 37}
    first_group = groups[0][0]
    assert {m["surf"] for m in zf[:first_group]} == {0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 32, 33, 38}   # the core, in list order, comes first
    g = G.sbc_full_mc_specs()
    assert abs(np.sqrt(fr[1]["words"][7]) - g.vp_flange_rad) < 1e-6 and fr[1]["words"][8] < 0 < fr[1]["words"][9] < 2 * g.vp_can_OAL


def test_program_of_a_tilted_scene():
    sl, _ = cases.coaxial_scene(5)
    pr = program(sl)
    by = {r["surf"]: r for r in pr if r["kind"] != "FRAME"}
    assert sum(1 for r in pr if r["kind"] == "FRAME") == 1
    assert by[11]["kind"] == "GENERIC"                      # the off-axis sphere fits no frame
    assert by[12]["kind"] == "CYL" and by[12]["residual"] > 0 and by[12]["natoms"] == 1  # half cylinder: axial range inline, the half-space stays generic
    assert all(by[s]["kind"] != "GENERIC" and by[s]["residual"] == 0 for s in range(11))
    assert by[10]["kind"] == "TORUS" and by[10]["natoms"] == 2


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_frames_vs_oracle_tilted_scene(seed):
    sl, origin = cases.coaxial_scene(seed)
    sp, rays = cases.isotropic_rays(3000, 40 + seed, origin=tuple(origin), polarized=True)
    a = O.trace(sp, rays.copy(), sl, 12, 1e-7, 1e-5, -1, True, True, False, seed=seed, compat="matlab")
    b = H.trace(sp, rays.copy(), sl, 12, 1e-7, (1e-5, 1e-5), -1.0, True, True, compat="matlab", seed=seed)
    compare_results(a, b, growth=3.0)
    hit = Counter(int(s) for r in b.ray_interfaces for s in r.surface_number)
    assert len(hit) >= 10  # the scene is exercised, not just its outer wall
    # no-dice mode (ray splitting) takes the same nearest-hit search
    a = O.trace(sp[:400], rays[:400].copy(), sl, 6, 1e-7, np.array([1e-2, 1e-2]), -1, True, False, False, seed=seed, compat="matlab")
    b = H.trace(sp[:400], rays[:400].copy(), sl, 6, 1e-7, (1e-2, 1e-2), -1.0, True, False, compat="matlab", seed=seed)
    compare_results(a, b, growth=3.0)


def test_frames_on_equals_frames_off():
    sl, origin = cases.coaxial_scene(7)
    sp, rays = cases.isotropic_rays(2000, 8, origin=tuple(origin))
    on = H.trace(sp, rays.copy(), sl, 10, 1e-7, (1e-5, 1e-5), -1.0, True, True, compat="matlab", seed=3)
    os.environ["RT_NO_FRAMES"] = "1"
    try:
        off = H.trace(sp, rays.copy(), sl, 10, 1e-7, (1e-5, 1e-5), -1.0, True, True, compat="matlab", seed=3)
    finally:
        del os.environ["RT_NO_FRAMES"]
    compare_results(off, on, growth=3.0)


def test_rays_starting_all_over_the_jar():
    """nearest hit from start points spread through the whole jar geometry (camera can included), not only from the
    source point of the benchmark.  Only the geometry of the first hit is compared: many of these points lie inside
    conductors, where the optics are undefined (n_incident = inf)."""
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs())
    rng = np.random.default_rng(12)
    n = 6000
    _, rays = cases.isotropic_rays(n, 13)
    sp = np.column_stack([rng.uniform(-28, 28, n), rng.uniform(-28, 28, n), rng.uniform(-19, 99, n)])
    sp[: n // 3] = np.array([0, -10.0, 60.0]) + rng.normal(size=(n // 3, 3)) * np.array([3, 3, 8.0])
    a = O.trace(sp, rays.copy(), sl, 1, [1e-6], 1e-4, -1, True, True, False, compat="matlab", seed=5).ray_interfaces[0]
    b = H.trace(sp, rays.copy(), sl, 1, 1e-6, (1e-4, 1e-4), -1.0, True, True, compat="matlab", seed=5).ray_interfaces[0]
    assert np.array_equal(a.ray_index, b.ray_index) and np.array_equal(a.surface_index, b.surface_index)
    assert len(set(np.abs(a.surface_index).astype(int))) >= 30  # nearly every surface is somebody's first hit
    for f in ("distance_traveled", "intersection_point", "surface_normal"):
        assert np.allclose(getattr(a, f), getattr(b, f), rtol=1e-9, atol=1e-9), f


@pytest.mark.parametrize("tir", [-1.0, 0.5])
def test_lean_single_child_step_vs_oracle(tir):
    """tally-only single-child runs take bounce_phys_single(); the oracle takes the reference's order of operations"""
    sp, rays = cases.isotropic_rays(4000, 21, origin=(1, 2, 3), polarized=True)
    a = O.trace(sp, rays.copy(), cases.mixed_geometry(), 25, 1e-7, 1e-5, tir, False, True, False, seed=31, compat="matlab")
    b = H.trace(sp, rays.copy(), cases.mixed_geometry(), 25, 1e-7, (1e-5, 1e-5), tir, False, True, compat="matlab", seed=31)
    assert a.absorption_table.shape == b.absorption_table.shape
    assert np.allclose(a.absorption_table, b.absorption_table, rtol=1e-7, atol=1e-7)
    assert np.array_equal(a.n_scat, b.n_scat)
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs())
    sp, rays = cases.isotropic_rays(6000, 3, origin=(0, 0, 10))
    a = O.trace(sp, rays, sl, 100, [1e-6], 1e-4, tir, False, True, False, compat="matlab", seed=11)
    b = H.trace(sp, rays, sl, 100, 1e-6, (1e-4, 1e-4), tir, False, True, compat="matlab", seed=11)
    # light keeps bouncing off the reflector for all 100 steps; with tir_handling > 0 the reference's weights grow without
    # bound along such chains (1e8 and more, either sign), so the tolerance is relative to the largest tally
    scale = max(1.0, np.abs(a.absorption_table).max())
    assert np.array_equal(a.n_scat, b.n_scat) and np.allclose(a.absorption_table, b.absorption_table, rtol=1e-7, atol=1e-7 * scale)
