"""GPU parity of the level-synchronous path (k_level_init / k_level_hit / k_level_phys / k_level_compact + the
persistent tail) and of the frame program, forced on with rt_set_option so that small inputs take them too.
Through the C ABI, against the CPU oracle (which knows neither frames nor kernels)."""
import ctypes as C
import types

import numpy as np
import pytest

import cases
from compare_util import compare_results
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def gpu_with(options, *a, **k):
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    c = Context(0)
    try:
        for key, v in options.items():
            c.set_option(key, v)
        out = RayTracer2(*a, ctx=c, **k)
    finally:
        c.close()
    r = types.SimpleNamespace(ray_interfaces=list(out[0]), absorption_table=out[1], raytable=out[2])
    if len(out) > 3:
        r.final = out[3]
    return r


@pytest.mark.parametrize("opts", [dict(level_steps=1000, level_min_rays=0), dict(level_steps=2, level_min_rays=0), dict(level_steps=0),
                                  dict(level_steps=1000, level_min_rays=0, frames=0), dict(level_steps=1000, level_min_rays=0, hit_split=1)])
def test_tilted_scene_records_and_final_rows(opts):
    sl, origin = cases.coaxial_scene(2)
    sp, rays = cases.isotropic_rays(20000, 42, origin=tuple(origin), polarized=True)
    args = (12, 1e-7, 1e-5, -1, True, True, False)
    got = gpu_with(opts, sp, rays.copy(), sl, *args, seed=2, compat="matlab", return_final=True)
    ref = O.trace(sp, rays.copy(), sl, *args, seed=2, compat="matlab", final=True)
    compare_results(ref, got, growth=3.0)
    assert np.allclose(ref.final, got.final, rtol=1e-6, atol=1e-6, equal_nan=True)


@pytest.mark.parametrize("opts", [dict(level_steps=1000, level_min_rays=0), dict(level_steps=3, level_min_rays=0), dict(level_steps=0)])
def test_tally_only_runs_take_the_lean_step(opts):
    from raytracer_amd import geometries as G
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs())
    sp, rays = cases.isotropic_rays(30000, 3, origin=(0, 0, 10))
    args = (100, [1e-6], 1e-4, -1, False, True, False)
    got = gpu_with(opts, sp, rays.copy(), sl, *args, seed=11, compat="matlab")
    ref = O.trace(sp, rays.copy(), sl, *args, seed=11, compat="matlab")
    assert ref.absorption_table.shape == got.absorption_table.shape
    # light bounces between jar and reflector for all 100 steps: a billiard, in which the last-bit differences between
    # the two builds of the same arithmetic grow by a factor of a few per bounce.  The first steps agree to rounding,
    # later ones as two samples of the same distribution do (the paths have decorrelated, the random streams have not)
    a, b = ref.absorption_table, got.absorption_table
    assert np.allclose(a[:6], b[:6], rtol=1e-7, atol=1e-7)
    for k in range(a.shape[0]):
        assert abs(a[k].sum() - b[k].sum()) < 5e-3 * a[k].sum() + 1e-9, k
    assert np.allclose(a.sum(axis=(0, 3)), b.sum(axis=(0, 3)), rtol=2e-2, atol=2e-3 * sp.shape[0])   # by category and surface
    # conservation, MATLAB/Examples/RunExamples/RunSBCGeometry.m:87-92
    t = got.absorption_table
    assert abs(t[:, 0:4].sum() + t[-1, 4].sum() - sp.shape[0]) < 1e-6 * sp.shape[0]


def test_steps_between_host_syncs_and_segment_count_change_nothing():
    """run_trace launches up to `sync_every` level steps on the grid of the first before it reads the live count, picks the
    number of queue segments by run size, and with lanes = 2 traces two halves at once: per-ray results do not depend on
    any of that (tallies: float atomics, to 1e-9)"""
    from raytracer_amd import geometries as G
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs(black_cones=True))      # rays die within ~15 steps: batches run past the last live ray
    sp, rays = cases.isotropic_rays(40000, 5, origin=(0, 0, 10))
    args = (40, [1e-6], 1e-4, -1, False, True, False)
    tabs = [gpu_with(dict(level_min_rays=0, **o), sp, rays.copy(), sl, *args, seed=3, compat="matlab").absorption_table
            for o in (dict(sync_every=1, seg_per_cu=8), dict(), dict(sync_every=16, seg_per_cu=1), dict(sync_every=3, sync_rays_max=100),
                      dict(lanes=2, lanes_min_rays=1000),    # the two halves traced at the same time on two streams, taking turns with the nearest-hit launch
                      dict(lanes=3, lanes_min_rays=1000, sync_every=1), dict(lanes=4, lanes_min_rays=1000, lanes_alternate=0))]
    assert tabs[0].any()
    for t in tabs[1:]:
        assert t.shape == tabs[0].shape and np.allclose(t, tabs[0], rtol=1e-9, atol=1e-9)


def test_ray_splitting_runs_take_the_lanes_too():
    """no-dice mode: both children are followed, the halves of the initial rays are traced as two lanes; tables to 1e-9"""
    sl, origin = cases.coaxial_scene(2)
    sp, rays = cases.isotropic_rays(30000, 8, origin=tuple(origin), polarized=True)
    args = (8, 1e-7, [1e-3, 1e-3], -1, False, False, False)
    one = gpu_with(dict(lanes=1, level_min_rays=0), sp, rays.copy(), sl, *args, seed=9, compat="matlab")
    two = gpu_with(dict(lanes=2, lanes_min_rays=1000, level_min_rays=0), sp, rays.copy(), sl, *args, seed=9, compat="matlab")
    ref = O.trace(sp, rays.copy(), sl, *args, seed=9, compat="matlab")
    assert one.absorption_table.any() and one.absorption_table.shape == two.absorption_table.shape == ref.absorption_table.shape
    assert np.allclose(one.absorption_table, two.absorption_table, rtol=1e-9, atol=1e-9)
    assert np.allclose(ref.absorption_table, two.absorption_table, rtol=1e-6, atol=1e-6)


def test_two_lanes_write_the_same_device_final_rows():
    """lanes = 2: the second half of the rays is traced by a second context into the same device buffer of final rows"""
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    from raytracer_amd import geometries as G
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs(black_cones=True))
    sp, rays = cases.isotropic_rays(30001, 9, origin=(0, 0, 10))
    got = []
    for lanes in (1, 2):
        c = Context(0)
        try:
            c.set_option("level_min_rays", 0); c.set_option("lanes", lanes); c.set_option("lanes_min_rays", 1000)
            out = RayTracer2(sp, rays.copy(), sl, 30, [1e-6], 1e-4, -1, False, True, False, seed=4, compat="matlab", ctx=c, return_final="device")
            got.append((out[1].copy(), out[3].cpu().numpy().copy()))
        finally:
            c.close()
    assert np.allclose(got[0][0], got[1][0], rtol=1e-9, atol=1e-9) and got[0][0].any()
    assert np.array_equal(got[0][1], got[1][1], equal_nan=True) and np.any(got[0][1][:, 19] > 0)


@pytest.mark.parametrize("opts", [dict(level_steps=1000, level_min_rays=0), dict(level_steps=0)])
def test_jar_with_diffuse_reflectors(opts):
    """SURVEY 8d config 5 as worded: 'diffuse' reflector surfaces (MATLAB/RayTracer2.m:394-415) in the jar geometry with
    Rayleigh scattering and absorption in the target (bench.py --workload sbc_full_mc_diffuse).  Every reflector hit
    re-draws the direction from the ray's own random stream, so rounding differences do not grow as they do off the
    UNIFIED mirror facets: the oracle's tallies are met step by step for all 100 steps."""
    from raytracer_amd import geometries as G
    sl = G.create_arbc_geometry(G.sbc_full_mc_specs(reflector="diffuse"))
    sp, rays = cases.isotropic_rays(20000, 3, origin=(0, 0, 10))
    args = (100, [1e-6], 1e-4, -1, False, True, False)
    got = gpu_with(opts, sp, rays.copy(), sl, *args, seed=11, compat="matlab")
    ref = O.trace(sp, rays.copy(), sl, *args, seed=11, compat="matlab")
    a, b = ref.absorption_table, got.absorption_table
    assert a.shape == b.shape
    assert np.allclose(a[:6], b[:6], rtol=1e-8, atol=1e-8)
    for k in range(a.shape[0]):
        assert abs(a[k].sum() - b[k].sum()) < 1e-3 * a[k].sum() + 1e-9, k
    assert np.allclose(a.sum(axis=(0, 3)), b.sum(axis=(0, 3)), rtol=5e-3, atol=1e-3 * sp.shape[0])
    t = got.absorption_table
    assert abs(t[:, 0:4].sum() + t[-1, 4].sum() - sp.shape[0]) < 1e-6 * sp.shape[0]   # RunSBCGeometry.m:87-92


def test_mixed_geometry_all_surface_types_level_path():
    sp, rays = cases.isotropic_rays(30000, 21, origin=(1, 2, 3), polarized=True)
    args = (25, 1e-7, 1e-5, 0.5, True, True, False)
    got = gpu_with(dict(level_steps=1000, level_min_rays=0), sp, rays.copy(), cases.mixed_geometry(), *args, seed=31, compat="matlab")
    ref = O.trace(sp, rays.copy(), cases.mixed_geometry(), *args, seed=31, compat="matlab")
    compare_results(ref, got, growth=3.0, tol_max=1e-4)


def test_level_path_chunked_by_workspace():
    """a workspace too small for one pass: the level path runs chunk after chunk, results are the same"""
    sl, origin = cases.coaxial_scene(3)
    sp, rays = cases.isotropic_rays(40000, 43, origin=tuple(origin))
    args = (10, 1e-7, 1e-5, -1, False, True, False)
    a = gpu_with(dict(level_steps=1000, level_min_rays=0), sp, rays.copy(), sl, *args, seed=4, compat="matlab")
    b = gpu_with(dict(level_steps=1000, level_min_rays=0, workspace_gb=3e-3), sp, rays.copy(), sl, *args, seed=4, compat="matlab")
    assert a.absorption_table.shape == b.absorption_table.shape and np.allclose(a.absorption_table, b.absorption_table, rtol=1e-9, atol=1e-9)


def test_record_streaming_npz_shards(tmp_path):
    """with a record sink the records leave chunk by chunk; together the shards hold exactly what an
    accumulating run returns."""
    from raytracer_amd.RayTracer2 import Context, RayTracer2, npz_shards
    sl, origin = cases.coaxial_scene(4)
    sp, rays = cases.isotropic_rays(30000, 44, origin=tuple(origin), polarized=True)
    args = (10, 1e-7, 1e-5, -1, True, True, False)
    whole = gpu_with(dict(level_steps=1000, level_min_rays=0), sp, rays.copy(), sl, *args, seed=6, compat="matlab")
    c = Context(0)
    try:
        c.set_option("workspace_gb", 0.03)  # ~7000 rays of records per chunk
        sink = npz_shards(str(tmp_path))
        out = RayTracer2(sp, rays.copy(), sl, *args, seed=6, compat="matlab", ctx=c, record_sink=sink)
    finally:
        c.close()
    assert len(out[0]) == 0 and len(sink.paths) >= 3
    assert np.allclose(out[1], whole.absorption_table, rtol=1e-9, atol=1e-9)
    shards = [np.load(p) for p in sink.paths]
    assert sum(int(z["n_rays"]) for z in shards) == sp.shape[0]
    for k, ri in enumerate(whole.ray_interfaces):
        parts = [z["k%d_ray_index" % k] for z in shards if "k%d_ray_index" % k in z.files]
        got_idx = np.concatenate(parts)
        assert got_idx.size == ri.ray_index.size
        if k == 0:  # step 0 is in ray order in both
            assert np.array_equal(got_idx, ri.ray_index)
            got_p = np.concatenate([z["k0_intersection_point"] for z in shards])
            assert np.allclose(got_p, ri.intersection_point, rtol=1e-9, atol=1e-9)
        else:       # later steps: same rows, grouped by chunk
            o1 = np.lexsort((np.abs(got_idx),)); o2 = np.lexsort((np.abs(ri.ray_index),))
            got_l = np.concatenate([z["k%d_distance_traveled" % k] for z in shards if "k%d_distance_traveled" % k in z.files])
            assert np.array_equal(got_idx[o1], ri.ray_index[o2]) and np.allclose(got_l[o1], ri.distance_traveled[o2], rtol=1e-8, atol=1e-8)  # the chunks take the persistent kernel, the whole run the level kernels


def test_record_streaming_overlapped_copy_hands_over_the_same_chunks():
    """the copy of one chunk's records runs under the sink call of the chunk before and the kernels of the chunk after
    (two pinned stages): the sink sees what it sees when copy and sink run between the kernels, call for call"""
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    sl, origin = cases.coaxial_scene(4)
    sp, rays = cases.isotropic_rays(20011, 45, origin=tuple(origin), polarized=True)
    args = (10, 1e-7, 1e-5, -1, True, True, False)

    def run(overlap, workspace_gb):
        calls = []
        c = Context(0)
        try:
            c.set_option("workspace_gb", workspace_gb); c.set_option("stream_overlap", overlap)
            out = RayTracer2(sp, rays.copy(), sl, *args, seed=9, compat="matlab", ctx=c,
                             record_sink=lambda chunk, first, n, ris: calls.append((chunk, first, n, [dict(vars(r)) for r in ris])))
        finally:
            c.close()
        return out, calls
    out_a, a = run(1, 0.03)   # overlapped: a third of the workspace per stage
    out_b, b = run(0, 0.01)   # serial, same chunk size
    assert len(a) == len(b) >= 6 and np.allclose(out_a[1], out_b[1], rtol=1e-9, atol=1e-9)  # tallies: float atomics, order varies
    assert [x[:3] for x in a] == [x[:3] for x in b] and [x[0] for x in a] == list(range(len(a)))
    for (_, _, _, ra), (_, _, _, rb) in zip(a, b):
        assert len(ra) == len(rb)
        for da, db in zip(ra, rb):
            for f in da:
                assert np.array_equal(da[f], db[f], equal_nan=True), f


@pytest.mark.parametrize("params,n_in,n_med", [((.2, .8, .2, .3, .1), 1.5, 1.0), ((.05, .5, .3, .2, .05), np.inf, 1.5), ((0, 0, .25, .5, .1), np.inf, 1.5),
                                               ((.6, .3, .5, .1, .1), 1.3, 1.5)])
def test_unified_rays_get_their_own_launch(params, n_in, n_med):
    """tally-only runs put the rays that end on UNIFIED surfaces aside and give them a launch in which the whole wave
    draws facet normals together (unified_wave): same uniforms consumed, same tallies as the ray-by-ray loop - with the
    separate launch, without it, and in the oracle (two interactions per ray: nothing for rounding to grow on)."""
    from raytracer_amd.RayTracer2 import RayTracer2, get_context
    n = 70001                                                  # not a multiple of anything: lists end inside a wave
    sp, rays = cases.unifiedtest_rays(n)
    rng = np.random.default_rng(4)
    rays[:, 0:3] += 0.3 * rng.normal(size=(n, 3)); rays[:, 3:6] = rng.normal(size=(n, 3)); rays[:, 7] = 0.4; rays[:, 8] = -0.2; rays[:, 9] = 0.1
    sl = cases.unifiedtest(params=params, n_panel_in=n_in, n_med=n_med)
    ref = O.trace(sp, rays.copy(), sl, 10, 1e-9, 0, -1, False, True, False, seed=8, compat="matlab")
    c = get_context(0)
    try:
        tabs = []
        for d in (1, 0):
            c.set_option("defer_unified", d); c.set_option("level_min_rays", 0)
            tabs.append(RayTracer2(sp, rays.copy(), sl, 10, 1e-9, 0, -1, False, True, False, seed=8, compat="matlab")[1])
            live, scat = c.step_counts(10)
            assert np.array_equal(scat, ref.n_scat) and np.array_equal(live, ref.n_live)
    finally:
        c.set_option("defer_unified", 1); c.set_option("level_min_rays", 1 << 15)
    assert tabs[0].shape == ref.absorption_table.shape
    assert np.allclose(tabs[0], ref.absorption_table, rtol=1e-10, atol=1e-9)
    assert np.allclose(tabs[1], ref.absorption_table, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("sigma,maxiter", [(1.5, 0), (3.0, 0), (1e-4, 0), (1e-7, 0), (0.3, 3), (0.1, 1)])
def test_facet_screening_keeps_the_loop_exact(sigma, maxiter):
    """the single-precision screening of facet-normal tries must never drop a try the loop would accept: wide lobes (most
    tries fail the range test), needle-thin ones (the margin is relative to sigma), and loops cut short by unified_maxiter
    (the last try is kept whatever it gives) - step counts and tallies equal the oracle's ray-by-ray loop"""
    from raytracer_amd.RayTracer2 import RayTracer2, get_context
    n = 50003
    sp, rays = cases.unifiedtest_rays(n)
    rng = np.random.default_rng(5)
    rays[:, 0:3] += 0.4 * rng.normal(size=(n, 3)); rays[:, 3:6] = rng.normal(size=(n, 3)); rays[:, 7] = -0.3
    sl = cases.unifiedtest(params=(sigma, .5, .6, .2, .1), n_panel_in=1.5, n_med=1.0)
    ref = O.trace(sp, rays.copy(), sl, 10, 1e-9, 0, -1, False, True, False, seed=12, compat="matlab", unified_maxiter=maxiter)
    c = get_context(0)
    try:
        c.set_option("level_min_rays", 0)
        tab = RayTracer2(sp, rays.copy(), sl, 10, 1e-9, 0, -1, False, True, False, seed=12, compat="matlab", unified_maxiter=maxiter)[1]
        live, scat = c.step_counts(10)
    finally:
        c.set_option("level_min_rays", 1 << 15)
    assert np.array_equal(scat, ref.n_scat) and np.array_equal(live, ref.n_live)
    assert tab.shape == ref.absorption_table.shape and np.allclose(tab, ref.absorption_table, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("opts", [dict(level_steps=1000, level_min_rays=0), dict(level_steps=3, level_min_rays=0)])
def test_ray_splitting_runs_on_the_level_path(opts):
    """no-dice mode step by step: both children of every interaction go into the next step's queue (segments with room to
    grow); records, table and final rows equal the oracle's, also when the persistent kernel takes over after 3 steps"""
    sp, rays = cases.isotropic_rays(3000, 5, origin=(3, 0, 0), polarized=True)
    args = ([10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False)
    got = gpu_with(opts, sp, rays.copy(), cases.sphere_plane(), *args, compat="python-strict", return_final=True)
    ref = O.trace(sp, rays.copy(), cases.sphere_plane(), *args, compat="python-strict", final=True)
    compare_results(ref, got)
    assert np.allclose(ref.final, got.final, rtol=1e-9, atol=1e-9, equal_nan=True)
    sp, rays = cases.isotropic_rays(4000, 21, origin=(1, 2, 3), polarized=True)
    args = (8, 1e-7, np.array([2e-2, 2e-2]), 0.5, True, False, False)
    got = gpu_with(opts, sp, rays.copy(), cases.mixed_geometry(), *args, seed=32, compat="matlab")
    ref = O.trace(sp, rays.copy(), cases.mixed_geometry(), *args, seed=32, compat="matlab")
    compare_results(ref, got, growth=3.0)
    # tallies only
    got = gpu_with(opts, sp, rays.copy(), cases.mixed_geometry(), 8, 1e-7, np.array([2e-2, 2e-2]), 0.5, False, False, False, seed=32, compat="matlab")
    assert np.allclose(ref.absorption_table, got.absorption_table, rtol=1e-7, atol=1e-7)
