"""LED-to-camera matching on the device (rt_match_endpoints) against the oracle's literal restatement of
python/BubbleImageCreator.py:278-357: matched camera ray and distance bit-exact, intensities to 1e-12 (one exp and two
acos per pair), pixel sums to 1e-9 (atomics add in another order than the reference's LED order)."""
import numpy as np
import pytest

from oracle import oracle as O
import matchcases as M

pytestmark = pytest.mark.gpu


def _check(cam_rows, led_rows, surfaces, **kw):
    """octree as deep as the density asks for, two levels at most, and the plain uniform grid: all equal to the oracle"""
    from raytracer_amd.BubbleImageCreator import match_leds_to_camera
    from raytracer_amd.RayTracer2 import get_context
    okw = {}
    if "max_dist" in kw:
        okw["max_dist"] = kw["max_dist"]
    pix0, lc0, ld0, la0 = O.match_endpoints(M.final_rays_of(cam_rows), M.final_rays_of(led_rows), surfaces, **okw)
    c = get_context(0)
    try:
        for lev in (10, 2, 0):
            c.set_option("match_max_level", lev)
            pix, lc, ld, la = match_leds_to_camera(cam_rows, led_rows, surfaces, details=True, ctx=c, **kw)
            assert np.array_equal(lc, lc0), lev
            assert np.array_equal(ld, ld0), lev
            assert np.allclose(la, la0, rtol=1e-12, atol=0), lev
            assert np.allclose(pix, pix0, rtol=1e-9, atol=1e-300), lev
    finally:
        c.set_option("match_max_level", 10)
    return pix, lc


def test_match_dense_clusters():
    """many end points per coarse cell (the octree goes several levels deep), some LED rays far from every camera ray"""
    rng = np.random.default_rng(21)
    cam = M.cloud(20000, rng, [1, 2], extent=3.0)
    cam[:, 2] = 0.1 * np.sin(cam[:, 0])                         # a thin sheet, as end points on a surface are
    led = M.cloud(3000, rng, [1, 2], extent=9.0); led[:, 2] *= 0.05
    _check(cam, led, [1, 2], max_dist=5.0)
    _check(cam, led, [2], max_dist=0.7)
    cam2 = M.cloud(6000, rng, [4], extent=0.01, lattice=0.001)  # everything within one leaf cell of the deepest level
    led2 = M.cloud(500, rng, [4], extent=0.012, lattice=0.001)
    _check(cam2, led2, [4], max_dist=5.0)


@pytest.mark.parametrize("seed,n_cam,n_led,max_dist", [(1, 3000, 2000, 5.0), (2, 500, 4000, 1.5), (3, 4000, 300, 40.0), (4, 1, 50, 100.0)])
def test_match_random_clouds(seed, n_cam, n_led, max_dist):
    rng = np.random.default_rng(seed)
    cam = M.cloud(n_cam, rng, [2, 5, 9, 11]); led = M.cloud(n_led, rng, [2, 5, 9, 11, 30])
    led[::13, 19] = 0; led[5, 0] = np.nan                      # rays that never hit anything, broken end points: no part
    if n_cam > 20:
        cam[::11, 19] = 0; cam[7, 2] = np.inf
    else:
        cam[:, 16] = 9 * cam[:, 17]
    pix, lc = _check(cam, led, [5, 9, 30], max_dist=max_dist)
    assert (lc >= 0).sum() > 0


def test_match_ties_and_zero_distances():
    """end points on a coarse lattice: many equal distances (lowest camera row wins) and LED rays sitting exactly on
    camera rays (the reference's distance-0 sentinel: the next camera ray on the surface replaces such a match)"""
    rng = np.random.default_rng(7)
    cam = M.cloud(1500, rng, [1, 2], extent=6.0, lattice=1.0); led = M.cloud(1500, rng, [1, 2], extent=6.0, lattice=1.0)
    pix, lc = _check(cam, led, [1, 2], max_dist=2.5)
    d0 = np.linalg.norm(cam[np.maximum(lc, 0), 0:3] - led[:, 0:3], axis=1)
    assert ((lc >= 0) & (d0 == 0)).any() and ((lc >= 0) & (d0 > 0)).any()
    # every camera ray on the LED's spot: the last one keeps it at distance 0
    cam2 = cam[:40].copy(); cam2[:, 0:3] = [1, 2, 3]; cam2[:, 16] = cam2[:, 17]
    led2 = led[:10].copy(); led2[:, 0:3] = [1, 2, 3]; led2[:, 16] = led2[:, 17]
    pix, lc = _check(cam2, led2, [1])
    assert np.all(lc == 39)
    # cell borders: points at multiples of max_dist, negative coordinates included
    cam3 = M.cloud(800, rng, [1], extent=10.0, lattice=2.5); led3 = M.cloud(800, rng, [1], extent=10.0, lattice=1.25)
    _check(cam3, led3, [1], max_dist=2.5)


def test_match_empty_sets():
    rng = np.random.default_rng(9)
    cam = M.cloud(100, rng, [1]); led = M.cloud(80, rng, [1])
    pix, lc = _check(cam, led[:0], [1])
    assert pix.shape == (100,) and not pix.any() and lc.shape == (0,)
    pix, lc = _check(cam[:0], led, [1])
    assert pix.shape == (0,) and np.all(lc == -1)
    pix, lc = _check(cam, led, [])
    assert not pix.any() and np.all(lc == -1)
    pix, lc = _check(cam, led, [2])                            # nobody ends on the listed surface
    assert not pix.any() and np.all(lc == -1)
    from raytracer_amd.BubbleImageCreator import match_leds_to_camera
    from raytracer_amd._lib import RayTracerLibraryError
    with pytest.raises(RayTracerLibraryError):
        match_leds_to_camera(cam, led, [1], max_dist=0.0)
    with pytest.raises(RayTracerLibraryError):
        match_leds_to_camera(cam, led, list(range(65)))


def test_match_device_tensors_and_scale():
    """rows already in HBM (torch tensors), 2e6 x 2e6 end points: same numbers as through the host on a slice, and at
    full size the properties the rule guarantees - matched pairs closer than max_dist, on the same surface, no camera
    ray on that surface any nearer (checked by brute force for a sample), every LED contribution landing in one pixel."""
    import torch
    from raytracer_amd.BubbleImageCreator import match_leds_to_camera
    rng = np.random.default_rng(11)
    n = 2_000_000
    cam = M.cloud(n, rng, [3, 4, 5], extent=60.0); led = M.cloud(n, rng, [3, 4, 5], extent=60.0)
    tc = torch.from_numpy(cam).cuda(); tl = torch.from_numpy(led).cuda()
    pix, lc, ld, la = match_leds_to_camera(tc, tl, [3, 4], max_dist=1.0, details=True)
    pix, lc, ld, la = pix.cpu().numpy(), lc.cpu().numpy(), ld.cpu().numpy(), la.cpu().numpy()
    pix_h, lc_h, ld_h, la_h = match_leds_to_camera(cam, led, [3, 4], max_dist=1.0, details=True)
    assert np.array_equal(lc, lc_h) and np.array_equal(ld, ld_h) and np.array_equal(la, la_h) and np.allclose(pix, pix_h, rtol=1e-12)
    m = lc >= 0
    assert m.sum() > 1000
    assert np.all(ld[m] < 1.0) and np.all(np.abs(cam[lc[m], 16]) == np.abs(led[m, 16])) and np.all(np.isin(np.abs(led[m, 16]), [3, 4]))
    assert np.array_equal(ld[m], np.sqrt(((cam[lc[m], 0] - led[m, 0]) ** 2 + (cam[lc[m], 1] - led[m, 1]) ** 2) + (cam[lc[m], 2] - led[m, 2]) ** 2))
    assert np.isclose(pix.sum(), la.sum(), rtol=1e-10)
    tot = np.zeros(n); np.add.at(tot, lc[m], la[m])
    assert np.allclose(pix, tot, rtol=1e-10, atol=1e-300)
    for l in rng.choice(n, 40, replace=False):
        on = np.nonzero(np.abs(cam[:, 16]) == abs(led[l, 16]))[0]
        d = np.sqrt(((cam[on, 0] - led[l, 0]) ** 2 + (cam[on, 1] - led[l, 1]) ** 2) + (cam[on, 2] - led[l, 2]) ** 2)
        if abs(led[l, 16]) in (3, 4) and d.min() < 1.0:
            assert lc[l] == on[np.argmin(d)] and ld[l] == d.min()
        else:
            assert lc[l] == -1


def test_bubble_image_pipeline():
    """python/BubbleImageCreator.py:125-357 in small on the jar geometry: camera trace, LED trace, end points, matching.
    The end points made in the kernels equal the reference's reduction over ray_interfaces (:134-165), and the image
    equals the oracle's matching of the same end points."""
    from raytracer_amd import geometries as G
    from raytracer_amd import finalRays as F
    from raytracer_amd.BubbleImageCreator import LIGHT_SURFACES, arbc_lights, create_bubble_image, light_surface_indices
    from raytracer_amd.RayTracer2 import RayTracer2
    specs = G.ArBCSpecs(cam_resolution=np.array([48, 64]), lights_nrays=300)
    sl = G.create_arbc_geometry(specs)
    cam = G.arbc_camera(specs)
    sp, rays = arbc_lights(specs, np.random.default_rng(3))
    # with the dimensions of CreateArBCGeometry.m the camera mostly sees the top cone: let it give light too
    lit = LIGHT_SURFACES + ('reflector/diffuser topcone',)
    out = create_bubble_image(sl, cam, sp, rays.copy(), seed=17, descriptions=lit)   # RayTracer2 normalises its `rays` argument in place
    assert out["image"].shape == (48, 64)
    fin_c, fin_l = out["final_cam"], out["final_led"]
    sf = light_surface_indices(sl, lit)
    assert out["light_surfaces"] == sf and len(sf) == 4
    pix0, lc0, _, _ = O.match_endpoints(M.final_rays_of(fin_c), M.final_rays_of(fin_l), sf)
    assert (lc0 >= 0).sum() > 50 and np.count_nonzero(pix0) > 20
    assert np.allclose(out["pixel_intensity"], pix0, rtol=1e-9, atol=1e-300)
    assert np.array_equal(out["image"][5, 7], out["pixel_intensity"][7 * 48 + 5])
    # the same with the final rows of both traces left in HBM (rt_set_final_rows_device): identical end points and image
    dev = create_bubble_image(sl, cam, sp, rays.copy(), seed=17, descriptions=lit, on_device=True)
    assert np.array_equal(dev["final_cam"].cpu().numpy(), fin_c, equal_nan=True)
    assert np.array_equal(dev["final_led"].cpu().numpy(), fin_l, equal_nan=True)
    assert np.allclose(dev["pixel_intensity"], out["pixel_intensity"], rtol=1e-12, atol=1e-300)
    # LED end points: kernel reduction == reference reduction over the records of the same run
    ri, _, _, fin = RayTracer2(sp, rays.copy(), sl, 18, 1e-5, [1e-5, 100], seed=18, return_final=True)
    assert fin.shape == fin_l.shape
    a = F.from_rows(fin); b = O.final_rays(ri, sp.shape[0]); c = F.from_interfaces(ri, sp.shape[0])
    for f in ("incoming_ray", "intersection_point", "surface_normal", "ray_index", "surface_index"):
        assert np.array_equal(getattr(b, f), getattr(c, f)), f
        assert np.array_equal(getattr(a, f)[b.found], getattr(b, f)[b.found]), f
    assert np.array_equal(a.found, b.found)
