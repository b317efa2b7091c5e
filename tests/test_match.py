"""Image formation, CPU side: the oracle's literal restatement of python/BubbleImageCreator.py:278-357 against values
worked out by hand, the host reduction finalRays.from_interfaces against the oracle's restatement of :134-165, and the
LED ray generator.  ("parity unpinned": those lines of the reference do not run; they are followed as text.)"""
import math

import numpy as np

from oracle import oracle as O
import matchcases as M


def test_oracle_match_by_hand():
    # surface 3: cameras 0,1,2 ; surface 4: camera 3.  LED 0 sits exactly on camera 0 (distance 0 -> the "unset"
    # sentinel lets camera 1 replace it, then camera 2 at the same distance does not win the tie); LED 1 is nearest to
    # camera 2; LED 2 on surface 4 is further than max_dist from camera 3; LED 3 sits exactly on camera 3, the last
    # camera of its surface, and keeps it at distance 0.
    cam_p = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [10, 10, 10.]])
    led_p = np.array([[0, 0, 0], [-1.5, 0, 0], [10, 10, 16], [10, 10, 10.]])
    nrm = np.tile([0, 0, 1.], (4, 1))
    cam_d = M.unit(np.array([[0, 0, -1], [0, 1, -1], [1, 0, -1], [0, 0, -1.]]))
    led_d = M.unit(np.array([[0, 1, -1], [0, 0, -1], [0, 0, -1], [1, 1, -1.]]))
    cam = M.final_rays_of(M.rows_from(cam_p, cam_d, nrm, np.array([3, -3, 3, 4])))
    led = M.final_rays_of(M.rows_from(led_p, led_d, nrm, np.array([-3, 3, 4, -4]), s0=np.array([1.0, 0.5, 1.0, 2.0])))
    pix, lc, ld, la = O.match_endpoints(cam, led, [3, 4])
    assert lc.tolist() == [1, 2, -1, 3]
    assert ld.tolist() == [1.0, 0.5, 0.0, 0.0]

    def add(s0, ang, ci, dist, li):
        return s0 * math.exp(-(ang ** 2) / (2 * 10 * math.pi / 180) - ci ** 2 / (2 * 45 * math.pi / 180) - dist ** 2 / (2 * 2 ** 2) - li ** 2 / (2 * 45 * math.pi / 180))
    q = math.pi / 4
    e0 = add(1.0, 0.0, q, 1.0, q)                      # LED 0 -> camera 1: same direction, both 45 degrees off the normal
    e1 = add(0.5, q, q, 0.5, 0.0)                      # LED 1 -> camera 2
    e3 = add(2.0, math.acos(1 / math.sqrt(3)), 0.0, 0.0, math.acos(1 / math.sqrt(3)))
    assert np.allclose(la, [e0, e1, 0.0, e3], rtol=1e-13, atol=0)
    assert np.allclose(pix, [0.0, e0, e1, e3], rtol=1e-13, atol=0)
    # an unlisted surface gives nothing
    pix, lc, _, _ = O.match_endpoints(cam, led, [4])
    assert lc.tolist() == [-1, -1, -1, 3] and pix[:3].tolist() == [0, 0, 0]


def _fake_interfaces(rng, n_rays, n_steps, base):
    out = []
    for k in range(n_steps):
        m = int(rng.integers(0, 2 * n_rays))
        ri = O.FinalRays()
        ids = rng.integers(base, base + n_rays, m).astype(float) * rng.choice([1, 1, 1, -1], m)   # repeats (ray splitting), reflected ones negative
        ri.ray_index = ids
        ri.incoming_ray = rng.normal(size=(m, 10)); ri.intersection_point = rng.normal(size=(m, 3)); ri.surface_normal = rng.normal(size=(m, 3))
        ri.surface_index = rng.integers(-5, 6, m).astype(float)
        out.append(ri)
    return out


def test_final_rays_host_reduction():
    from raytracer_amd import finalRays as F
    rng = np.random.default_rng(5)
    for base in (0, 1):
        for n_rays, n_steps in ((1, 1), (17, 4), (200, 7)):
            ris = _fake_interfaces(rng, n_rays, n_steps, base)
            a = F.from_interfaces(ris, n_rays, index_base=base); b = O.final_rays(ris, n_rays, index_base=base)
            for f in ("incoming_ray", "intersection_point", "surface_normal", "ray_index", "surface_index", "found", "cos_incident", "theta_incident"):
                assert np.array_equal(getattr(a, f), getattr(b, f), equal_nan=True), (f, base, n_rays)
    # nothing to reduce
    a = F.from_interfaces([], 3)
    assert not a.found.any() and a.incoming_ray.shape == (3, 10)


def test_final_rays_from_rows():
    from raytracer_amd import finalRays as F
    rng = np.random.default_rng(2)
    rows = M.cloud(50, rng, [1, 2])
    rows[::7, 19] = 0                                      # rays that never interacted
    fr = F.from_rows(rows)
    hit = rows[:, 19] > 0
    assert np.array_equal(fr.found, hit)
    assert np.array_equal(fr.intersection_point[hit], rows[hit, 0:3]) and not fr.intersection_point[~hit].any()
    assert np.array_equal(fr.surface_index[hit], rows[hit, 16])
    assert np.allclose(fr.cos_incident[hit], -np.sum(rows[hit, 3:6] * rows[hit, 13:16], axis=1))
    assert np.all(fr.cos_incident[hit] >= 0)


def test_lights():
    from raytracer_amd import geometries as G
    from raytracer_amd.BubbleImageCreator import arbc_lights, light_surface_indices
    g = G.ArBCSpecs(lights_nrays=400)
    sp, rays = arbc_lights(g, np.random.default_rng(0))
    n = 3 * g.lights_number * g.lights_nrays
    assert sp.shape == (n, 3) and rays.shape == (n, 10)
    assert np.allclose(np.linalg.norm(rays[:, 0:3], axis=1), 1)
    assert len(np.unique(np.round(sp, 9), axis=0)) == 3 * g.lights_number
    vp_axis = np.array([0, -np.sin(g.vp_theta), np.cos(g.vp_theta)]); vp_center = np.array([0, -g.vp_s, g.vp_elev])
    for c in range(1, 4):
        a = 2 * np.pi / 3 * c
        rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        axis = rot @ vp_axis; ctr = rot @ vp_center
        blk = slice((c - 1) * n // 3, c * n // 3)
        rel = sp[blk] - ctr
        assert np.allclose(rel @ axis, g.lights_height)                                   # ring plane
        assert np.allclose(np.linalg.norm(rel - np.outer(rel @ axis, axis), axis=1), g.lights_radius)   # ring radius
        mu = rays[blk, 0:3] @ (-axis)
        assert mu.min() >= np.cos(g.lens_angle / 2) - 1e-12 and mu.max() <= 1 + 1e-12    # inside the cone, facing the chamber
        assert abs(mu.mean() - (1 + np.cos(g.lens_angle / 2)) / 2) < 0.02                 # uniform in cos
    assert np.all(rays[:, 6:10] == [1, 0, 0, 0])
    sl = G.create_arbc_geometry(g)
    idx = light_surface_indices(sl)
    assert [sl[i].description for i in idx] == ['reflector/diffuser', 'reflector/diffuser cone', 'reflector/diffuser botcone']
    assert light_surface_indices(sl, compat="matlab") == [i + 1 for i in idx]
