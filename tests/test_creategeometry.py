"""createGeometry / geospecs drop-in modules (python/createGeometry.py:61,1242, python/geospecs.py:5-103): return layout,
shapes and row order on the CPU (the camera directions come from the oracle's restatement of GenerateRaysFromCamera.m
there: the product computes them on the device), and on the GPU the module as it ships."""
import numpy as np
import pytest

from oracle import oracle as O


def _check_layout(out, gs):
    surface_list, rays, ray_startingpoints, pixels = out                               # createGeometry.py:1242
    npix = int(gs.cam_resolution[0]) * int(gs.cam_resolution[1]); nled = 3 * gs.lights_number * gs.lights_nrays
    assert len(surface_list) == 39 + (1 if gs.bubble_present else 0)
    assert len(rays) == 2 and len(ray_startingpoints) == 2 and len(pixels) == 1
    assert rays[0].shape == (npix, 10) and ray_startingpoints[0].shape == (npix, 3) and pixels[0].shape == (npix, 2)
    assert rays[1].shape == (nled, 10) and ray_startingpoints[1].shape == (nled, 3)
    # rows in the order of i_pix(:), j_pix(:) (GenerateRaysFromCamera.m:44-46): i runs fastest, both 1-based
    assert np.array_equal(pixels[0][:3], [[1, 1], [2, 1], [3, 1]]) and np.array_equal(pixels[0][-1], gs.cam_resolution)
    assert np.array_equal(rays[0][:, 3:10], np.tile([0, 0, 1, 1, 0, 0, 0], (npix, 1)))   # :1118
    assert np.abs(np.linalg.norm(rays[0][:, 0:3], axis=1) - 1).max() < 1e-12
    vp_s = (gs.vp_focuslen - gs.vp_nip_top) * np.sin(gs.vp_theta); vp_elev = (gs.vp_focuslen - gs.vp_nip_top) * np.cos(gs.vp_theta) + gs.vp_focuselev   # :222-223
    vp_axis = np.array([0, -np.sin(gs.vp_theta), np.cos(gs.vp_theta)])
    assert np.allclose(ray_startingpoints[0], np.array([0, -vp_s, vp_elev]) + gs.cam_z * vp_axis)   # :1122
    assert np.array_equal(rays[1][:, 6:10], np.tile([1, 0, 0, 0], (nled, 1)))            # :1220
    assert np.unique(ray_startingpoints[1], axis=0).shape[0] == 3 * gs.lights_number     # one point per LED
    # the camera looks into the chamber: the central pixel's ray runs down the viewport axis
    i = (int(gs.cam_resolution[1]) // 2) * int(gs.cam_resolution[0]) + int(gs.cam_resolution[0]) // 2
    assert np.dot(rays[0][i, 0:3], -vp_axis) > 0.99
    s = surface_list[0]
    for name in ("description", "shape", "param_list", "inbounds_function", "n_outside", "n_inside", "surface_type", "absorption"):   # surface.py:51-62
        assert hasattr(s, name)
    if gs.bubble_present:
        assert surface_list[-1].description == 'bubble' and surface_list[-1].n_inside == gs.n_air


def _specs():
    from raytracer_amd import geospecs as GS
    gs = GS.geospecs()
    gs.cam_resolution = np.array([12, 16]); gs.bubble_present = True; gs.bubble_position = np.array([0, 0, -5]); gs.lights_nrays = 7
    return gs


def test_geospecs_defaults_are_the_reference_ones():
    from raytracer_amd import geospecs as GS
    gs = GS.geospecs()
    assert gs.n_target == 1.17 and gs.n_jar == 1.4512 and gs.n_pressurewall == np.inf and gs.ojar_cylrad == 7.5 and gs.ijar_axrad == 13
    assert gs.vp_theta == 22 * np.pi / 180 and gs.vp_focuslen == 33.5 and gs.vp_focuselev == -10 and gs.cam_lenstype == "theta"
    assert np.array_equal(gs.cam_resolution, [480, 640]) and gs.lights_number == 5 and gs.lens_angle == (2 / 3) * np.pi and gs.bubble_present is False
    assert len(vars(gs)) == 67   # as many as python/geospecs.py sets (compared attribute by attribute with the imported reference when this was written)


def test_creategeometry_layout_cpu(monkeypatch):
    from raytracer_amd import createGeometry as CG
    from raytracer_amd import table as T
    monkeypatch.setattr(CG.GenerateRaysFromCamera, "GenerateRaysFromCamera", lambda *a: list(O.generate_rays_from_camera(*a)))
    gs = _specs()
    out = CG.createGeometry(gs, np.random.default_rng(0))
    _check_layout(out, gs)
    tab, atoms, code = T.compile_surfaces(out[0])          # every predicate, the bubble's np.size(p, 0) form included, compiles
    assert tab.shape == (40, 32)


@pytest.mark.gpu
def test_creategeometry_on_gpu_and_trace():
    from raytracer_amd import createGeometry as CG
    from raytracer_amd.RayTracer2 import RayTracer2
    gs = _specs()
    surface_list, rays, sp, pixels = CG.createGeometry(gs, np.random.default_rng(0))
    _check_layout([surface_list, rays, sp, pixels], gs)
    d_ref, pix_ref = O.generate_rays_from_camera(gs.cam_resolution, gs.cam_sensorsize / gs.cam_resolution, .5 * (1 + gs.cam_resolution), gs.cam_f,
                                                 gs.cam_pitch + gs.vp_theta - np.pi / 2, gs.cam_yaw, gs.cam_roll, gs.cam_barreld, gs.cam_lenstype)
    assert np.array_equal(pixels[0], pix_ref) and np.abs(rays[0][:, 0:3] - d_ref).max() < 1e-12
    # the call of python/BubbleImageCreator.py:129 on what createGeometry returned
    ri, tab, _ = RayTracer2(sp[0], rays[0], surface_list, 18, 1e-5, [1e-5, 100], seed=3)
    ref = O.trace(sp[0], rays[0], surface_list, 18, 1e-5, [1e-5, 100], seed=3)
    assert len(ri) == len(ref.ray_interfaces) and np.allclose(tab, ref.absorption_table, rtol=1e-7, atol=1e-7)
