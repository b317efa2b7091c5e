"""Camera rays generated on the device (rt_generate_camera_rays / rt_trace_camera) against the numpy restatement of
MATLAB/Examples/ImagingExamples/GenerateRaysFromCamera.m in oracle/oracle.py ("parity unpinned": the reference's
numpy port of that file does not run), and the camera run of BubbleImageCreator.py:129 against the oracle tracer."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lens,dist", [("tan", []), ("theta", [0.015888108817219724, 0.04648232478103316]), ("sin", [-0.02]), ("fisheye?", [0.01, 0.002, 0.0003])])
def test_generate_rays_from_camera(lens, dist):
    from raytracer_amd.GenerateRaysFromCamera import GenerateRaysFromCamera
    res = [37, 53]; pitch = [4.8e-4 * 20, 5.1e-4 * 20]; ctr = [.5 * (1 + res[0]) + 0.3, .5 * (1 + res[1]) - 1.2]
    args = (res, pitch, ctr, 0.42, 0.1 - np.pi / 2, -0.07, 0.21, dist, lens)
    d, pix = GenerateRaysFromCamera(*args)
    d0, pix0 = O.generate_rays_from_camera(*args)
    assert np.array_equal(pix, pix0)
    assert np.allclose(d, d0, rtol=0, atol=1e-13)
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-13)
    # scalar pixel pitch, defaults
    d, _ = GenerateRaysFromCamera([5, 4], 0.01, [3, 2.5], 0.8)
    d0, _ = O.generate_rays_from_camera([5, 4], 0.01, [3, 2.5], 0.8)
    assert np.allclose(d, d0, atol=1e-13)
    with pytest.raises(ValueError):
        GenerateRaysFromCamera([5, 4, 3], 0.01, [3, 2.5], 0.8)


def test_camera_run_on_the_jar_geometry():
    """SURVEY 8d config 3 in small: camera rays of CreateArBCGeometry.m through the jar, K=18, min_travel 1e-5,
    thresholds [1e-5, 100] (python/BubbleImageCreator.py:129); final row per pixel = last record per ray id
    (BubbleImageCreator.py:149-161)."""
    from raytracer_amd import geometries as G
    from raytracer_amd.GenerateRaysFromCamera import camera_rays
    from raytracer_amd.RayTracer2 import trace_camera
    specs = G.ArBCSpecs(cam_resolution=np.array([96, 128]), cam_sensorsize=np.array([.1, .1]))
    sl = G.create_arbc_geometry(specs)
    cam = G.arbc_camera(specs)
    sp, rays, pix = camera_rays(cam)
    assert sp.shape[0] == 96 * 128 and np.allclose(sp, sp[0])
    d0, pix0 = O.generate_rays_from_camera(specs.cam_resolution, specs.cam_sensorsize / specs.cam_resolution, .5 * (1 + specs.cam_resolution), specs.cam_f,
                                           specs.cam_pitch + specs.vp_theta - np.pi / 2, specs.cam_yaw, specs.cam_roll, [specs.cam_barreld], specs.cam_lenstype)
    assert np.allclose(rays[:, 0:3], d0, atol=1e-13) and np.array_equal(pix, pix0)
    table, final, ris = trace_camera(cam, sl, 18, 1e-5, [1e-5, 100], [], True, True, compat="matlab", seed=17)
    ref = O.trace(sp, rays.copy(), sl, 18, 1e-5, np.array([1e-5, 100.]), [], True, True, False, compat="matlab", seed=17, final=True)
    assert table.shape == ref.absorption_table.shape and np.allclose(table, ref.absorption_table, rtol=1e-8, atol=1e-8)
    assert len(ris) == len(ref.ray_interfaces)
    for a, b in zip(ref.ray_interfaces, ris):
        assert np.array_equal(a.ray_index, b.ray_index) and np.array_equal(a.surface_index, b.surface_index)
        assert np.allclose(a.intersection_point, b.intersection_point, rtol=1e-8, atol=1e-8)
    assert np.allclose(ref.final, final, rtol=1e-6, atol=1e-6, equal_nan=True)
    # the image: most pixels end on something, and the last record of a pixel is what the final row holds
    last = {}
    for r in ris:
        for k, i in enumerate(np.abs(r.ray_index).astype(np.int64)):
            last[i] = np.asarray(r.intersection_point)[k]
    idx = np.array(sorted(last)); pts = np.array([last[i] for i in idx])
    assert np.allclose(final[idx - 1, 0:3], pts, rtol=1e-12, atol=1e-12)   # compat='matlab': ray_index is 1-based


def test_final_rows_device_buffer_too_small():
    """rt_set_final_rows_device: a buffer with fewer rows than rays is refused before anything runs, and the context is
    usable afterwards"""
    import torch
    from raytracer_amd import geometries as G
    from raytracer_amd._lib import RayTracerLibraryError
    from raytracer_amd.RayTracer2 import Context, trace_camera
    import ctypes as C
    specs = G.ArBCSpecs(cam_resolution=np.array([40, 50]))
    sl = G.create_arbc_geometry(specs); cam = G.arbc_camera(specs)
    c = Context(0)
    try:
        small = torch.zeros((100, 32), dtype=torch.float64, device="cuda:0")
        c._check(c.L.rt_set_final_rows_device(c.h, C.c_void_p(small.data_ptr()), 100))
        with pytest.raises(RayTracerLibraryError):
            trace_camera(cam, sl, 18, 1e-5, [1e-5, 100], ctx=c, seed=1)
        c.final_rows_host()
        _, host_rows, _ = trace_camera(cam, sl, 18, 1e-5, [1e-5, 100], ctx=c, seed=1)
        _, dev_rows, _ = trace_camera(cam, sl, 18, 1e-5, [1e-5, 100], ctx=c, seed=1, device_rows=True)
        assert np.array_equal(dev_rows.cpu().numpy(), host_rows, equal_nan=True)
    finally:
        c.close()
