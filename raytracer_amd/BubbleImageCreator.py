"""Image formation of the reference's python/BubbleImageCreator.py as functions: trace the camera rays, trace the LED
rays, reduce both to their end points, let every LED ray light the camera pixel whose ray ends nearest to it on the
same reflector surface (:278-357, only present as MATLAB comments in the numpy port, which stops at :250).

The matching runs on the device (rt_match_endpoints, include/rt_mi355x.h): a uniform grid instead of the reference's
N_led x N_cam double loop, same answer.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import RayTracer2 as _rt

LIGHT_SURFACES = ('reflector/diffuser', 'reflector/diffuser cone', 'reflector/diffuser botcone')   # :282-283


def light_surface_indices(surface_list, descriptions=LIGHT_SURFACES, compat="python"):
    """surface numbers, in the convention of surface_index, of the surfaces that give off light (:278-283)"""
    base = 0 if (_rt.COMPAT[compat] & _lib.COMPAT_PY_INDEX) else 1
    return [i + base for i, s in enumerate(surface_list) if getattr(s, "description", None) in descriptions]


def match_params(max_dist=5.0, angle_sigma_deg=10.0, incidence_sigma_deg=45.0, dist_sigma=2.0, flags=0):
    """the constants of :321 and :338 (their units are the reference's: degrees enter un-squared)"""
    mp = _lib.MatchParams()
    mp.max_dist = float(max_dist); mp.angle_den = 2 * angle_sigma_deg * np.pi / 180
    mp.cam_incidence_den = 2 * incidence_sigma_deg * np.pi / 180; mp.led_incidence_den = 2 * incidence_sigma_deg * np.pi / 180
    mp.dist_den = 2 * float(dist_sigma) ** 2; mp.flags = int(flags)
    return mp


def match_leds_to_camera(cam_rows, led_rows, surfaces, max_dist=5.0, angle_sigma_deg=10.0, incidence_sigma_deg=45.0, dist_sigma=2.0,
                         details=False, ctx=None, device=0):
    """pixel_intensity[n_cam] from the final rows of the camera and LED traces.  numpy arrays go through the host; torch
    tensors on the GPU are used in place (outputs are torch tensors then).  details=True also returns, per LED ray,
    the matched camera row (-1 if none), its distance and the intensity added."""
    c = ctx if ctx is not None else _rt.get_context(device)
    sf = np.ascontiguousarray(surfaces, np.int32).ravel()
    on_dev = hasattr(cam_rows, "data_ptr")
    if on_dev != hasattr(led_rows, "data_ptr"):
        raise ValueError("cam_rows and led_rows must both be numpy arrays or both be GPU tensors")
    if on_dev:
        import torch
        if not (cam_rows.is_cuda and led_rows.is_cuda and cam_rows.dtype == torch.float64 and led_rows.dtype == torch.float64):
            raise ValueError("device rows must be float64 tensors on the GPU")
        cam = cam_rows.contiguous(); led = led_rows.contiguous()
        if cam.dim() != 2 or led.dim() != 2 or cam.shape[1] != _lib.FINAL_COLS or led.shape[1] != _lib.FINAL_COLS:
            raise ValueError("rows must be [n][%d]" % _lib.FINAL_COLS)
        n_cam, n_led = cam.shape[0], led.shape[0]
        pix = torch.zeros(n_cam, dtype=torch.float64, device=cam.device)
        lc = torch.zeros(n_led, dtype=torch.int64, device=cam.device); ld = torch.zeros(n_led, dtype=torch.float64, device=cam.device); la = torch.zeros_like(ld)
        torch.cuda.synchronize(cam.device)
        ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        flags = _lib.MATCH_DEVICE_POINTERS
    else:
        cam = np.ascontiguousarray(cam_rows, np.float64); led = np.ascontiguousarray(led_rows, np.float64)
        if cam.ndim != 2 or led.ndim != 2 or cam.shape[1] != _lib.FINAL_COLS or led.shape[1] != _lib.FINAL_COLS:
            raise ValueError("rows must be [n][%d]" % _lib.FINAL_COLS)
        n_cam, n_led = cam.shape[0], led.shape[0]
        pix = np.zeros(n_cam); lc = np.zeros(n_led, np.int64); ld = np.zeros(n_led); la = np.zeros(n_led)
        ptr = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        flags = 0
    mp = match_params(max_dist, angle_sigma_deg, incidence_sigma_deg, dist_sigma, flags)
    c._check(c.L.rt_match_endpoints(c.h, ptr(cam), n_cam, ptr(led), n_led, sf.ctypes.data_as(C.POINTER(C.c_int32)), sf.size, C.byref(mp),
                                    ptr(pix), ptr(lc), ptr(ld), ptr(la)))
    return (pix, lc, ld, la) if details else pix


def arbc_lights(geospecs, rng=None):
    """LED rays of the jar geometry as python/createGeometry.py:1125-1220 means them (those lines do not run; their MATLAB
    comments and intent are followed): lights_number LEDs on a ring of radius lights_radius around each of the three
    viewport positions (the viewport rotated by 120, 240, 360 degrees about z), lights_height along the viewport axis,
    each sending lights_nrays rays uniformly into the cone of full angle lens_angle that faces the chamber.  The
    port draws the angles with randn where uniform deviates are meant (:1203-1204, arccos of values outside [-1,1]);
    uniform ones are used.  Returns (ray_startingpoints [n,3], rays [n,10]) with n = 3 * lights_number * lights_nrays."""
    g = geospecs
    rng = rng if rng is not None else np.random.default_rng()
    vp_axis = np.array([0, -np.sin(g.vp_theta), np.cos(g.vp_theta)])
    vp_center = np.array([0, -g.vp_s, g.vp_elev], dtype=float)
    per_cam = int(g.lights_nrays) * int(g.lights_number)
    sp = np.zeros((3 * per_cam, 3)); rays = np.zeros((3 * per_cam, 10))
    y = np.array([0., 1., 0.])
    for c in range(1, 4):
        ca = (2 * np.pi / 3) * c
        rot = np.array([[np.cos(ca), -np.sin(ca), 0], [np.sin(ca), np.cos(ca), 0], [0, 0, 1]])
        c_point = rot @ vp_center; c_axis = rot @ vp_axis
        zp = c_axis / np.linalg.norm(c_axis)
        yp = y - np.dot(y, zp) * zp; yp /= np.linalg.norm(yp)
        xp = np.cross(yp, zp); xp /= np.linalg.norm(xp)
        P = np.column_stack([xp, yp, zp])
        rad0 = c_point - np.dot(c_point, c_axis) * c_axis / np.linalg.norm(c_axis) ** 2
        rad0 /= np.linalg.norm(rad0)
        start = per_cam * (c - 1)
        for n in range(1, int(g.lights_number) + 1):
            la = (2 * np.pi / g.lights_number) * n
            rz = np.array([[np.cos(la), -np.sin(la), 0], [np.sin(la), np.cos(la), 0], [0, 0, 1]])
            rad = (P @ rz @ np.linalg.inv(P)) @ rad0
            loc = c_point + g.lights_height * zp + g.lights_radius * rad
            sp[start + (n - 1) * g.lights_nrays: start + n * g.lights_nrays] = loc
        zq = -zp                                                               # facing the chamber (:1196)
        yq = y - np.dot(y, zq) * zq; yq /= np.linalg.norm(yq)
        xq = np.cross(yq, zq); xq /= np.linalg.norm(xq)
        th = 2 * np.pi * rng.random(per_cam)
        omc = (1 - np.cos(g.lens_angle / 2)) * rng.random(per_cam)
        ph = np.arccos(1 - omc)
        rays[start:start + per_cam, 0:3] = (np.sin(ph) * np.cos(th))[:, None] * xq + (np.sin(ph) * np.sin(th))[:, None] * yq + (1 - omc)[:, None] * zq
    rays[:, 3:6] = rng.random((3 * per_cam, 3)) - rng.random((3 * per_cam, 3))    # :1216 (rand - rand)
    rays[:, 6:10] = [1, 0, 0, 0]
    return sp, rays


def create_bubble_image(surface_list, cam, led_startingpoints, led_rays, max_scat=18, min_travel_len=1e-5, follow_thresh=(1e-5, 100),
                        descriptions=LIGHT_SURFACES, compat="python", seed=None, ctx=None, device=0, on_device=False, **match_kw):
    """The whole of python/BubbleImageCreator.py:125-357 for one camera: returns a dict with `image` (pixel_intensity
    shaped [resolution[0], resolution[1]], pixel (i, j) at [i-1, j-1]), the final rows of both traces and the tables.
    on_device=True keeps the final rows of both traces in HBM (torch tensors): only the pixel intensities cross PCIe."""
    c = ctx if ctx is not None else _rt.get_context(device)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    tab_c, fin_c, _ = _rt.trace_camera(cam, surface_list, max_scat, min_travel_len, follow_thresh, compat=compat, seed=seed, ctx=c,
                                       device_rows=on_device)                                                                          # :129
    out = _rt.RayTracer2(led_startingpoints, led_rays, surface_list, max_scat, min_travel_len, follow_thresh, full_output=False,
                         compat=compat, seed=seed + 1, ctx=c, return_final="device" if on_device else True)                             # :185
    fin_l = out[3]
    sf = light_surface_indices(surface_list, descriptions, compat)
    pix = match_leds_to_camera(fin_c, fin_l, sf, ctx=c, **match_kw)
    if on_device:
        pix = pix.cpu().numpy()
    res = (int(cam.resolution[0]), int(cam.resolution[1]))
    return {"image": pix.reshape(res, order="F"), "pixel_intensity": pix, "final_cam": fin_c, "final_led": fin_l,
            "table_cam": tab_c, "table_led": out[1], "light_surfaces": sf}
