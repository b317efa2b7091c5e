"""Drop-in for the reference's `createGeometry` module: createGeometry(gs) -> [surface_list, rays, ray_startingpoints,
pixels] (python/createGeometry.py:61, 1242).  The reference file is a half-finished port ("STOP CONVERTING SURFACES
HERE", :631; Python `and`/`^` on arrays, :293, :400) and does not run; the geometry it is a port of,
MATLAB/Examples/CreateExamples/CreateArBCGeometry.m, is what geometries.create_arbc_geometry builds.  This module puts
the pieces together under the reference's names and return layout:

    surface_list          the jar geometry (39 surfaces, + the bubble sphere of :1097-1110 when gs.bubble_present)
    rays[0]               camera rays [n_pix, 10] = [direction, 0 0 1, 1 0 0 0]              (:1113-1118)
    ray_startingpoints[0] camera position repeated n_pix times                              (:1122)
    pixels[0]             (i, j) of every camera ray, 1-based                                (:1119)
    rays[1], ray_startingpoints[1]  the LED rays of :1125-1220 (BubbleImageCreator.arbc_lights)

`gs` is a geospecs.geospecs (or anything with those attributes).  The viewport position is derived as :222-223 do:
vp_s = (vp_focuslen - vp_nip_top) sin(vp_theta), vp_elev = (vp_focuslen - vp_nip_top) cos(vp_theta) + vp_focuselev.
Camera ray directions come from GenerateRaysFromCamera (the device function, as in the reference's call at :1113).
"""
import numpy as np

from . import GenerateRaysFromCamera
from . import geometries as _G
from .surface import surface


def arbc_specs(gs):
    """geospecs attribute bag -> geometries.ArBCSpecs (same names; vp_s / vp_elev derived, createGeometry.py:222-223)"""
    kw = {}
    spec = _G.ArBCSpecs()
    for name in vars(spec):
        if hasattr(gs, name) and getattr(gs, name) is not None:
            kw[name] = getattr(gs, name)
    if hasattr(gs, "vp_focuslen"):
        kw["vp_s"] = (gs.vp_focuslen - gs.vp_nip_top) * np.sin(gs.vp_theta)
        kw["vp_elev"] = (gs.vp_focuslen - gs.vp_nip_top) * np.cos(gs.vp_theta) + gs.vp_focuselev
    return _G.ArBCSpecs(**kw)


def createGeometry(gs, rng=None):
    spec = arbc_specs(gs)
    surface_list = _G.create_arbc_geometry(spec)
    if getattr(gs, "bubble_present", False):                                    # :1097-1110
        b = surface()
        b.description = 'bubble'; b.shape = 'sphere'; b.param_list = [np.asarray(gs.bubble_position, float), gs.bubble_radius]
        b.inbounds_function = lambda p: np.reshape(p[:, 2, :] > -500, (np.size(p, 0), -1))
        b.n_outside = gs.n_target; b.n_inside = gs.n_air; b.surface_type = 'normal'; b.absorption = 0
        surface_list.append(b)
    res = np.asarray(spec.cam_resolution, float)
    pitch = np.asarray(spec.cam_sensorsize, float) / res                        # :241
    vp_axis = np.array([0, -np.sin(spec.vp_theta), np.cos(spec.vp_theta)]); vp_center = np.array([0, -spec.vp_s, spec.vp_elev], dtype=float)
    raydirections, pixelmap = GenerateRaysFromCamera.GenerateRaysFromCamera(res, pitch, .5 * (1 + res), spec.cam_f, spec.cam_pitch + spec.vp_theta - (np.pi / 2),
                                                                           spec.cam_yaw, spec.cam_roll, spec.cam_barreld, spec.cam_lenstype)   # :1113
    n = np.size(raydirections, 0)
    rays = [np.hstack([raydirections, np.tile([0., 0, 1, 1, 0, 0, 0], (n, 1))])]                                               # :1118
    pixels = [pixelmap]                                                                                                          # :1119
    ray_startingpoints = [np.tile(vp_center + [spec.cam_x, 0, 0] + spec.cam_z * vp_axis + spec.cam_y * np.cross(vp_axis, [1, 0, 0]), (n, 1))]   # :1122
    from .BubbleImageCreator import arbc_lights
    led_sp, led_rays = arbc_lights(spec, rng)                                                                                   # :1125-1220
    ray_startingpoints.append(led_sp); rays.append(led_rays)
    return [surface_list, rays, ray_startingpoints, pixels]
