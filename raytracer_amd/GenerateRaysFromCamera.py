"""Drop-in for the reference's GenerateRaysFromCamera module (python/GenerateRaysFromCamera.py, which does not run;
MATLAB/Examples/ImagingExamples/GenerateRaysFromCamera.m is the specification): the directions are computed on the
MI355X by the same device function that rt_trace_camera uses to generate camera rays in HBM.

    [ray_direction, pixels] = GenerateRaysFromCamera(resolution, pixel_pitch, pixel_center, focal_length,
                                                     pitch, yaw, roll, radial_distortion, lens_type)
"""
import ctypes as C

import numpy as np

from . import _lib


def make_camera(resolution, pixel_pitch, pixel_center, focal_length, pitch=0, yaw=0, roll=0, radial_distortion=(), lens_type="tan",
                position=(0.0, 0.0, 0.0), polref=(0.0, 0.0, 1.0), stokes=(1.0, 0.0, 0.0, 0.0)):
    """rt_camera from the reference's arguments; input checks of GenerateRaysFromCamera.m:29-37 (same message)."""
    def scalar(x):
        a = np.atleast_1d(np.asarray(x, float)).ravel()
        if a.size != 1:
            raise ValueError('impropper input to GenerateRaysFromCamera')
        return float(a[0])
    pp = np.atleast_1d(np.asarray(pixel_pitch, float)).ravel()
    rd = np.atleast_1d(np.asarray(radial_distortion if radial_distortion is not None else [], float)).ravel()
    if np.size(resolution) != 2 or np.size(pixel_center) != 2 or pp.size < 1 or pp.size > 2 or rd.size > 8:
        raise ValueError('impropper input to GenerateRaysFromCamera')
    if pp.size == 1:
        pp = np.array([pp[0], pp[0]])
    cam = _lib.Camera()
    cam.resolution[0], cam.resolution[1] = int(round(float(resolution[0]))), int(round(float(resolution[1])))
    cam.lens_type = _lib.LENS_TYPES.get(lens_type, 0)  # anything else: 'tan' (GenerateRaysFromCamera.m:65-66)
    cam.n_distortion = int(rd.size)
    cam.pixel_pitch[0], cam.pixel_pitch[1] = float(pp[0]), float(pp[1])
    cam.pixel_center[0], cam.pixel_center[1] = float(pixel_center[0]), float(pixel_center[1])
    cam.focal_length, cam.pitch, cam.yaw, cam.roll = scalar(focal_length), scalar(pitch), scalar(yaw), scalar(roll)
    for k in range(rd.size):
        cam.distortion[k] = float(rd[k])
    for k in range(3):
        cam.position[k] = float(position[k]); cam.polref[k] = float(polref[k])
    for k in range(4):
        cam.stokes[k] = float(stokes[k])
    return cam


def camera_rays(cam, ctx=None, device=0):
    """(start N x 3, rays N x 10, pixels N x 2) of every pixel, in the reference's row order"""
    from .RayTracer2 import get_context
    c = ctx if ctx is not None else get_context(device)
    n = int(cam.resolution[0]) * int(cam.resolution[1])
    sp = np.zeros((n, 3)); rays = np.zeros((n, 10)); pix = np.zeros((n, 2), np.int32)
    P = C.POINTER
    c._check(c.L.rt_generate_camera_rays(c.h, C.byref(cam), 0, n, sp.ctypes.data_as(P(C.c_double)), rays.ctypes.data_as(P(C.c_double)), pix.ctypes.data_as(P(C.c_int32))))
    return sp, rays, pix


def GenerateRaysFromCamera(resolution, pixel_pitch, pixel_center, focal_length, pitch=0, yaw=0, roll=0, radial_distortion=[], lens_type='tan'):
    cam = make_camera(resolution, pixel_pitch, pixel_center, focal_length, pitch, yaw, roll, radial_distortion, lens_type)
    _, rays, pix = camera_rays(cam)
    return [rays[:, 0:3].copy(), pix.astype(np.int64)]
