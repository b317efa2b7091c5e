"""Drop-in for the reference's `geospecs` module (python/geospecs.py:3-104): an attribute bag of chamber dimensions,
refractive indices, camera and light parameters with the reference's names and default values, handed to
createGeometry.createGeometry(gs).  The values are input data; they are tabulated here instead of assigned one by one.
"""
import numpy as np

# name -> default; grouped as python/geospecs.py groups them
_DEFAULTS = (
    # bubble (:7-9)
    ("bubble_present", False), ("bubble_radius", 1), ("bubble_position", lambda: np.zeros(3, dtype=int)),
    # indices of refraction (:12-17)
    ("n_target", 1.17), ("n_jar", 1.4512), ("n_hydraulic", 1.22), ("n_pressurewindow", 1.7569), ("n_pressurewall", np.inf), ("n_air", 1.00),
    # outer and inner jar, cm (:21-34)
    ("ojar_thick", .25), ("ojar_cylrad", 7.5), ("ojar_axrad", 15), ("ojar_knucklerad", 2.5), ("ojar_cyllength", 40), ("ojar_elevation", 20),
    ("ijar_thick", .25), ("ijar_cylrad", 6.5), ("ijar_axrad", 13), ("ijar_knucklerad", 2.5), ("ijar_cyllength", 20), ("ijar_elevation", 0),
    # viewport position and dimensions (:37-53)
    ("vp_theta", 22 * np.pi / 180), ("vp_focuselev", -10), ("vp_focuslen", 33.5),
    ("vp_win_rad", 1.73 * .5 * 2.54), ("vp_air_rad", 1.25 * .5 * 2.54), ("vp_can_rad", 2 * 2.54), ("vp_can_wall", .125 * 2.54),
    ("vp_flange_rad", 3.375 * 2.54), ("vp_nip_rad", 1.75 * .5 * 2.54), ("vp_win_thick", .25 * 2.54), ("vp_nip_top", .5), ("vp_can_OAL", 6 * 2.54),
    ("vp_flange_thick", lambda: np.full((5,), 1) * 2.54 * .5),
    # reflector / diffuser, its cones (:56-70), pressure vessel (:73-77)
    ("rd_rad", 12), ("rd_top", 100), ("rd_bot", 0), ("rdcone_top", 120), ("rdcone_toprad", 16),
    ("rdtopcone_apex", 150), ("rdtopcone_rad", 10.5), ("rdtopcone_bot", -20), ("rdbotcone_apex", -15.2), ("rdbotcone_rad", 10.5), ("rdbotcone_bot", -20),
    ("pv_bot", -20), ("pv_top", +100), ("pv_rad", 30), ("pv_thick", 1), ("pv_axrad", 15),
    # camera (:85-97)
    ("cam_x", 0), ("cam_y", 0), ("cam_z", 5), ("cam_f", .8), ("cam_barreld", 0), ("cam_lenstype", "theta"),
    ("cam_sensorsize", lambda: np.full((2,), .1)), ("cam_resolution", lambda: np.array([480, 640])), ("deres", None),
    ("cam_pitch", 0), ("cam_yaw", 0), ("cam_roll", 0),
    # lights (:100-104)
    ("lights_number", 5), ("lights_height", -8.5), ("lights_radius", 7.5), ("lights_nrays", 100), ("lens_angle", (2 / 3) * np.pi),
)


class geospecs(object):
    def __init__(self):
        for name, v in _DEFAULTS:
            setattr(self, name, v() if callable(v) else v)
