"""Geometry builders that emit reference-style `surface` lists (python/surface.py:51-62) for the benchmark configs.

create_arbc_geometry : the 39-surface argon bubble chamber of MATLAB/Examples/CreateExamples/CreateArBCGeometry.m:50-629
    (4 cylinders, 4 spheres, 4 torus knuckles of the two quartz jars; viewport cylinders/planes; reflector cones and
    pressure-vessel heads as quadsurfaces) -- the "SBC/ArBC jar geometry" of SURVEY.md section 8d configs 3-5.  The
    half-ported python/createGeometry.py is not runnable; dimensions default to python/geospecs.py.
create_sbc_geometry  : the 11-surface xenon chamber of MATLAB/Examples/CreateExamples/CreateSBCGeometry.m:9-231, the
    geometry of MATLAB/Examples/RunExamples/RunSBCGeometry.m (bulk absorption + Rayleigh scattering in the target).
Predicates are ordinary numpy lambdas over p[N,3,M], exactly what the reference's drivers would write; the bounds
compiler (raytracer_amd/bounds.py) turns them into device programs.
"""
import numpy as np

from .surface import surface


def _mk(desc, shape, params, inb, n_out, n_in, stype="normal", absorption=0.0, **extra):
    s = surface()
    s.description = desc; s.shape = shape; s.param_list = params; s.inbounds_function = inb
    s.n_outside = n_out; s.n_inside = n_in; s.surface_type = stype; s.absorption = absorption
    for k, v in extra.items():
        setattr(s, k, v)
    return s


class ArBCSpecs(object):
    """defaults of CreateArBCGeometry.m:77-146 (same names)"""

    def __init__(self, **kw):
        self.n_target = 1.17; self.n_jar = 1.4512; self.n_hydraulic = 1.22; self.n_pressurewindow = 1.7569
        self.n_pressurewall = np.inf; self.n_air = 1.00
        self.ojar_thick = .25; self.ojar_cylrad = 7.5; self.ojar_axrad = 15; self.ojar_knucklerad = 2.5; self.ojar_cyllength = 40; self.ojar_elevation = 20
        self.ijar_thick = .25; self.ijar_cylrad = 6.5; self.ijar_axrad = 13; self.ijar_knucklerad = 2.5; self.ijar_cyllength = 20; self.ijar_elevation = 0
        self.vp_s = 10; self.vp_elev = 60; self.vp_win_rad = 1.73 * .5 * 2.54; self.vp_air_rad = 1.25 * .5 * 2.54; self.vp_can_rad = 2 * 2.54
        self.vp_can_wall = .125 * 2.54; self.vp_flange_rad = 3.375 * 2.54; self.vp_nip_rad = 1.75 * .5 * 2.54; self.vp_win_thick = .25 * 2.54
        self.vp_nip_top = .5; self.vp_theta = 6 * np.pi / 180; self.vp_can_OAL = 6 * 2.54; self.vp_flange_thick = np.array([1, 1, 1, 1, 1]) * 2.54 * .5
        self.rd_rad = 12; self.rd_top = 100; self.rd_bot = 0; self.rdcone_top = 120; self.rdcone_toprad = 16
        self.rdtopcone_apex = 150; self.rdtopcone_rad = 10.5; self.rdtopcone_bot = -20
        self.rdbotcone_apex = -15.2; self.rdbotcone_rad = 10.5; self.rdbotcone_bot = -20
        self.pv_bot = -20; self.pv_top = +100; self.pv_rad = 30; self.pv_thick = 1; self.pv_axrad = 15
        # extras for the full-MC configuration; inf / 'normal' reproduce the MATLAB file
        self.target_rayleigh = np.inf; self.target_abslength = np.inf
        self.reflector_type = "normal"; self.reflector_absorption = 1.0; self.reflector_unifiedparams = [0, 1, 0, 1, 0]
        self.reflector_n_inside = None
        self.reflector_all = False   # True: the strip cone, top cone and bottom cone get the reflector's optics too
        # camera, CreateArBCGeometry.m:135-146
        self.cam_x = 0; self.cam_y = 0; self.cam_z = 5; self.cam_f = .8; self.cam_barreld = 0; self.cam_lenstype = 'theta'
        self.cam_sensorsize = np.array([.1, .1]); self.cam_resolution = np.array([480, 640]); self.cam_pitch = 0; self.cam_yaw = 0; self.cam_roll = 0
        # lights, python/geospecs.py:100-104 (see BubbleImageCreator.arbc_lights)
        self.lights_number = 5; self.lights_height = -8.5; self.lights_radius = 7.5; self.lights_nrays = 100; self.lens_angle = (2 / 3) * np.pi
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("unknown geospec %r" % k)
            setattr(self, k, v)


def arbc_camera(geospecs=None):
    """The camera of CreateArBCGeometry.m:171, 632-637 as the rt_camera of rt_trace_camera: looks down the viewport axis
    from vp_center + cam_z * vp_axis; every ray is [direction, 0 0 1, 1 0 0 0]."""
    from .GenerateRaysFromCamera import make_camera
    g = geospecs if geospecs is not None else ArBCSpecs()
    vp_axis = np.array([0, -np.sin(g.vp_theta), np.cos(g.vp_theta)])
    vp_center = np.array([0, -g.vp_s, g.vp_elev], dtype=float)
    res = np.asarray(g.cam_resolution, float)
    pitch = np.asarray(g.cam_sensorsize, float) / res
    pos = vp_center + np.array([g.cam_x, 0, 0]) + g.cam_z * vp_axis + g.cam_y * np.cross(vp_axis, [1, 0, 0])
    return make_camera(res, pitch, .5 * (1 + res), g.cam_f, g.cam_pitch + g.vp_theta - (np.pi / 2), g.cam_yaw, g.cam_roll,
                       np.atleast_1d(g.cam_barreld), g.cam_lenstype, position=pos, polref=(0, 0, 1), stokes=(1, 0, 0, 0))


def create_arbc_geometry(geospecs=None):
    g = geospecs if geospecs is not None else ArBCSpecs()
    # derived dimensions, CreateArBCGeometry.m:156-204
    t_o = np.array([0, g.ojar_thick]); t_i = np.array([0, g.ijar_thick])
    r1 = np.concatenate([g.ojar_cylrad - t_o, g.ijar_cylrad - t_i])
    r2 = np.concatenate([g.ojar_knucklerad - t_o, g.ijar_knucklerad - t_i])
    r3 = np.concatenate([g.ojar_axrad - t_o, g.ijar_axrad - t_i])
    s = r3 * (r1 - r2) / (r3 - r2)
    z = r2 * np.sqrt(1 - (s / r3) ** 2)
    d = r3 * z * ((1 / r3) - (1 / r2))
    vp_axis = np.array([0, -np.sin(g.vp_theta), np.cos(g.vp_theta)])
    vp_center = np.array([0, -g.vp_s, g.vp_elev], dtype=float)
    ft = np.asarray(g.vp_flange_thick, float)
    head_out_Q = np.diag([g.pv_rad ** -2., g.pv_rad ** -2., g.pv_axrad ** -2.])
    head_in_Q = np.diag([(g.pv_rad - g.pv_thick) ** -2., (g.pv_rad - g.pv_thick) ** -2., (g.pv_axrad - g.pv_thick) ** -2.])
    head_out_P = np.array([0, 0, -2 * g.pv_top * g.pv_axrad ** -2.]); head_in_P = np.array([0, 0, -2 * g.pv_top * (g.pv_axrad - g.pv_thick) ** -2.])
    head_out_R = (g.pv_top / g.pv_axrad) ** 2 - 1; head_in_R = (g.pv_top / (g.pv_axrad - g.pv_thick)) ** 2 - 1

    def cone(b, z0):
        return np.diag([1., 1., -b ** 2]), np.array([0, 0, 2 * b ** 2 * z0]), -(b * z0) ** 2
    rd_cone_b = (g.rdcone_toprad - g.rd_rad) / (g.rdcone_top - g.rd_top)
    rd_cone = cone(rd_cone_b, g.rd_top - (g.rd_rad / rd_cone_b))
    rd_stcone_b = (g.rdcone_toprad - g.rdtopcone_rad) / (g.rdtopcone_bot - g.rdcone_top)
    rd_stcone = cone(rd_stcone_b, g.rdtopcone_bot + (g.rdtopcone_rad / rd_stcone_b))
    rd_topcone = cone(g.rdtopcone_rad / (g.rdtopcone_apex - g.rdtopcone_bot), g.rdtopcone_apex)
    rd_botcone = cone(g.rdbotcone_rad / (g.rdbotcone_apex - g.rdbotcone_bot), g.rdbotcone_apex)

    z0 = np.array([0, 0, 0.]); zax = np.array([0, 0, 1.])
    tgt = dict(ray=g.target_rayleigh, ab=g.target_abslength)

    def bulk(out_is_target, in_is_target):
        return dict(abslength_outside=tgt["ab"] if out_is_target else np.inf, abslength_inside=tgt["ab"] if in_is_target else np.inf,
                    rayleigh_outside=tgt["ray"] if out_is_target else np.inf, rayleigh_inside=tgt["ray"] if in_is_target else np.inf)

    def zband(lo, hi, lo_incl=False, hi_incl=False):
        if lo_incl and not hi_incl:
            return lambda p: np.reshape((p[:, 2, :] < hi) & (p[:, 2, :] >= lo), (p.shape[0], -1))
        if hi_incl and not lo_incl:
            return lambda p: np.reshape((p[:, 2, :] > lo) & (p[:, 2, :] <= hi), (p.shape[0], -1))
        return lambda p: np.reshape((p[:, 2, :] > lo) & (p[:, 2, :] < hi), (p.shape[0], -1))

    def above(zz):
        return lambda p: np.reshape(p[:, 2, :] > zz, (p.shape[0], -1))

    def knuckle(elev, zz, rr):
        return lambda p: np.reshape((p[:, 2, :] > elev) & (p[:, 2, :] <= (zz + elev)) & ((p[:, 0, :] ** 2 + p[:, 1, :] ** 2) > rr ** 2), (p.shape[0], -1))

    c3 = vp_center[None, :, None]; a3 = vp_axis[None, :, None]

    def ax(p):  # axial coordinate along the viewport axis
        return np.sum((p - c3) * a3, axis=1)

    def axband(lo, hi):
        return lambda p: np.reshape((ax(p) > lo) & (ax(p) <= hi), (p.shape[0], -1))

    def disk(center, rlo, rhi):
        cc = np.asarray(center, float)[None, :, None]
        if rlo is None:
            return lambda p: np.reshape(np.sum((p - cc) ** 2, axis=1) <= rhi ** 2, (p.shape[0], -1))
        return lambda p: np.reshape((np.sum((p - cc) ** 2, axis=1) > rlo ** 2) & (np.sum((p - cc) ** 2, axis=1) <= rhi ** 2), (p.shape[0], -1))

    def off_axis2(p):  # squared distance from the viewport axis
        return np.sum((p - c3) ** 2, axis=1) - ax(p) ** 2

    nj, nt, nh, nw, npw, na = g.n_jar, g.n_target, g.n_hydraulic, g.n_pressurewall, g.n_pressurewindow, g.n_air
    ie, oe = g.ijar_elevation, g.ojar_elevation
    sl = []
    # four silica cylinders (CreateArBCGeometry.m:208-243)
    sl.append(_mk('inside surface of inner quartz cylinder', 'cylinder', [z0, zax, r1[3]], zband(ie - g.ijar_cyllength, ie, lo_incl=True), nj, nh))
    sl.append(_mk('outside surface of inner quartz cylinder', 'cylinder', [z0, zax, r1[2]], zband(ie - g.ijar_cyllength, ie, lo_incl=True), nt, nj, **bulk(True, False)))
    sl.append(_mk('inside surface of outer quartz cylinder', 'cylinder', [z0, zax, r1[1]], zband(oe - g.ojar_cyllength, oe, lo_incl=True), nj, nt, **bulk(False, True)))
    sl.append(_mk('outside surface of outer quartz cylinder', 'cylinder', [z0, zax, r1[0]], zband(oe - g.ojar_cyllength, oe, lo_incl=True), nh, nj))
    # four silica domes (:245-280)
    sl.append(_mk('inside surface of inner quartz hemisphere', 'sphere', [np.array([0, 0, ie + d[3]]), r3[3]], above(z[3] + ie), nj, nh))
    sl.append(_mk('outside surface of inner quartz hemisphere', 'sphere', [np.array([0, 0, ie + d[2]]), r3[2]], above(z[2] + ie), nt, nj, **bulk(True, False)))
    sl.append(_mk('inside surface of outer quartz hemisphere', 'sphere', [np.array([0, 0, oe + d[1]]), r3[1]], above(z[1] + oe), nj, nt, **bulk(False, True)))
    sl.append(_mk('outside surface of outer quartz hemisphere', 'sphere', [np.array([0, 0, oe + d[0]]), r3[0]], above(z[0] + oe), nh, nj))
    # four silica knuckles (:282-321)
    sl.append(_mk('inside surface of inner quartz knuckle', 'torus', [np.array([0, 0, ie]), zax, r1[3] - r2[3], r2[3]], knuckle(ie, z[3], r1[3] - r2[3]), nj, nh))
    sl.append(_mk('outside surface of inner quartz knuckle', 'torus', [np.array([0, 0, ie]), zax, r1[2] - r2[2], r2[2]], knuckle(ie, z[2], r1[2] - r2[2]), nt, nj, **bulk(True, False)))
    sl.append(_mk('inside surface of outer quartz knuckle', 'torus', [np.array([0, 0, oe]), zax, r1[1] - r2[1], r2[1]], knuckle(oe, z[1], r1[1] - r2[1]), nj, nt, **bulk(False, True)))
    sl.append(_mk('outside surface of outer quartz knuckle', 'torus', [np.array([0, 0, oe]), zax, r1[0] - r2[0], r2[0]], knuckle(oe, z[0], r1[0] - r2[0]), nh, nj))
    # viewport (:324-509)
    nt_, f = g.vp_nip_top, ft
    sl.append(_mk('sight glass wall', 'cylinder', [vp_center, vp_axis, g.vp_air_rad], axband(0, nt_ + f[1]), nw, na, absorption=1))
    sl.append(_mk('camera can inner wall', 'cylinder', [vp_center, vp_axis, g.vp_can_rad - g.vp_can_wall], axband(nt_ + f[1], g.vp_can_OAL + f[1] + nt_), nw, na, absorption=1))
    sl.append(_mk('camera can outer wall', 'cylinder', [vp_center, vp_axis, g.vp_can_rad], axband(nt_ + f[1] + f[2], g.vp_can_OAL + nt_ + f[1] - f[3]), 1, nw, absorption=1))
    lo1, hi1, lo2, hi2 = -f[0] + nt_, nt_ + f[1] + f[2], nt_ + f[1] + g.vp_can_OAL - f[3], nt_ + f[1] + g.vp_can_OAL + f[4]
    sl.append(_mk('flange outer edge', 'cylinder', [vp_center, vp_axis, g.vp_flange_rad],
                  lambda p: np.reshape(((ax(p) > lo1) & (ax(p) <= hi1)) | ((ax(p) > lo2) & (ax(p) <= hi2)), (p.shape[0], -1)), 1, nw, absorption=1))
    sl.append(_mk('window wall', 'cylinder', [vp_center, vp_axis, g.vp_win_rad], axband(-g.vp_win_thick, 0), nh, npw, absorption=1))
    sl.append(_mk('window retainer outer wall', 'cylinder', [vp_center, vp_axis, g.vp_win_rad], axband(0, nt_), nh, nw, absorption=1))
    sl.append(_mk('pressure vessel nipple wall', 'cylinder', [vp_center, vp_axis, g.vp_nip_rad], axband(-f[0] + nt_, nt_), nw, nh, absorption=1))
    sl.append(_mk('air side of viewport', 'plane', [vp_center, vp_axis], disk(vp_center, None, g.vp_air_rad), na, npw))
    c = vp_center - vp_axis * g.vp_win_thick
    sl.append(_mk('hydraulic side of viewport', 'plane', [c, vp_axis], disk(c, None, g.vp_win_rad), npw, nh))
    sl.append(_mk('viewport retainer', 'plane', [vp_center, vp_axis], disk(vp_center, g.vp_air_rad, g.vp_win_rad), nw, npw, absorption=1))
    c = vp_center - vp_axis * (f[0] - nt_)
    sl.append(_mk('nipple bottom', 'plane', [c, vp_axis], disk(c, g.vp_nip_rad, g.vp_flange_rad), nw, nh, absorption=1))
    c = vp_center + vp_axis * nt_
    sl.append(_mk('nipple top', 'plane', [c, vp_axis], disk(c, g.vp_win_rad, g.vp_flange_rad), nw, nh, absorption=1))
    c = vp_center + vp_axis * (nt_ + f[1])
    sl.append(_mk('can bot', 'plane', [c, vp_axis], disk(c, g.vp_air_rad, g.vp_flange_rad), na, nw, absorption=1))
    c = vp_center + vp_axis * (nt_ + f[1] + f[2])
    sl.append(_mk('can bot_top', 'plane', [c, vp_axis], disk(c, g.vp_can_rad, g.vp_flange_rad), 1, nw, absorption=1))
    c = vp_center + vp_axis * (nt_ + f[1] + g.vp_can_OAL - f[3])
    sl.append(_mk('can top_bot', 'plane', [c, vp_axis], disk(c, g.vp_can_rad, g.vp_flange_rad), nw, 1, absorption=1))
    c = vp_center + vp_axis * (nt_ + f[1] + g.vp_can_OAL)
    sl.append(_mk('can top', 'plane', [c, vp_axis], disk(c, None, g.vp_flange_rad), nw, na, absorption=1))
    c = vp_center + vp_axis * (nt_ + f[1] + g.vp_can_OAL + f[4])
    sl.append(_mk('can very top', 'plane', [c, vp_axis], disk(c, None, g.vp_flange_rad), 1, nw, absorption=1))
    # reflector / diffuser and pressure vessel (:511-629); black in the MATLAB file, optionally reflective here
    rkw = dict(stype=g.reflector_type, absorption=g.reflector_absorption)
    n_refl_in = nh if g.reflector_n_inside is None else g.reflector_n_inside
    if g.reflector_type == "unified":
        rkw["unifiedparams"] = list(g.reflector_unifiedparams)
    sl.append(_mk('reflector/diffuser', 'cylinder', [z0, zax, g.rd_rad], zband(g.rd_bot, g.rd_top, hi_incl=True), n_refl_in, nh, **rkw))
    sl.append(_mk('reflector/diffuser cone', 'quadsurface', list(rd_cone), zband(g.rd_top, g.rdcone_top), n_refl_in, nh, **rkw))
    nr2 = g.vp_nip_rad ** 2
    ckw = rkw if g.reflector_all else dict(absorption=1)   # black in the MATLAB file
    n_cone = n_refl_in if g.reflector_all else nh          # the reflector's far side: away from the jar
    sl.append(_mk('reflector/diffuser strip cone', 'quadsurface', list(rd_stcone),
                  lambda p: np.reshape((p[:, 2, :] > g.rdcone_top) & (p[:, 2, :] < g.rdtopcone_bot) & (off_axis2(p) > nr2), (p.shape[0], -1)), n_cone, nh, **ckw))
    sl.append(_mk('reflector/diffuser topcone', 'quadsurface', list(rd_topcone), zband(g.rdtopcone_bot, g.rdtopcone_apex), n_cone, nh, **ckw))
    sl.append(_mk('reflector/diffuser botcone', 'quadsurface', list(rd_botcone), zband(g.rdbotcone_bot, g.rdbotcone_apex), nh, n_cone, **ckw))
    sl.append(_mk('PV - cylinder outer wall', 'cylinder', [z0, zax, g.pv_rad], zband(g.pv_bot, g.pv_top), 1, nw, absorption=1))
    sl.append(_mk('PV - cylinder inner wall', 'cylinder', [z0, zax, g.pv_rad - g.pv_thick], zband(g.pv_bot, g.pv_top), nw, nh, absorption=1))
    fr2 = g.vp_flange_rad ** 2
    head_inb = lambda p: np.reshape((p[:, 2, :] > g.pv_top) & (off_axis2(p) > fr2), (p.shape[0], -1))  # noqa: E731
    sl.append(_mk('PV - outer top', 'quadsurface', [head_out_Q, head_out_P, head_out_R], head_inb, 1, nw, absorption=1))
    sl.append(_mk('PV - inner top', 'quadsurface', [head_in_Q, head_in_P, head_in_R], head_inb, nw, nh, absorption=1))
    sl.append(_mk('PV - bot', 'plane', [np.array([0, 0, g.pv_bot], float), np.array([0, 0, -1.])],
                  lambda p: np.reshape((p[:, 0, :] ** 2 + p[:, 1, :] ** 2) <= g.pv_rad ** 2, (p.shape[0], -1)), nw, nh, absorption=1))
    return sl


class SBCSpecs(object):
    """defaults of CreateSBCGeometry.m:14-44"""

    def __init__(self, **kw):
        self.itube_ID = 2; self.itube_OD = 2.3; self.otube_ID = 2.4; self.otube_OD = 3
        self.icap_thick = .5; self.ocap_thick = .5; self.PMTgap = 0.01; self.PMTwinthick = 0.1
        self.piezocover_reflectance = 0.85; self.target_height = 1.25 * 2.54 - .5; self.can_obot = -1.125 * 2.54 * 4
        self.n_vacuum = 1; self.n_xenon = 1.69; self.n_quartz = 1.59
        self.rayleigh_xenon = 29; self.abslength_xenon = 300; self.abslength_quartz = .083; self.abslength_silica = 7; self.side_absorb = 0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("unknown geospec %r" % k)
            setattr(self, k, v)


def create_sbc_geometry(geospecs=None):
    g = geospecs if geospecs is not None else SBCSpecs()
    n_jar, n_target, nv = g.n_quartz, g.n_xenon, g.n_vacuum
    inf = np.inf
    z0 = np.array([0, 0, 0.]); zax = np.array([0, 0, 1.])

    def zb(lo, hi):
        return lambda p: np.reshape((p[:, 2, :] < hi) & (p[:, 2, :] > lo), (p.shape[0], -1))

    def disk(r):
        return lambda p: np.reshape((p[:, 0, :] ** 2 + p[:, 1, :] ** 2) <= r ** 2, (p.shape[0], -1))

    def b(ao, ai, ro, ri):
        return dict(abslength_outside=ao, abslength_inside=ai, rayleigh_outside=ro, rayleigh_inside=ri)
    th, ic, oc = g.target_height, g.icap_thick, g.ocap_thick
    sl = []
    sl.append(_mk('ID of inner tube', 'cylinder', [z0, zax, .5 * g.itube_ID], zb(g.can_obot - 2, -ic), n_jar, nv, **b(g.abslength_quartz, inf, inf, inf)))
    sl.append(_mk('bottom-side of inner tube cap', 'plane', [np.array([0, 0, -ic]), zax], disk(.5 * g.itube_ID), n_jar, nv, **b(g.abslength_silica, inf, inf, inf)))
    if g.piezocover_reflectance == 0:
        sl.append(_mk('reflector under of inner tube cap', 'plane', [np.array([0, 0, -ic - g.PMTgap]), zax], disk(.5 * g.otube_OD), nv, inf, absorption=1, **b(inf, inf, inf, inf)))
    else:
        sl.append(_mk('reflector under of inner tube cap', 'plane', [np.array([0, 0, -ic - g.PMTgap]), zax], disk(.5 * g.itube_ID), nv, inf,
                      absorption=1 - g.piezocover_reflectance, **b(inf, inf, inf, inf)))
    sl.append(_mk('OD of inner tube', 'cylinder', [z0, zax, .5 * g.itube_OD], zb(g.can_obot - 2, 0), n_target, n_jar, absorption=g.side_absorb,
                  **b(g.abslength_xenon, g.abslength_quartz, g.rayleigh_xenon, inf)))
    sl.append(_mk('top-side of inner tube cap', 'plane', [z0, zax], disk(.5 * g.itube_OD), n_target, n_jar, **b(g.abslength_xenon, g.abslength_silica, g.rayleigh_xenon, inf)))
    sl.append(_mk('ID of outer tube', 'cylinder', [z0, zax, .5 * g.otube_ID], zb(g.can_obot - 2, th), n_jar, n_target, absorption=g.side_absorb,
                  **b(g.abslength_quartz, g.abslength_xenon, inf, g.rayleigh_xenon)))
    sl.append(_mk('bottom-side of outer tube cap', 'plane', [np.array([0, 0, th]), zax], disk(.5 * g.otube_ID), n_jar, n_target, **b(g.abslength_silica, g.abslength_xenon, inf, g.rayleigh_xenon)))
    sl.append(_mk('OD of outer tube', 'cylinder', [z0, zax, .5 * g.otube_OD], zb(g.can_obot - 2, th + oc), nv, n_jar, **b(inf, g.abslength_quartz, inf, inf)))
    sl.append(_mk('top-side of outer tube cap', 'plane', [np.array([0, 0, th + oc]), zax], disk(.5 * g.otube_OD), nv, n_jar, **b(inf, g.abslength_silica, inf, inf)))
    sl.append(_mk('bottom-side of PMT window', 'plane', [np.array([0, 0, th + oc + g.PMTgap]), zax], disk(.5 * g.otube_OD), n_jar, nv, **b(g.abslength_silica, inf, inf, inf)))
    sl.append(_mk('photocathode', 'plane', [np.array([0, 0, th + oc + g.PMTgap + g.PMTwinthick]), zax], disk(.5 * g.otube_OD), nv, n_jar, absorption=1, **b(inf, g.abslength_silica, inf, inf)))
    return sl


def sbc_full_mc_specs(black_cones=False, reflector="unified"):
    """jar geometry with Rayleigh/absorption lengths of CreateSBCGeometry.m:40-41 in the target
    fluid and reflective (UNIFIED, rough conductor, 10 % absorption) reflector/diffuser surfaces instead of the black
    ones: the cylinder, its cone and the cones around the jar.  In single-child mode a reflector hit scales the ray's
    intensity by 0.9 instead of ending it, so light keeps bouncing between jar and reflector until the intensity
    threshold or max_scatters ends it (about 87 interactions per ray at K = 100, threshold 1e-4).
    black_cones=True leaves the cones black as in the MATLAB file: nearly every ray ends on the top cone after 3-4
    interactions (the workload of the builds r1a-r1f in DESIGN.md).
    reflector="diffuse": config 5 as SURVEY 8d words it - the reflector surfaces are of surface_type 'diffuse'
    (MATLAB/RayTracer2.m:394-415: Lambertian re-draw of the normal, then a mirror step off a conductor), same 10 %
    absorption."""
    if reflector == "diffuse":
        return ArBCSpecs(target_rayleigh=29.0, target_abslength=300.0, reflector_type="diffuse", reflector_absorption=0.1,
                         reflector_n_inside=np.inf, reflector_all=not black_cones)
    return ArBCSpecs(target_rayleigh=29.0, target_abslength=300.0, reflector_type="unified", reflector_absorption=0.1,
                     reflector_unifiedparams=[0.1, 0.0, 0.3, 0.2, 0.05], reflector_n_inside=np.inf, reflector_all=not black_cones)
