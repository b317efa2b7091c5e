"""Seeded test geometries / ray sets shared by the golden generator and the tests.

Geometries are the ones of the reference's own demo scripts (python/TestGeometry.py:41-106,
python/SphereTest.py:20-31, python/UnifiedTest.py:12-37) plus SURVEY.md section 8d's config 1.
`cls` is the surface class to instantiate: the reference's `surface.surface` when generating goldens,
anything attribute-bag-like otherwise.
"""
import types

import numpy as np


class Surf(object):
    """attribute bag with the fields of python/surface.py:51-62"""

    def __init__(self):
        self.description = None
        self.n_outside = None
        self.n_inside = None
        self.surface_type = None
        self.absorption = None
        self.shape = None
        self.param_list = None
        self.inbounds_function = None


def isotropic_rays(n, seed, origin=(0.0, 0.0, 0.0), polarized=False):
    rng = np.random.default_rng(seed)
    mu = rng.uniform(-1, 1, n)
    ph = rng.uniform(0, 2 * np.pi, n)
    st = np.sqrt(1 - mu * mu)
    rays = np.zeros((n, 10))
    rays[:, 0] = st * np.cos(ph); rays[:, 1] = st * np.sin(ph); rays[:, 2] = mu
    rays[:, 3] = 1.0
    # avoid a polref parallel to the direction
    bad = np.abs(rays[:, 0]) > 0.999
    rays[bad, 3] = 0; rays[bad, 4] = 1
    rays[:, 6] = 1.0
    if polarized:
        pol = rng.uniform(-1, 1, (n, 3)) * 0.55
        rays[:, 7:10] = pol
    sp = np.tile(np.asarray(origin, float), (n, 1))
    return sp, rays


def sphere_plane(cls=Surf):
    """dielectric sphere (n 1.5 inside) + absorbing plane below it."""
    s = cls()
    s.description = 'sphere R=10 n_in 1.5'
    s.shape = 'sphere'; s.param_list = [np.array([0, 0, 0]), 10]
    s.inbounds_function = lambda p: np.reshape((p[:, 2, :] >= -10) * (p[:, 2, :] <= 10), (p.shape[0], -1))
    s.n_outside = 1.0; s.n_inside = 1.5; s.surface_type = 'normal'; s.absorption = 0
    pl = cls()
    pl.description = 'absorbing plane z=-20'
    pl.shape = 'plane'; pl.param_list = [np.array([0, 0, -20]), np.array([0, 0, 1])]
    pl.inbounds_function = lambda p: np.reshape((p[:, 0] ** 2 + p[:, 1] ** 2) < 10000, (p.shape[0], -1))
    pl.n_outside = 1.0; pl.n_inside = np.inf; pl.surface_type = 'normal'; pl.absorption = 1
    return [s, pl]


def testgeometry(cls=Surf, cyl_type='unified', cyl_params=(0, 0, 0, 0, 0)):
    """python/TestGeometry.py:41-106 (two-cell cylinder).  cyl_type='normal' gives a deterministic variant."""
    sl = []
    a = cls(); a.description = 'bottom cylinder'; a.shape = 'cylinder'
    a.param_list = [np.array([0, 0, 0]), np.array([0, 0, 1]), 10]
    a.inbounds_function = lambda p: np.reshape((p[:, 2, :] > 0) * (p[:, 2, :] < 5), (p.shape[0], -1))
    a.n_outside = 1.5; a.n_inside = 1.5; a.surface_type = cyl_type; a.absorption = 0
    b = cls(); b.description = 'top cylinder'; b.shape = 'cylinder'
    b.param_list = [np.array([0, 0, 0]), np.array([0, 0, 1]), 10]
    b.inbounds_function = lambda p: np.reshape((p[:, 2, :] >= 5) * (p[:, 2, :] < 10), (p.shape[0], -1))
    b.n_outside = 1.5; b.n_inside = 2; b.surface_type = cyl_type; b.absorption = 0
    if cyl_type == 'unified':
        a.unifiedparams = list(cyl_params); b.unifiedparams = list(cyl_params)
    if cyl_type == 'normal':
        a.n_outside = 1.0; b.n_outside = 1.0
    c = cls(); c.description = 'top cap'; c.shape = 'plane'
    c.param_list = [np.array([0, 0, 10]), np.array([0, 0, 1])]
    c.inbounds_function = lambda p: np.reshape((p[:, 0] ** 2 + p[:, 1] ** 2) < 100, (p.shape[0], -1))
    c.n_outside = 1.5; c.n_inside = 2; c.surface_type = 'normal'; c.absorption = 1
    d = cls(); d.description = 'mid disk'; d.shape = 'plane'
    d.param_list = [np.array([0, 0, 5]), np.array([0, 0, 1])]
    d.inbounds_function = lambda p: np.reshape((p[:, 0] ** 2 + p[:, 1] ** 2 < 100), (p.shape[0], -1))
    d.n_outside = 2; d.n_inside = 1.5; d.surface_type = 'normal'; d.absorption = 0
    e = cls(); e.description = 'bottom cap'; e.shape = 'plane'
    e.param_list = [np.array([0, 0, 0]), np.array([0, 0, 1])]
    e.inbounds_function = lambda p: np.reshape((p[:, 0] ** 2 + p[:, 1] ** 2 < 100), (p.shape[0], -1))
    e.n_outside = 1.5; e.n_inside = 1.5; e.surface_type = 'normal'; e.absorption = 1
    return [a, b, c, d, e]


def unifiedtest(cls=Surf, params=(0, 0, .25, .5, .1), n_panel_in=np.inf, n_med=1.5):
    """python/UnifiedTest.py:12-37: absorbing hemisphere over a UNIFIED panel."""
    h = cls(); h.description = 'hemisphere'; h.shape = 'sphere'
    h.param_list = [np.array([0, 0, 0]), 10]
    h.inbounds_function = lambda p: np.reshape((p[:, 2, :] >= 0) * (p[:, 2, :] <= 10), (p.shape[0], -1))
    h.n_outside = np.inf; h.n_inside = n_med; h.surface_type = 'normal'; h.absorption = 1
    q = cls(); q.description = 'panel'; q.shape = 'plane'
    q.param_list = [np.array([0, 0, 0]), np.array([0, 0, 1])]
    q.inbounds_function = lambda p: np.reshape((p[:, 0] ** 2 + p[:, 1] ** 2) < 100, (p.shape[0], -1))
    q.n_outside = n_med; q.n_inside = n_panel_in; q.surface_type = 'unified'
    q.unifiedparams = list(params); q.absorption = 0
    return [h, q]


def unifiedtest_rays(n):
    """python/UnifiedTest.py:43-62: n identical rays from (-5,0,5) towards the origin."""
    sp = np.tile(np.array([-5.0, 0.0, 5.0]), (n, 1))
    rays = np.zeros((n, 10))
    rays[:, 3] = 1; rays[:, 6] = 1
    rays[:, 0] = 1; rays[:, 1] = 0; rays[:, 2] = -1
    return sp, rays


def spheretest(cls=Surf):
    """python/SphereTest.py:20-31: one absorbing sphere."""
    s = cls(); s.description = 'sphere'; s.shape = 'sphere'
    s.param_list = [np.array([0, 0, 0]), 10]
    s.inbounds_function = lambda p: np.reshape((p[:, 2, :] >= -10) * (p[:, 2, :] <= 10), (p.shape[0], -1))
    s.n_outside = np.inf; s.n_inside = 2; s.surface_type = 'normal'; s.absorption = 1
    return [s]


RI_FIELDS = ("incoming_ray", "reflected_ray", "refracted_ray", "intersection_point", "surface_normal", "ray_index",
             "surface_index", "distance_traveled", "n_incident", "n_transmitted", "bulkabs_incident",
             "bulkabs_transmitted", "rayleigh_incident", "rayleigh_transmitted")


def pack_interfaces(ris, prefix="s"):
    """list of rayInterfaces -> dict of arrays for np.savez"""
    d = {"num_interfaces": np.array(len(ris))}
    for k, ri in enumerate(ris):
        for f in RI_FIELDS:
            d["%s%d_%s" % (prefix, k, f)] = np.asarray(getattr(ri, f), dtype=np.float64)
    return d


def unpack_interfaces(z, prefix="s"):
    out = []
    for k in range(int(z["num_interfaces"])):
        ri = types.SimpleNamespace()
        for f in RI_FIELDS:
            setattr(ri, f, z["%s%d_%s" % (prefix, k, f)])
        out.append(ri)
    return out


def mixed_geometry(cls=Surf):
    """Small closed scene exercising every shape and surface type: a torus and a quadric cone inside an absorbing sphere,
    a diffuse disk, a retro disk, a UNIFIED rough cylinder, with bulk absorption and Rayleigh scattering in the fill."""
    def mk(desc, shape, pl, inb, no, ni, st, ab, **kw):
        s = cls(); s.description = desc; s.shape = shape; s.param_list = pl; s.inbounds_function = inb
        s.n_outside = no; s.n_inside = ni; s.surface_type = st; s.absorption = ab
        for k, v in kw.items():
            setattr(s, k, v)
        return s
    fill = dict(abslength_inside=80.0, rayleigh_inside=25.0, abslength_outside=np.inf, rayleigh_outside=np.inf)
    sl = []
    sl.append(mk('outer sphere', 'sphere', [np.array([0, 0, 0.]), 30.], lambda p: np.reshape(p[:, 2, :] > -1e9, (p.shape[0], -1)), np.inf, 1.33, 'normal', 1.0, **fill))
    sl.append(mk('glass torus', 'torus', [np.array([0, 0, 4.]), np.array([0, 0, 1.]), 8., 2.], lambda p: np.reshape(p[:, 2, :] > -1e9, (p.shape[0], -1)), 1.33, 1.5, 'normal', 0.0,
                 abslength_outside=80.0, rayleigh_outside=25.0, abslength_inside=200.0, rayleigh_inside=np.inf))
    Q = np.diag([1., 1., -0.25]); Pv = np.array([0, 0, -5.]); R = -25.0  # cone x^2+y^2 = (z+10)^2/4
    sl.append(mk('mirror cone', 'quadsurface', [Q, Pv, R], lambda p: np.reshape((p[:, 2, :] > -9.5) & (p[:, 2, :] < -4), (p.shape[0], -1)), 1.6, 1.33, 'normal', 0.2, **fill))
    sl.append(mk('diffuse disk', 'plane', [np.array([0, 0, 15.]), np.array([0, 0, 1.])], lambda p: np.reshape((p[:, 0, :] ** 2 + p[:, 1, :] ** 2) < 10000, (p.shape[0], -1)), np.inf, 1.33, 'diffuse', 0.3, **fill))
    sl.append(mk('retro disk', 'plane', [np.array([0, 0, -15.]), np.array([0, 0, 1.])], lambda p: np.reshape((p[:, 0, :] ** 2 + p[:, 1, :] ** 2) < 10000, (p.shape[0], -1)), 1.33, np.inf, 'retro', 0.1,
                 abslength_outside=80.0, rayleigh_outside=25.0, abslength_inside=np.inf, rayleigh_inside=np.inf))
    sl.append(mk('rough cylinder', 'cylinder', [np.array([0, 0, 0.]), np.array([.1, 0, 1.]), 20.], lambda p: np.reshape((p[:, 2, :] > -12) & (p[:, 2, :] <= 12), (p.shape[0], -1)), 1.5, 1.33, 'unified', 0.05,
                 unifiedparams=[.15, .7, .3, .2, .1], **fill))
    return sl


def coaxial_scene(seed=0, cls=Surf):
    """Random scene around a tilted axis (unit vector a, point c0): cylinders, planes normal to a, spheres centred on the
    axis, a cone and an ellipsoid of revolution, a torus - all bounded by ranges of the axial coordinate, of the radius
    about the axis and of the distance from a point of the axis, i.e. everything the frame program evaluates without the
    hit point - plus one off-axis sphere and one bound the frame program must leave to the generic evaluator."""
    rng = np.random.default_rng(seed)
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    c0 = rng.normal(size=3) * 3
    e1 = np.cross(a, [1.0, 0, 0]); e1 /= np.linalg.norm(e1); e2 = np.cross(a, e1)
    Pp = np.eye(3) - np.outer(a, a)  # projector on the plane normal to the axis

    def ax(p):  # axial coordinate relative to c0
        return (p[:, 0, :] - c0[0]) * a[0] + (p[:, 1, :] - c0[1]) * a[1] + (p[:, 2, :] - c0[2]) * a[2]

    def rad2(p):
        dx = p[:, 0, :] - c0[0]; dy = p[:, 1, :] - c0[1]; dz = p[:, 2, :] - c0[2]
        t = dx * a[0] + dy * a[1] + dz * a[2]
        return (dx - t * a[0]) ** 2 + (dy - t * a[1]) ** 2 + (dz - t * a[2]) ** 2

    def dist2(p, t):
        q = c0 + t * a
        return (p[:, 0, :] - q[0]) ** 2 + (p[:, 1, :] - q[1]) ** 2 + (p[:, 2, :] - q[2]) ** 2

    def mk(desc, shape, pl, inb, no, ni, st='normal', ab=0.0, **kw):
        s = cls(); s.description = desc; s.shape = shape; s.param_list = pl; s.inbounds_function = inb
        s.n_outside = no; s.n_inside = ni; s.surface_type = st; s.absorption = ab
        for k, v in kw.items():
            setattr(s, k, v)
        return s
    sl = []
    sl.append(mk('outer wall', 'cylinder', [c0 + 3.0 * a, a, 20.0], lambda p: np.reshape((ax(p) > -15) & (ax(p) <= 15), (p.shape[0], -1)), np.inf, 1.3, ab=1.0))
    sl.append(mk('glass cylinder in', 'cylinder', [c0, -a, 6.0], lambda p: np.reshape((ax(p) >= -8) & (ax(p) < 8), (p.shape[0], -1)), 1.3, 1.5))
    sl.append(mk('glass cylinder out', 'cylinder', [c0 - 2.0 * a, a, 6.5], lambda p: np.reshape((ax(p) >= -8) & (ax(p) < 8), (p.shape[0], -1)), 1.5, 1.3))
    sl.append(mk('top cap', 'plane', [c0 + 15.0 * a, a], lambda p: np.reshape(rad2(p) < 400.0, (p.shape[0], -1)), np.inf, 1.3, ab=1.0))
    sl.append(mk('bottom cap', 'plane', [c0 - 15.0 * a + 0.7 * e1, -a], lambda p: np.reshape(dist2(p, -15.0) <= 400.0, (p.shape[0], -1)), np.inf, 1.3, ab=0.6))
    sl.append(mk('ring plane', 'plane', [c0 + 8.0 * a, a], lambda p: np.reshape((rad2(p) > 36.0) & (rad2(p) < 42.25), (p.shape[0], -1)), 1.5, 1.3))
    sl.append(mk('lens sphere', 'sphere', [c0 + 11.0 * a, 4.0], lambda p: np.reshape(ax(p) > -1e9, (p.shape[0], -1)), 1.3, 1.6))
    sl.append(mk('dome', 'sphere', [c0 - 8.0 * a, 6.0], lambda p: np.reshape(ax(p) < -8, (p.shape[0], -1)), 1.3, 1.5))
    # cone about the axis, apex at c0 - 20 a: |x_perp|^2 = 0.09 (xa + 20)^2
    Q = Pp - 0.09 * np.outer(a, a); q0 = c0 - 20.0 * a
    sl.append(mk('cone', 'quadsurface', [Q, -2 * Q @ q0, float(q0 @ Q @ q0)], lambda p: np.reshape((ax(p) > 9) & (ax(p) < 14), (p.shape[0], -1)), 1.3, 1.7, ab=0.3))
    # ellipsoid of revolution centred at c0 + 2 a
    Q2 = Pp / 9.0 + np.outer(a, a) / 16.0; q2 = c0 + 2.0 * a
    sl.append(mk('ellipsoid', 'quadsurface', [Q2, -2 * Q2 @ q2, float(q2 @ Q2 @ q2) - 1.0], lambda p: np.reshape(dist2(p, 2.0) > 9.5, (p.shape[0], -1)), 1.3, 1.45))
    sl.append(mk('torus', 'torus', [c0 - 11.0 * a, a, 9.0, 1.5], lambda p: np.reshape((ax(p) > -12.2) & (rad2(p) > 64.0), (p.shape[0], -1)), 1.3, 1.5))
    sl.append(mk('off-axis ball', 'sphere', [c0 + 9.0 * e1 + 4.0 * a, 2.0], lambda p: np.reshape(ax(p) > -1e9, (p.shape[0], -1)), 1.3, 2.0, ab=0.5))
    sl.append(mk('half cylinder', 'cylinder', [c0, a, 12.0], lambda p: np.reshape((ax(p) > -3) & (ax(p) < 3) & ((p[:, 0, :] - c0[0]) * e1[0] + (p[:, 1, :] - c0[1]) * e1[1] + (p[:, 2, :] - c0[2]) * e1[2] > 0), (p.shape[0], -1)), 1.3, 1.5))
    origin = c0 + 1.0 * a + 0.5 * e2
    return sl, origin
