"""Analytic known answers for the sections of the path that no executable reference pins (torus,
quadsurface, Rayleigh, diffuse, retro, bulk absorption, UNIFIED sigma_alpha > 0, tir_handling >= 0).

Every check takes `trace(sp, rays, surfacelist, *args, **kw)` -> object with .ray_interfaces / .absorption_table, so
the same known answers are put to the CPU oracle (tests/test_oracle_invariants.py) and to the HIP path through the
C ABI (tests/test_gpu_invariants.py).  The expected values come from closed forms, numpy.roots or numerical
integration of the densities the cited MATLAB lines sample from - never from the code under test.

Statistical checks are chi-square tests at 3 sigma: chi2 < dof + 3 sqrt(2 dof), seeds fixed.
"""
import numpy as np

import cases

MATLAB = dict(compat="matlab")


def chi2_ok(obs, exp, label=""):
    """Pearson chi-square of observed counts against expected counts (same total), bins with exp < 20 merged into one"""
    obs = np.asarray(obs, float).ravel(); exp = np.asarray(exp, float).ravel()
    small = exp < 20
    if small.any():
        obs = np.append(obs[~small], obs[small].sum()); exp = np.append(exp[~small], exp[small].sum())
        if exp[-1] < 5:
            obs = obs[:-1]; exp = exp[:-1]
    chi2 = float(np.sum((obs - exp) ** 2 / exp)); dof = obs.size - 1
    lim = dof + 3 * np.sqrt(2 * dof)
    assert chi2 < lim, "%s: chi2 %.1f over %d dof exceeds the 3 sigma limit %.1f" % (label, chi2, dof, lim)
    return chi2, dof


def mk(desc, shape, pl, inb, no, ni, st='normal', ab=0.0, **kw):
    s = cases.Surf(); s.description = desc; s.shape = shape; s.param_list = pl; s.inbounds_function = inb
    s.n_outside = no; s.n_inside = ni; s.surface_type = st; s.absorption = ab
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def everywhere(p):
    return np.reshape(p[:, 2, :] > -1e300, (p.shape[0], -1))


def everywhere_plane(p):
    return np.reshape(p[:, 2] > -1e300, (p.shape[0], -1))


def absorbing_sphere(R, n_in=1.0, **kw):
    return mk('absorbing sphere', 'sphere', [np.array([0, 0, 0.]), float(R)], everywhere, np.inf, n_in, 'normal', 1.0, **kw)


def rays_from(dirs, polref=None, stokes=(1, 0, 0, 0)):
    d = np.atleast_2d(np.asarray(dirs, float)); n = d.shape[0]
    rays = np.zeros((n, 10)); rays[:, 0:3] = d
    if polref is None:
        pol = np.cross(d, [0, 0, 1.0]); bad = np.linalg.norm(pol, axis=1) < 1e-6
        pol[bad] = np.cross(d[bad], [1.0, 0, 0])
    else:
        pol = np.broadcast_to(np.asarray(polref, float), (n, 3)).copy()
    rays[:, 3:6] = pol; rays[:, 6:10] = np.asarray(stokes, float)
    return rays


# ------------------------------------------------------------------------------------------------ torus (a7)
def torus_quartic(center, axis, r1, r2, sp, d):
    """a4..a0 exactly as MATLAB/RayToX/RayToTorus.m:88-103 builds them (numpy, row by row)"""
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    x = sp - center
    y = (x @ axis)[:, None] * axis
    u = y - x
    w = (d @ axis)[:, None] * axis
    v = w - d
    s = lambda a, b: np.sum(a * b, axis=1)  # noqa: E731
    a4 = (s(v, v) + s(w, w)) ** 2
    a3 = 4 * (s(v, v) + s(w, w)) * (s(u, v) + s(y, w))
    a2 = 4 * (s(u, v) + s(y, w)) ** 2 - 4 * r1 ** 2 * s(v, v) + 2 * (s(v, v) + s(w, w)) * (s(u, u) + r1 ** 2 + s(y, y) - r2 ** 2)
    a1 = 4 * (s(u, v) + s(y, w)) * (s(u, u) + r1 ** 2 + s(y, y) - r2 ** 2) - 8 * r1 ** 2 * s(u, v)
    a0 = (s(u, u) + r1 ** 2 + s(y, y) - r2 ** 2) ** 2 - 4 * r1 ** 2 * s(u, u)
    return np.column_stack([a4, a3, a2, a1, a0])


def roots_reference(coef, tangency=1e-6):
    """numpy.roots (companion-matrix eigenvalues, the algorithm of MATLAB roots(), RayToTorus.m:137-141) per ray.
    Returns sorted real roots (nan-padded to 4), and `tangent` = the classification real/complex is not decidable at
    double precision: two roots closer than `tangency` (relative to the root scale)."""
    n = coef.shape[0]
    real = np.full((n, 4), np.nan); tangent = np.zeros(n, bool)
    for i in range(n):
        r = np.roots(coef[i])
        scale = max(1.0, np.max(np.abs(r)))
        dist = np.abs(r[:, None] - r[None, :]) + np.eye(r.size) * 1e300
        tangent[i] = dist.min() < tangency * scale
        rr = np.sort(r.real[r.imag == 0])          # MATLAB/RayTracer2.m:341 keeps roots with imag == 0
        real[i, :rr.size] = rr
    return real, tangent


TORUS = dict(center=np.array([0.3, -0.2, 0.5]), axis=np.array([0.2, -0.1, 1.0]), r1=8.0, r2=2.0)


def torus_rays(n, seed, grazing=200):
    """random rays through the neighbourhood of the torus + `grazing` rays built tangent to it, offset along the
    normal by +-1e-3 .. 1e-9"""
    rng = np.random.default_rng(seed)
    sp = rng.uniform(-14, 14, (n, 3)); tgt = rng.uniform(-9, 9, (n, 3))
    d = tgt - sp
    ax = TORUS["axis"] / np.linalg.norm(TORUS["axis"])
    e1 = np.cross(ax, [1.0, 0, 0]); e1 /= np.linalg.norm(e1); e2 = np.cross(ax, e1)
    th = rng.uniform(0, 2 * np.pi, grazing); ps = rng.uniform(0, 2 * np.pi, grazing)
    ring = np.cos(th)[:, None] * e1 + np.sin(th)[:, None] * e2
    nrm = np.cos(ps)[:, None] * ring + np.sin(ps)[:, None] * ax
    pt = TORUS["center"] + TORUS["r1"] * ring + TORUS["r2"] * nrm
    t1 = np.cross(nrm, ax); t1 /= np.linalg.norm(t1, axis=1, keepdims=True); t2 = np.cross(nrm, t1)
    al = rng.uniform(0, 2 * np.pi, grazing)
    tang = np.cos(al)[:, None] * t1 + np.sin(al)[:, None] * t2
    eps = 10.0 ** rng.integers(-9, -2, grazing) * rng.choice([-1.0, 1.0], grazing)
    gsp = pt + eps[:, None] * nrm - 6.0 * tang
    return np.vstack([sp, gsp]), np.vstack([d, tang])


def check_torus_first_hit(trace, n=10000, seed=7, coaxial=False):
    """First hit of every ray on a bare torus inside an absorbing sphere against numpy.roots of RayToTorus.m:97-103:
    same hit / miss, distance to 1e-9 relative.  Rays within `tangency` of a double root are excluded and counted.
    coaxial: a cylinder on the torus' axis, out of reach, joins the scene - the HIP path then treats the torus as a
    member of a frame (rt_device.hpp PK_TORUS) instead of by the generic formulas; the answer is the same."""
    sp, d = torus_rays(n, seed)
    tor = mk('torus', 'torus', [TORUS["center"], TORUS["axis"], TORUS["r1"], TORUS["r2"]], everywhere, 1.0, 1.0)
    sl = [absorbing_sphere(60.0), tor]
    if coaxial:
        sl.append(mk('far cylinder', 'cylinder', [TORUS["center"], TORUS["axis"], 100.0], everywhere, 1.0, 1.0))
    rays = rays_from(d)
    res = trace(sp, rays, sl, 1, 1e-9, 0.0, -1.0, True, False, False, **MATLAB)
    ri = res.ray_interfaces[0]
    assert ri.ray_index.size == sp.shape[0]                       # every ray ends somewhere (closed sphere)
    order = np.argsort(np.abs(ri.ray_index), kind="stable")
    six = np.abs(ri.surface_index[order]).astype(int); l = ri.distance_traveled[order]
    real, tangent = roots_reference(torus_quartic(TORUS["center"], TORUS["axis"], TORUS["r1"], TORUS["r2"], sp, d))
    pos = np.where(real > 1e-9, real, np.inf)
    first = np.min(np.where(np.isnan(pos), np.inf, pos), axis=1)
    expect_hit = np.isfinite(first)
    ok = ~tangent
    assert ok.sum() > 0.97 * sp.shape[0]
    assert np.array_equal((six == 2)[ok], expect_hit[ok]), "hit/miss differs from numpy.roots on %d rays" % int(((six == 2) != expect_hit)[ok].sum())
    h = ok & expect_hit
    err = np.abs(l[h] - first[h]) / np.maximum(1.0, np.abs(first[h]))
    assert err.max() < 1e-9, err.max()
    return dict(rays=int(sp.shape[0]), excluded_near_tangency=int(tangent.sum()), hits=int(h.sum()), worst_rel=float(err.max()))


# ------------------------------------------------------------------------------------------------ quadric (a6)
def check_quadric_closed_forms(trace, n=4000, seed=3):
    """ellipsoid: hits from the unit-sphere problem in scaled coordinates; cone: sign change of x^2+y^2-k^2 z^2 along the
    ray bracketed and bisected (scipy.brentq); paraboloid along its axis: the linear_cut branch of
    RayToQuadsurface.m:92-101 (a == 0: one root -c/b)."""
    from scipy.optimize import brentq
    rng = np.random.default_rng(seed)
    out = {}
    # --- ellipsoid x^2/9 + y^2/16 + z^2/4 = 1 (glass), rays from inside
    abc = np.array([3.0, 4.0, 2.0])
    ell = mk('ellipsoid', 'quadsurface', [np.diag(1 / abc ** 2), np.zeros(3), -1.0], everywhere, 1.0, 1.0)
    sp = rng.uniform(-1, 1, (n, 3)) * abc * 0.5
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    res = trace(sp, rays_from(d), [absorbing_sphere(50.0), ell], 1, 1e-9, 0.0, -1.0, True, False, False, **MATLAB)
    ri = res.ray_interfaces[0]; o = np.argsort(np.abs(ri.ray_index), kind="stable")
    s2, d2 = sp / abc, d / abc                                    # unit sphere in scaled coordinates
    A = np.sum(d2 * d2, 1); B = np.sum(s2 * d2, 1); Cc = np.sum(s2 * s2, 1) - 1
    lexp = (-B + np.sqrt(B * B - A * Cc)) / A
    assert np.all(np.abs(ri.surface_index[o]) == 2)
    err = np.abs(ri.distance_traveled[o] - lexp) / lexp
    assert err.max() < 1e-12, err.max()
    nexp = (sp + lexp[:, None] * d) / abc ** 2; nexp /= np.linalg.norm(nexp, axis=1, keepdims=True)
    assert np.abs(ri.surface_normal[o] + nexp).max() < 1e-12      # hit from inside: normal turned against the ray
    out["ellipsoid_worst_rel"] = float(err.max())
    # --- cone x^2 + y^2 = (z/2)^2, z in (1, 9), rays from outside towards the axis
    k2 = 0.25
    cone = mk('cone', 'quadsurface', [np.diag([1.0, 1.0, -k2]), np.zeros(3), 0.0],
              lambda p: np.reshape((p[:, 2, :] > 1) & (p[:, 2, :] < 9), (p.shape[0], -1)), 1.0, 1.0)
    m = 600
    ph = rng.uniform(0, 2 * np.pi, m); z0 = rng.uniform(2, 8, m)
    sp = np.column_stack([12 * np.cos(ph), 12 * np.sin(ph), z0])
    tgt = np.column_stack([rng.uniform(-1, 1, m), rng.uniform(-1, 1, m), rng.uniform(2.5, 7.5, m)])
    d = tgt - sp; d /= np.linalg.norm(d, axis=1, keepdims=True)
    res = trace(sp, rays_from(d), [absorbing_sphere(50.0), cone], 1, 1e-9, 0.0, -1.0, True, False, False, **MATLAB)
    ri = res.ray_interfaces[0]; o = np.argsort(np.abs(ri.ray_index), kind="stable")
    f = lambda t, i: (sp[i, 0] + t * d[i, 0]) ** 2 + (sp[i, 1] + t * d[i, 1]) ** 2 - k2 * (sp[i, 2] + t * d[i, 2]) ** 2  # noqa: E731
    lexp = np.zeros(m)
    for i in range(m):
        ts = np.linspace(0, 12.5, 2001); fs = f(ts, i)
        j = np.nonzero(fs[:-1] * fs[1:] < 0)[0][0]                # outside (f > 0) -> inside the cone
        lexp[i] = brentq(f, ts[j], ts[j + 1], args=(i,), xtol=1e-14, rtol=1e-15)
    zh = sp[:, 2] + lexp * d[:, 2]
    inb = (zh > 1) & (zh < 9)
    assert inb.all()
    assert np.all(np.abs(ri.surface_index[o]) == 2)
    err = np.abs(ri.distance_traveled[o] - lexp) / lexp
    assert err.max() < 1e-11, err.max()
    out["cone_worst_rel"] = float(err.max())
    # --- paraboloid z = x^2 + y^2, rays parallel to the axis: a = 0 exactly, b = -d_z, l = -c/b
    par = mk('paraboloid', 'quadsurface', [np.diag([1.0, 1.0, 0.0]), np.array([0, 0, -1.0]), 0.0], everywhere, 1.0, 1.0)
    m = 500
    xy = rng.uniform(-1.5, 1.5, (m, 2)); z0 = rng.uniform(5, 9, m)
    sp = np.column_stack([xy, z0]); d = np.tile([0, 0, -1.0], (m, 1))
    d[m // 2:, 0] = 1e-13                                         # |4ac/b^2| < 100 eps: still the linear branch (:93)
    res = trace(sp, rays_from(d), [absorbing_sphere(50.0), par], 1, 1e-9, 0.0, -1.0, True, False, False, **MATLAB)
    ri = res.ray_interfaces[0]; o = np.argsort(np.abs(ri.ray_index), kind="stable")
    lexp = z0 - np.sum(xy ** 2, 1)                                # distance straight down to the bowl
    assert np.all(np.abs(ri.surface_index[o]) == 2)
    err = np.abs(ri.distance_traveled[o] - lexp) / lexp
    assert err.max() < 1e-11, err.max()
    out["linear_branch_worst_rel"] = float(err.max())
    return out


# ------------------------------------------------------------------------------------------------ bulk absorption (a1)
def check_beer_lambert(trace, n=5000, seed=5):
    """MATLAB/RayTracer2.m:443-447: a ray that travels l through a medium of absorption length lam loses 1 - exp(-l/lam) to
    tally category 1 (bulk) and brings exp(-l/lam) to the absorbing wall (category 0).  Off-centre source, so l varies."""
    lam, R = 5.0, 7.0
    sl = [absorbing_sphere(R, 1.33, abslength_inside=lam, abslength_outside=np.inf, rayleigh_inside=np.inf, rayleigh_outside=np.inf)]
    sp, rays = cases.isotropic_rays(n, seed, origin=(1.5, -2.0, 0.5))
    res = trace(sp, rays, sl, 3, 1e-9, 1e-12, -1.0, False, True, False, **MATLAB)
    d = rays[:, 0:3] / np.linalg.norm(rays[:, 0:3], axis=1, keepdims=True)
    b = np.sum(sp * d, 1); c = np.sum(sp * sp, 1) - R * R
    l = -b + np.sqrt(b * b - c)
    t = res.absorption_table
    bulk = np.sum(1 - np.exp(-l / lam)); wall = np.sum(np.exp(-l / lam))
    assert abs(t[0, 1].sum() - bulk) < 1e-12 * n and abs(t[0, 0].sum() - wall) < 1e-12 * n
    assert abs(t.sum() - n) < 1e-12 * n and t[1:].sum() == 0      # everything ends at step 0
    return dict(bulk=float(t[0, 1].sum()), expected=float(bulk))


# ------------------------------------------------------------------------------------------------ diffuse (a11), retro (a12)
def _disk_scene(surface_type, absorption):
    disk = mk('disk', 'plane', [np.array([0, 0, 0.]), np.array([0, 0, 1.])], everywhere_plane, 1.0, np.inf, surface_type, absorption)
    return [absorbing_sphere(40.0), disk]


def _downward_rays(n, seed, max_angle=1.2):
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, max_angle, n); ph = rng.uniform(0, 2 * np.pi, n)
    d = np.column_stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), -np.cos(th)])
    sp = np.tile([0.0, 0.0, 5.0], (n, 1))
    return sp, d, th


def check_diffuse_lambert(trace, n=200000, seed=11):
    """MATLAB/RayTracer2.m:394-415: cos(theta_out) = sqrt(u) (Lambert: density 2 cos, <cos> = 2/3), azimuth uniform,
    independent of the incoming direction; the pseudo-normal handed to the Fresnel step mirrors the ray exactly there."""
    sp, d, th_in = _downward_rays(n, seed)
    rays = rays_from(d, stokes=(1, 0.3, -0.2, 0.1))
    res = trace(sp, rays, _disk_scene('diffuse', 0.3), 1, 1e-9, 0.0, -1.0, True, False, False, seed=99, **MATLAB)
    ri = res.ray_interfaces[0]
    assert ri.ray_index.size == n and np.all(np.abs(ri.surface_index) == 2)
    out = ri.reflected_ray
    assert np.abs(np.linalg.norm(out[:, 0:3], axis=1) - 1).max() < 1e-12
    ct = out[:, 2]
    assert ct.min() >= 0
    sig = np.sqrt((0.5 - 4.0 / 9.0) / n)
    assert abs(ct.mean() - 2.0 / 3.0) < 3 * sig, (ct.mean(), sig)
    chi2_ok(np.histogram(ct ** 2, bins=20, range=(0, 1))[0], np.full(20, n / 20.0), "diffuse cos^2")
    phi = np.arctan2(out[:, 1], out[:, 0])
    chi2_ok(np.histogram(phi, bins=16, range=(-np.pi, np.pi))[0], np.full(16, n / 16.0), "diffuse azimuth")
    order = np.argsort(np.abs(ri.ray_index), kind="stable")
    steep = th_in < 0.4; flat = th_in > 0.9
    ct_o = ct[order]
    for cut in (steep, flat):
        m = int(cut.sum())
        assert abs(ct_o[cut].mean() - 2.0 / 3.0) < 3 * np.sqrt((0.5 - 4.0 / 9.0) / m)
    assert np.abs(out[:, 6] - 0.7).max() < 1e-12 and np.abs(ri.refracted_ray[:, 6]).max() == 0   # conductor behind: (1 - absorption) reflected
    assert np.abs(ri.surface_normal - np.array([0, 0, 1.0])).max() < 1e-15                       # the record keeps the mechanical normal
    return dict(mean_cos=float(ct.mean()))


def check_retro(trace, n=2000, seed=12):
    """MATLAB/RayTracer2.m:332-333: the optical normal of a retro surface is -direction: the ray comes straight back"""
    sp, d, _ = _downward_rays(n, seed)
    rays = rays_from(d, stokes=(1, 0.3, -0.2, 0.1))
    res = trace(sp, rays, _disk_scene('retro', 0.1), 1, 1e-9, 0.0, -1.0, True, False, False, **MATLAB)
    ri = res.ray_interfaces[0]; o = np.argsort(np.abs(ri.ray_index), kind="stable")
    assert np.abs(ri.reflected_ray[o, 0:3] + d).max() < 1e-15
    assert np.abs(ri.reflected_ray[o, 6] - 0.9).max() < 1e-12
    assert np.abs(ri.surface_normal - np.array([0, 0, 1.0])).max() < 1e-15
    return {}


# ------------------------------------------------------------------------------------------------ Rayleigh (a10)
def rayleigh_expected(nc, nphi, f, phi0, n):
    """bin integrals of p(c, phi) = 3/(16 pi) [ (1 + c^2) - f (1 - c^2) cos 2(phi - phi0) ], the dipole law that
    RayleighScattering.m:84-126 samples (azimuth mixture :84-98, polar cubic :118-126)"""
    ce = np.linspace(-1, 1, nc + 1); pe = np.linspace(-np.pi, np.pi, nphi + 1)
    A = np.diff(ce + ce ** 3 / 3); B = np.diff(ce - ce ** 3 / 3)
    dphi = np.diff(pe); ds = np.diff(np.sin(2 * (pe - phi0)) / 2)
    return n * 3 / (16 * np.pi) * (A[:, None] * dphi[None, :] - f * B[:, None] * ds[None, :])


def check_rayleigh_law(trace, n=300000, seed=13):
    """Scattering angles against the dipole law for unpolarised, fully and partially linearly polarised light, the
    free path against exp(-l/lambda) (RayTracer2.m:430), the Stokes update against the Rayleigh Mueller matrix
    (RayleighScattering.m:136-147)."""
    lam_s, lam_a = 2.0, 50.0
    sl = [absorbing_sphere(1e4, 1.2, abslength_inside=lam_a, abslength_outside=np.inf, rayleigh_inside=lam_s, rayleigh_outside=np.inf)]
    out = {}
    for stokes in ((1, 0, 0, 0), (1, 1, 0, 0), (2, 0.6, -1.0, 0.4)):
        sp, rays = cases.isotropic_rays(n, seed, origin=(0, 0, 0))
        rays[:, 3:6] = np.random.default_rng(seed + 1).normal(size=(n, 3)); rays[:, 6:10] = stokes
        res = trace(sp, rays, sl, 1, 1e-9, 0.0, -1.0, True, False, False, seed=1234, **MATLAB)
        ri = res.ray_interfaces[0]
        assert ri.ray_index.size == n and np.all(ri.is_rayleigh == 1) and np.all(ri.surface_index == 0)
        o = np.argsort(np.abs(ri.ray_index), kind="stable")
        inc, sc, l = ri.incoming_ray[o], ri.reflected_ray[o], ri.distance_traveled[o]
        d = inc[:, 0:3]
        ref = rays[:, 3:6] - np.sum(rays[:, 3:6] * d, 1, keepdims=True) * d; ref /= np.linalg.norm(ref, axis=1, keepdims=True)
        perp = np.cross(d, ref)
        c = np.sum(sc[:, 0:3] * d, 1); phi = np.arctan2(np.sum(sc[:, 0:3] * perp, 1), np.sum(sc[:, 0:3] * ref, 1))
        s0, s1, s2, s3 = stokes
        f = np.hypot(s1, s2) / s0; phi0 = 0.5 * np.arctan2(s2, s1)
        H = np.histogram2d(c, phi, bins=(8, 12), range=((-1, 1), (-np.pi, np.pi)))[0]
        chi2, dof = chi2_ok(H, rayleigh_expected(8, 12, f, phi0, n), "rayleigh angles, stokes %r" % (stokes,))
        out[str(stokes)] = (round(chi2, 1), dof)
        # free path: exponential with the Rayleigh length; intensity on arrival: Beer-Lambert over that path
        edges = -lam_s * np.log(1 - np.linspace(0, 1, 21)[:-1]); edges = np.append(edges, np.inf)
        chi2_ok(np.histogram(l, bins=edges)[0], np.full(20, n / 20.0), "rayleigh free path")
        assert np.abs(inc[:, 6] - s0 * np.exp(-l / lam_a)).max() < 1e-12 * s0
        # Mueller matrix in the scattering plane: I' ~ (1+c^2) I + (c^2-1) Q, Q' ~ (c^2-1) I + (c^2+1) Q, U' ~ 2c U, V' ~ 2c V; I' renormalised to I
        Q = s1 * np.cos(2 * phi) + s2 * np.sin(2 * phi); U = -s1 * np.sin(2 * phi) + s2 * np.cos(2 * phi)
        tr = np.exp(-l / lam_a)
        Ip = (1 + c * c) * s0 + (c * c - 1) * Q
        exp_s = np.column_stack([s0 * tr, ((c * c - 1) * s0 + (c * c + 1) * Q) * s0 / Ip * tr, 2 * c * U * s0 / Ip * tr, 2 * c * s3 * s0 / Ip * tr])
        assert np.abs(sc[:, 6:10] - exp_s).max() < 1e-9 * s0
        # new polarisation reference: unit, normal to the new direction, in the scattering plane
        assert np.abs(np.sum(sc[:, 3:6] * sc[:, 0:3], 1)).max() < 1e-9 and np.abs(np.linalg.norm(sc[:, 3:6], axis=1) - 1).max() < 1e-9
        nplane = np.cross(d, sc[:, 0:3]); big = np.linalg.norm(nplane, axis=1) > 1e-3
        assert np.abs(np.sum(sc[big, 3:6] * nplane[big], 1) / np.linalg.norm(nplane[big], axis=1)).max() < 1e-9
    return out


# ------------------------------------------------------------------------------------------------ Fresnel (a8)
def check_fresnel_known_answers(trace):
    """Brewster angle (rp = 0), normal incidence ((n1-n2)/(n1+n2))^2, s-polarised reflectance at Brewster incidence,
    total internal reflection onset, and tir_handling >= 0 (RefractionReflectionAtInterface.m:221-225: the
    'refracted' ray is the reflected one times tir_handling)."""
    n1, n2 = 1.0, 1.5
    flat = mk('interface', 'plane', [np.array([0, 0, 0.]), np.array([0, 0, 1.])], everywhere_plane, n1, n2)
    thB = np.arctan(n2 / n1)

    def shoot(theta, stokes, from_below=False, tir=-1.0):
        # plane of incidence = xz; polref = y is normal to it, so s1 = +1 is s-polarised, s1 = -1 is p-polarised
        d = np.array([[np.sin(theta), 0, np.cos(theta) if from_below else -np.cos(theta)]])
        sp = np.array([[0.0, 0, -3.0 if from_below else 3.0]])
        res = trace(sp, rays_from(d, polref=[0, 1.0, 0], stokes=stokes), [absorbing_sphere(30.0), flat], 1, 1e-9, 0.0, tir, True, False, False, **MATLAB)
        ri = res.ray_interfaces[0]
        assert abs(ri.surface_index[0]) == 2
        return ri.refracted_ray[0], ri.reflected_ray[0]
    t, r = shoot(thB, (1, -1, 0, 0))                                  # p-polarised at Brewster: nothing reflected
    assert abs(r[6]) < 1e-15 and abs(t[6] - 1) < 1e-15                # (unit vectors are unit to an ulp: an s-polarised 1e-16 of the ray may reflect)
    t, r = shoot(thB, (1, 1, 0, 0))                                   # s-polarised: Rs = sin^2(ti - tt), ti + tt = 90 deg
    tt = np.arcsin(np.sin(thB) * n1 / n2)
    assert abs(r[6] - np.sin(thB - tt) ** 2) < 1e-14 and abs(t[6] + r[6] - 1) < 1e-15
    t, r = shoot(0.0, (1, 0.5, 0, 0))                                 # normal incidence
    assert abs(r[6] - 0.04) < 1e-15 and abs(t[6] - 0.96) < 1e-15 and np.abs(r[0:3] - [0, 0, 1.0]).max() < 1e-15
    t, r = shoot(0.7, (1, 0, 0, 0))                                   # Snell
    assert abs(np.hypot(t[0], t[1]) - np.sin(0.7) * n1 / n2) < 1e-15
    thc = np.arcsin(n1 / n2)                                          # from the glass side: critical angle
    t, r = shoot(thc - 1e-6, (1, 0.2, 0.1, 0.3), from_below=True)
    assert t[6] > 1e-3 and abs(t[6] + r[6] - 1) < 1e-14 and t[2] > 0 and t[2] < 3e-3   # still transmitted, grazing
    t, r = shoot(thc + 1e-6, (1, 0.2, 0.1, 0.3), from_below=True)
    assert abs(t[6]) < 1e-15 and abs(r[6] - 1) < 1e-14                # total reflection, tir_handling < 0: zero-intensity stub
    assert np.abs(t[0:3] - [1.0, 0, 0]).max() < 1e-12                 # ... along the surface (.m:217-219)
    assert abs(np.hypot(np.hypot(r[7], r[8]), r[9]) - np.hypot(np.hypot(0.2, 0.1), 0.3)) < 1e-14   # a pure phase: polarisation degree kept
    t, r = shoot(thc + 0.2, (1, 0.2, 0.1, 0.3), from_below=True, tir=0.5)
    assert np.abs(t[6:10] - 0.5 * r[6:10]).max() < 1e-15 and np.abs(t[0:3] - r[0:3]).max() < 1e-15 and abs(r[6] - 1) < 1e-14
    return {}


# ------------------------------------------------------------------------------------------------ UNIFIED sigma_alpha > 0 (a9)
def facet_alpha_expected(sig, edges, n):
    """polar angle of the micro-facet normal, UnifiedReflectorModel.m:283-303: |N(0, sig)| kept with probability
    min(sin(alpha) / min(4 sig, 1), 1) below pi/2 -> density ~ exp(-alpha^2 / 2 sig^2) min(sin(alpha)/fmax, 1)"""
    from scipy.integrate import quad
    fmx = min(4 * sig, 1.0)
    dens = lambda a: np.exp(-a * a / (2 * sig * sig)) * min(np.sin(a) / fmx, 1.0)  # noqa: E731
    w = np.array([quad(dens, edges[i], edges[i + 1], epsabs=1e-14)[0] for i in range(len(edges) - 1)])
    return n * w / quad(dens, 0, np.pi / 2, epsabs=1e-14)[0]


def check_unified_facet_law(trace, n=200000):
    """A perfectly conducting rough panel with lobe probability 1 mirrors every ray off its micro-facet, so the facet
    normal is the bisector of incoming and outgoing direction; its polar angle must follow the law above and its
    azimuth must be uniform.  Normal incidence (sigma 0.1 and 0.2) and 30 degrees (sigma 0.1: the wrong-side cut
    :301 and a second crossing are out of reach at 6 sigma)."""
    out = {}
    for sig, theta in ((0.1, 0.0), (0.2, 0.0), (0.1, np.pi / 6)):
        panel = mk('panel', 'plane', [np.array([0, 0, 0.]), np.array([0, 0, 1.])], everywhere_plane, 1.0, np.inf, 'unified', 0.0,
                   unifiedparams=[sig, 0.0, 1.0, 0.0, 0.0])
        d = np.tile([np.sin(theta), 0, -np.cos(theta)], (n, 1)); sp = np.tile([0.0, 0, 4.0], (n, 1))
        res = trace(sp, rays_from(d, polref=[0, 1.0, 0]), [absorbing_sphere(30.0), panel], 1, 1e-9, 0.0, -1.0, True, False, False, seed=77, **MATLAB)
        ri = res.ray_interfaces[0]
        assert ri.ray_index.size == n and np.all(np.abs(ri.surface_index) == 2)
        dout = ri.reflected_ray[:, 0:3]
        assert np.abs(ri.reflected_ray[:, 6] - 1).max() < 1e-12 and np.abs(np.linalg.norm(dout, axis=1) - 1).max() < 1e-12
        fn = dout - d; fn /= np.linalg.norm(fn, axis=1, keepdims=True)
        alpha = np.arccos(np.clip(fn[:, 2], -1, 1)); phi = np.arctan2(fn[:, 1], fn[:, 0])
        edges = np.append(np.linspace(0, 3.5 * sig, 29), np.pi / 2)
        chi2, dof = chi2_ok(np.histogram(alpha, bins=edges)[0], facet_alpha_expected(sig, edges, n), "facet polar angle, sigma %g" % sig)
        chi2_ok(np.histogram(phi, bins=16, range=(-np.pi, np.pi))[0], np.full(16, n / 16.0), "facet azimuth, sigma %g" % sig)
        out[(sig, round(theta, 3))] = (round(chi2, 1), dof)
    return out


def check_unified_dielectric_reemission(trace, n=120000):
    """UNIFIED sigma_alpha > 0 on a dielectric-dielectric panel (UnifiedReflectorModel.m:196-203, 257-266): light that a
    facet transmits is inside the surface material, comes back as Lambertian re-emission about the mechanical normal -
    unpolarised, times P_diffuse-reemit - and has to cross the rough boundary again on its way out.
    (i) Index-matched panel (n 1.33 | 1.33): no facet reflects and none bends, so EVERY ray is re-emitted once and leaves
    as it was re-emitted: Lambert's law about the panel normal (<cos> = 2/3, cos^2 and azimuth uniform), unpolarised, with
    exactly P_reemit of its intensity; the rest is the surface's absorption (tally 0).  Steep and 60 degree incidence.
    (ii) Vacuum | glass with P_reemit = 1, lobe only: whatever happens in between - reflection off the first facet,
    re-emission, total reflection back into the material and another re-emission - every ray comes back out with all of
    its intensity (reflected + absorbed = 1 per ray, absorbed = 0), upwards, a unit vector with a unit reference normal to it."""
    out = {}
    for theta, pre in ((0.2, 0.8), (np.pi / 3, 0.55)):
        panel = mk('matched panel', 'plane', [np.array([0, 0, 0.]), np.array([0, 0, 1.])], everywhere_plane, 1.33, 1.33, 'unified', 0.0,
                   unifiedparams=[0.15, pre, 0.3, 0.3, 0.2])
        d = np.tile([np.sin(theta), 0, -np.cos(theta)], (n, 1)); sp = np.tile([0.0, 0, 4.0], (n, 1))
        res = trace(sp, rays_from(d, polref=[0, 1.0, 0], stokes=(1, 0.6, -0.3, 0.2)), [absorbing_sphere(30.0, n_in=1.33), panel], 1, 1e-9, 0.0, -1.0, True, False, False,
                    seed=123, **MATLAB)
        ri = res.ray_interfaces[0]
        assert ri.ray_index.size == n and np.all(np.abs(ri.surface_index) == 2)
        o = ri.reflected_ray
        assert np.abs(o[:, 6] - pre).max() < 1e-12 and np.abs(o[:, 7:10]).max() == 0.0            # P_reemit of the ray, unpolarised
        assert np.abs(ri.refracted_ray[:, 6]).max() == 0.0                                          # nothing is transmitted through a UNIFIED surface (RayTracer2.py:444)
        c = o[:, 2]; phi = np.arctan2(o[:, 1], o[:, 0])
        assert c.min() > 0 and np.abs(np.linalg.norm(o[:, 0:3], axis=1) - 1).max() < 1e-12
        assert np.abs(np.sum(o[:, 0:3] * o[:, 3:6], axis=1)).max() < 1e-12 and np.abs(np.linalg.norm(o[:, 3:6], axis=1) - 1).max() < 1e-12
        assert abs(c.mean() - 2.0 / 3.0) < 3 * np.sqrt(1.0 / 18.0 / n), c.mean()                   # Lambert: p(c) = 2c, var = 1/18
        chi2, dof = chi2_ok(np.histogram(c * c, bins=20, range=(0, 1))[0], np.full(20, n / 20.0), "re-emission cos^2, incidence %.2f" % theta)
        chi2_ok(np.histogram(phi, bins=16, range=(-np.pi, np.pi))[0], np.full(16, n / 16.0), "re-emission azimuth, incidence %.2f" % theta)
        tab = res.absorption_table
        assert abs(tab[0, 0, 1, :].sum() - (1 - pre) * n) < 1e-9 * n                                # absorbed in the panel: the rest of every ray
        out[round(theta, 2)] = (round(chi2, 1), dof)
    glass = mk('glass panel', 'plane', [np.array([0, 0, 0.]), np.array([0, 0, 1.])], everywhere_plane, 1.0, 1.5, 'unified', 0.0,
               unifiedparams=[0.1, 1.0, 1.0, 0.0, 0.0])
    m = n // 4
    d = np.tile([np.sin(0.5), 0, -np.cos(0.5)], (m, 1)); sp = np.tile([0.0, 0, 4.0], (m, 1))
    res = trace(sp, rays_from(d, polref=[0, 1.0, 0], stokes=(1, 0.5, 0.5, 0)), [absorbing_sphere(30.0), glass], 1, 1e-9, 0.0, -1.0, True, False, False, seed=321, **MATLAB)
    o = res.ray_interfaces[0].reflected_ray
    assert o.shape[0] == m and np.abs(o[:, 6] - 1).max() < 1e-12 and o[:, 2].min() > 0
    assert np.abs(np.linalg.norm(o[:, 0:3], axis=1) - 1).max() < 1e-12 and np.abs(np.linalg.norm(o[:, 3:6], axis=1) - 1).max() < 1e-9
    assert np.abs(np.sum(o[:, 0:3] * o[:, 3:6], axis=1)).max() < 1e-9
    assert np.all(np.hypot(np.hypot(o[:, 7], o[:, 8]), o[:, 9]) <= o[:, 6] * (1 + 1e-12))           # physical Stokes vectors
    assert abs(res.absorption_table[0, 0, 1, :].sum()) < 1e-9 * m                                  # nothing absorbed: reflected + absorbed = 1 per ray
    mostly_reemitted = np.mean(np.abs(o[:, 7]) < 0.45)                                              # a first-facet reflection keeps |s1| ~ 0.5, re-emitted light is polarised by its exit alone
    assert 0.85 < mostly_reemitted < 0.99, mostly_reemitted                                          # ~ 1 - R(0.5 rad) of a 1 -> 1.5 interface
    out["glass"] = round(float(mostly_reemitted), 3)
    return out


def check_raytable_rule(trace, n=400):
    """output_raytable (MATLAB/RayTracer2.m:249-253, 543-548; the numpy port's is broken) on SURVEY config 1 in
    ray-splitting mode: layer 1 holds the starting points and the normalised rays; layer k + 1 holds, for every ray whose
    lineage has never been reflected (ray_index > 0) and that scattered in step k, the point of that scatter and its
    refracted child; everything else stays zero.  The expected table is built here from the records of the same call
    (pinned by the reference's fixtures) - by the rule as the MATLAB lines state it, not by the wrapper's code."""
    sp, rays = cases.isotropic_rays(n, 4, origin=(3, 0, 0))
    rays[:, 0:3] *= 2.5                                                                              # not normalised on purpose (:244-245)
    K = 7
    res = trace(sp, rays.copy(), cases.sphere_plane(), K, 1e-9, 1e-6, -1.0, True, False, True, **MATLAB)
    rt = np.asarray(res.raytable)
    assert rt.shape == (K + 1, n, 13)
    want = np.zeros_like(rt)
    want[0, :, 0:3] = sp; want[0, :, 3:13] = rays
    want[0, :, 3:6] /= np.linalg.norm(want[0, :, 3:6], axis=1, keepdims=True); want[0, :, 6:9] /= np.linalg.norm(want[0, :, 6:9], axis=1, keepdims=True)
    rows = 0
    for k, ri in enumerate(res.ray_interfaces):
        cut = ri.ray_index > 0
        ix = np.round(ri.ray_index[cut]).astype(np.int64) - 1                                       # 1-based MATLAB indices
        assert np.unique(ix).size == ix.size                                                        # a never-reflected lineage is one ray per step
        want[k + 1, ix, 0:3] = ri.intersection_point[cut]; want[k + 1, ix, 3:13] = ri.refracted_ray[cut]
        rows += ix.size
    assert rows > 1.1 * n and len(res.ray_interfaces) >= 4 and np.any(want[2])                      # (step 0: every ray; step 1: the rays that leave the sphere towards the plane)
    assert np.array_equal(rt[0], want[0]) or np.abs(rt[0] - want[0]).max() < 1e-15
    assert np.array_equal(rt[1:], want[1:])
    return {"rows": rows, "steps": len(res.ray_interfaces)}


def check_torus_first_hit_in_frame(trace):
    return check_torus_first_hit(trace, seed=8, coaxial=True)


ALL_CHECKS = (check_torus_first_hit, check_torus_first_hit_in_frame, check_quadric_closed_forms, check_beer_lambert, check_diffuse_lambert, check_retro, check_rayleigh_law,
              check_fresnel_known_answers, check_unified_facet_law, check_unified_dielectric_reemission)
