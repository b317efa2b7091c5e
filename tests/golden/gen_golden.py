"""Generate the golden fixtures in tests/golden/ by importing the numpy reference from /root/reference/python.

Run in the build container only (the reference never travels to the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
Shims applied in this process only: np.round_ = np.round (removed in NumPy 2), stdout swallowed
(the reference prints per call, Q13).  Only paths that RUN in the numpy port are exercised:
plane/sphere/cylinder with rays starting inside closed shapes, 'normal' surfaces, Fresnel with tir < 0,
dice and no-dice modes, UNIFIED with sigma_alpha = 0.
"""
import contextlib
import io
import os
import sys

import numpy as np

np.round_ = np.round
import numpy.matlib  # noqa: E402,F401

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, "/root/reference/python")
sys.dont_write_bytecode = True

import cases  # noqa: E402
import RayTracer2 as REF  # noqa: E402
import RayToCylinder  # noqa: E402
import RayToPlane  # noqa: E402
import RayToSphere  # noqa: E402
import RefractionReflectionAtInterface as RRI  # noqa: E402
import surface as refsurface  # noqa: E402


def run_ref(sp, rays, sl, *args):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = REF.RayTracer2(sp.copy(), rays.copy(), sl, *args)
    return out


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


def main():
    S = refsurface.surface
    # G1: config 1, no-dice, 256 rays, everything
    sp, rays = cases.isotropic_rays(256, 0, origin=(3, 0, 0))
    ri, tab, _ = run_ref(sp, rays, cases.sphere_plane(S), [10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False)
    save("g1_cfg1_nodice_256.npz", sp=sp, rays=rays, table=tab, **cases.pack_interfaces(ri))

    # G1p: same with polarised input
    sp, rays = cases.isotropic_rays(256, 5, origin=(3, 0, 0), polarized=True)
    ri, tab, _ = run_ref(sp, rays, cases.sphere_plane(S), [6], [1e-9], np.array([1e-5, 1e-5]), [], True, False, False)
    save("g1p_cfg1_nodice_pol_256.npz", sp=sp, rays=rays, table=tab, **cases.pack_interfaces(ri))

    # G2: config 1 at 1e5 rays, summary only
    sp, rays = cases.isotropic_rays(100000, 0, origin=(3, 0, 0))
    ri, tab, _ = run_ref(sp, rays, cases.sphere_plane(S), [10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False)
    nper = np.array([len(r.ray_index) for r in ri])
    psum = np.array([r.intersection_point.sum(0) for r in ri])
    s0sum = np.array([[r.incoming_ray[:, 6].sum(), r.refracted_ray[:, 6].sum(), r.reflected_ray[:, 6].sum()] for r in ri])
    nplane = np.array([np.sum(np.abs(r.surface_index) == 1) for r in ri])
    save("g2_cfg1_nodice_1e5_summary.npz", table=tab, nper=nper, psum=psum, s0sum=s0sum, nplane=nplane)

    # G3: TestGeometry with 'normal' cylinders, no-dice
    sp, rays = cases.isotropic_rays(200, 1, origin=(0, 0, 7))
    ri, tab, _ = run_ref(sp, rays, cases.testgeometry(S, cyl_type='normal'), [8], [1e-9], np.array([1e-4, 1e-4]), [], True, False, False)
    save("g3_testgeom_normal_nodice_200.npz", sp=sp, rays=rays, table=tab, **cases.pack_interfaces(ri))

    # G4: TestGeometry as shipped (UNIFIED [0,0,0,0,0] cylinders), dice mode; only step 0 geometry is deterministic
    sp, rays = cases.isotropic_rays(2000, 2, origin=(0, 0, 7))
    np.random.seed(11)
    ri, tab, _ = run_ref(sp, rays, cases.testgeometry(S), 10)
    save("g4_testgeom_unified_dice_2000.npz", sp=sp, rays=rays, table=tab, **cases.pack_interfaces(ri[:1]),
         nper=np.array([len(r.ray_index) for r in ri]))

    # G5: UnifiedTest, 2e5 rays, dice: hemisphere hit distribution of the second interface
    n = 200000
    sp, rays = cases.unifiedtest_rays(n)
    np.random.seed(12)
    ri, tab, _ = run_ref(sp, rays, cases.unifiedtest(S), 10)
    p = ri[1].intersection_point
    save("g5_unifiedtest_2e5.npz", table=tab, n=np.array(n), hits=p.astype(np.float32), s0=ri[1].incoming_ray[:, 6].astype(np.float32),
         nper=np.array([len(r.ray_index) for r in ri]))

    # G9: UNIFIED dielectric panel (Lambert re-emission path), sigma_alpha = 0
    n = 100000
    sp, rays = cases.unifiedtest_rays(n)
    np.random.seed(13)
    sl = cases.unifiedtest(S, params=(0, .8, .2, .3, .1), n_panel_in=1.5, n_med=1.0)
    ri, tab, _ = run_ref(sp, rays, sl, 10)
    p = ri[1].intersection_point
    save("g9_unified_dielectric_1e5.npz", table=tab, n=np.array(n), hits=p.astype(np.float32), s0=ri[1].incoming_ray[:, 6].astype(np.float32),
         nper=np.array([len(r.ray_index) for r in ri]))

    # G8: config 1 in dice mode, statistics
    sp, rays = cases.isotropic_rays(20000, 3, origin=(3, 0, 0))
    np.random.seed(14)
    ri, tab, _ = run_ref(sp, rays, cases.sphere_plane(S), [10], [1e-9], np.array([1e-6, 1e-6]), [], True, True, False)
    save("g8_cfg1_dice_2e4.npz", table=tab, nper=np.array([len(r.ray_index) for r in ri]),
         nplane=np.array([np.sum(np.abs(r.surface_index) == 1) for r in ri]))

    # G6: Fresnel single-call vectors
    rng = np.random.default_rng(21)
    m = 600
    d = rng.normal(size=(m, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    nrm = rng.normal(size=(m, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    flip = np.sum(d * nrm, 1) > 0
    nrm[flip] *= -1  # normals face the ray, as RayTracer2 provides them
    nrm[-20:] = -d[-20:]  # normal incidence
    pol = rng.normal(size=(m, 3))
    s0 = rng.uniform(.5, 2, m)
    sv = rng.normal(size=(m, 3)); sv /= np.linalg.norm(sv, axis=1, keepdims=True); sv *= (s0 * rng.uniform(0, .95, m))[:, None]
    rays = np.concatenate([d, pol, s0[:, None], sv], axis=1)
    n1 = np.tile([1.0, 1.5, 1.5, 1.33], m // 4); n2 = np.tile([1.5, 1.0, np.inf, 1.0], m // 4)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        refr, refl = RRI.RefractionReflectionAtInterface(rays.copy(), nrm.copy(), n1.copy(), n2.copy(), -1)
    save("g6_fresnel.npz", rays=rays, normals=nrm, n1=n1, n2=n2, refr=refr, refl=refl)

    # G7: intersections (first root only for sphere/cylinder from outside: Q16; both roots from inside)
    m = 400
    spi = rng.uniform(-4, 4, size=(m, 3))
    di = rng.normal(size=(m, 3)); di /= np.linalg.norm(di, axis=1, keepdims=True)
    out = {"sp": spi, "d": di}
    for name, fn, pl in (("plane", RayToPlane.RayToPlane, [np.array([.5, -1., 5.]), np.array([.2, -.3, 1.])]),
                         ("sphere", RayToSphere.RayToSphere, [np.array([.3, .2, -.1]), 10.]),
                         ("cylinder", RayToCylinder.RayToCylinder, [np.array([.1, .2, .3]), np.array([.3, .2, 1.]), 10.])):
        with np.errstate(all="ignore"):
            p, nr, l, o = fn(spi.copy(), di.copy(), *pl)
        out[name + "_p"] = np.asarray(p); out[name + "_n"] = np.asarray(nr); out[name + "_l"] = np.asarray(l); out[name + "_o"] = np.asarray(o)
    save("g7_rayto.npz", **out)


if __name__ == "__main__":
    main()
