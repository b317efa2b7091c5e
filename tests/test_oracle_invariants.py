"""Known-answer tests of the CPU oracle for the sections no executable reference pins:
torus roots against numpy.roots, quadric closed forms and the linear_cut branch, Beer-Lambert, Lambert <cos> = 2/3,
the Rayleigh dipole law, Brewster / normal incidence / TIR onset / tir_handling >= 0, the UNIFIED facet-angle law.
tests/test_gpu_invariants.py puts the same questions to the HIP path."""
import numpy as np
import pytest

import invariants as inv
from oracle import oracle as O


def oracle_trace(sp, rays, sl, *args, **kw):
    return O.trace(sp, rays.copy(), sl, *args, **kw)


@pytest.mark.parametrize("check", inv.ALL_CHECKS, ids=lambda f: f.__name__)
def test_oracle_known_answers(check):
    print(check.__name__, check(oracle_trace))


def test_oracle_raytable_rule():
    print(inv.check_raytable_rule(oracle_trace))


def test_torus_all_roots_and_classification_vs_numpy_roots():
    """unit level (ora_ray_to_shape): all four roots of 10 200 rays, real/complex classification included, against
    numpy.roots of the coefficients of RayToTorus.m:97-103; near-tangent rays (two roots within 1e-6) excluded, counted"""
    sp, d = inv.torus_rays(10000, 17)
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    T = inv.TORUS
    p, nr, l, o, isreal = O.ray_to_shape("torus", [T["center"], T["axis"], T["r1"], T["r2"]], sp, d)
    real, tangent = inv.roots_reference(inv.torus_quartic(T["center"], T["axis"], T["r1"], T["r2"], sp, d))
    ok = ~tangent
    assert tangent.sum() < 0.03 * sp.shape[0]
    mine = np.where(isreal.astype(bool), l, np.nan); mine = np.sort(mine, axis=1)          # NaN last, like `real`
    assert np.array_equal(np.isnan(mine[ok]), np.isnan(real[ok])), "real/complex classification differs"
    m = ok[:, None] & ~np.isnan(real)
    err = np.abs(mine[m] - real[m]) / np.maximum(1.0, np.abs(real[m]))
    assert err.max() < 1e-9, err.max()
    print("torus roots: %d rays, %d excluded near tangency, worst rel %.2e" % (sp.shape[0], tangent.sum(), err.max()))
    # normals at the real roots: unit length, on the torus (RayToTorus.m:149-158)
    ax = T["axis"] / np.linalg.norm(T["axis"])
    for k in range(4):
        r = isreal[:, k].astype(bool) & ok
        x = p[r, k] - T["center"]; ya = x @ ax; u = x - ya[:, None] * ax
        assert np.abs((np.linalg.norm(u, axis=1) - T["r1"]) ** 2 + ya ** 2 - T["r2"] ** 2).max() < 1e-7
        assert np.abs(np.linalg.norm(nr[r, k], axis=1) - 1).max() < 1e-8
        assert np.all(np.sum(nr[r, k] * d[r], 1) <= 1e-12)   # turned against the ray


def test_rayleigh_unit_level_law():
    """ora_rayleigh on 10^6 rays: same law as the loop-level check, tighter"""
    n = 1000000
    rng = np.random.default_rng(3)
    rays = np.zeros((n, 10)); rays[:, 2] = 1; rays[:, 3] = 1; rays[:, 6:10] = (1, -0.5, 0.5, 0.1)
    out = O.rayleigh(rays, seed=5)
    c = out[:, 2]; phi = np.arctan2(out[:, 1], out[:, 0])
    f = np.hypot(-0.5, 0.5); phi0 = 0.5 * np.arctan2(0.5, -0.5)
    H = np.histogram2d(c, phi, bins=(10, 16), range=((-1, 1), (-np.pi, np.pi)))[0]
    inv.chi2_ok(H, inv.rayleigh_expected(10, 16, f, phi0, n), "ora_rayleigh")
    del rng
