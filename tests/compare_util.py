"""Tolerance-aware comparison of two RayTracer2 results (reference order, record by record)."""
import numpy as np

import cases


def near_normal(ri, eps=1e-6):
    """rays hitting at (numerically) normal incidence: the plane of incidence, hence the polarisation reference of the
    children, is rounding noise there in the reference itself"""
    d = ri.incoming_ray[:, 0:3]; n = ri.surface_normal
    c = np.cross(d, n)
    return np.linalg.norm(c, axis=1) < eps


def compare_results(a, b, tol=1e-9, check_table=True, label="", growth=1.0, tol_max=1e-6):
    """growth > 1: tolerance of step k is tol*growth**k (rounding differences are amplified by every reflection off a
    curved surface; the strict 1e-9 of the north_star is kept for the config-1 style tests)"""
    assert len(a.ray_interfaces) == len(b.ray_interfaces), (len(a.ray_interfaces), len(b.ray_interfaces))
    worst = 0.0
    for k, (x, y) in enumerate(zip(a.ray_interfaces, b.ray_interfaces)):
        assert x.ray_index.shape == y.ray_index.shape, (k, x.ray_index.shape, y.ray_index.shape)
        assert np.array_equal(x.ray_index, y.ray_index), k
        assert np.array_equal(x.surface_index + 0., y.surface_index + 0.), k
        nn = near_normal(x)
        for f in cases.RI_FIELDS:
            u = np.asarray(getattr(x, f), float).copy(); v = np.asarray(getattr(y, f), float).reshape(u.shape).copy()
            if f in ("refracted_ray", "reflected_ray"):
                u[nn, 3:6] = 0; v[nn, 3:6] = 0; u[nn, 7:9] = 0; v[nn, 7:9] = 0
            assert np.array_equal(np.isnan(u), np.isnan(v)), (k, f)
            m = np.isfinite(u)
            assert np.array_equal(u[~m & ~np.isnan(u)], v[~m & ~np.isnan(u)]), (k, f)
            if m.any():
                e = np.max(np.abs(u[m] - v[m]) / np.maximum(1, np.abs(u[m])))
                worst = max(worst, e)
                assert e < min(tol * growth ** k, tol_max), (label, k, f, e)
    if check_table:
        assert a.absorption_table.shape == b.absorption_table.shape
        e = np.max(np.abs(a.absorption_table - b.absorption_table) / np.maximum(1, np.abs(a.absorption_table))) if a.absorption_table.size else 0
        assert e < min(tol * growth ** len(a.ray_interfaces), tol_max), (label, "table", e)
    return worst
