"""Synthetic end-point clouds for the LED-to-camera matching tests (final rows of include/rt_mi355x.h)."""
import numpy as np


def rows_from(points, dirs, normals, six, s0=None, nint=None):
    n = points.shape[0]
    r = np.zeros((n, 32))
    r[:, 0:3] = points; r[:, 3:6] = dirs; r[:, 6:9] = [0, 0, 1]; r[:, 9] = 1.0 if s0 is None else s0
    r[:, 13:16] = normals; r[:, 16] = six; r[:, 17] = np.where(np.asarray(six) < 0, -1.0, 1.0); r[:, 18] = np.arange(n)
    r[:, 19] = 1 if nint is None else nint
    return r


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def cloud(n, rng, surfaces, extent=20.0, lattice=None):
    """n end points on random surfaces out of `surfaces` (random sign), directions at random; with `lattice` the
    coordinates are multiples of it (many exact ties and zero distances)."""
    p = rng.uniform(-extent, extent, (n, 3))
    if lattice:
        p = np.round(p / lattice) * lattice
    nrm = unit(rng.normal(size=(n, 3)))
    d = unit(rng.normal(size=(n, 3)))
    d = np.where((np.sum(d * nrm, axis=1) > 0)[:, None], -d, d)      # arriving against the normal
    six = rng.choice(surfaces, n) * rng.choice([-1, 1], n)
    return rows_from(p, d, nrm, six, s0=rng.uniform(0.1, 1.0, n))


class FR(object):
    pass


def final_rays_of(rows):
    """finalRays-style arrays for the oracle from final rows (rays without interaction: surface_index NaN)"""
    fr = FR()
    fr.intersection_point = rows[:, 0:3].copy(); fr.incoming_ray = rows[:, 3:13].copy(); fr.surface_normal = rows[:, 13:16].copy()
    hit = (rows[:, 19] > 0) & np.all(np.isfinite(rows[:, 0:3]), axis=1)
    fr.surface_index = np.where(hit, rows[:, 16], np.nan)
    cosi = -np.sum(fr.incoming_ray[:, 0:3] * fr.surface_normal, axis=1)
    fr.theta_incident = np.arccos(np.clip(cosi, -1, 1))
    return fr
