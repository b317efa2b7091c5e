"""Monte-Carlo parity by the north_star's criterion: per-surface hit histograms (and per-surface absorbed intensity) of
two runs with INDEPENDENT random streams must agree within 3 sigma (chi-square).

A run is split into batches of rays; per batch, step band and cell (surface, orientation; bulk scatters are a cell of
their own) we keep the number of interactions and the intensity absorbed there (absorption_table categories 0 + 1).
Hits of one ray are correlated (light bounces between jar and reflector dozens of times), so the variance of a cell
is not Poisson: it is estimated from the batch-to-batch scatter of both runs, the difference of the means becomes a
Welch t statistic per cell, converted to a normal deviate z, and chi2 = sum z^2 over the cells with enough hits.
"""
import numpy as np

BANDS = ((0, 6), (6, 21), (21, 100))        # step bands: the deterministic-looking start, the middle, the long tail


def batch_stats(ray_interfaces, table, n_surf):
    """(hits [bands, n_surf + 1, 2], absorbed [bands, n_surf, 2]) of one batch; records of step k = ray_interfaces[k]"""
    hits = np.zeros((len(BANDS), n_surf + 1, 2)); absorbed = np.zeros((len(BANDS), n_surf, 2))
    assert len(ray_interfaces) == table.shape[0]
    for k, ri in enumerate(ray_interfaces):
        b = next(i for i, (lo, hi) in enumerate(BANDS) if lo <= k < hi)
        ray = np.asarray(ri.is_rayleigh) != 0
        sn = np.asarray(ri.surface_number).astype(int); o = (np.asarray(ri.orientation) < 0).astype(int)
        np.add.at(hits[b], (np.where(ray, n_surf, sn), np.where(ray, 0, o)), 1.0)
        absorbed[b] += table[k, 0] + table[k, 1]
    return hits, absorbed


def chi2_between(xa, xb, min_mean=40.0):
    """xa, xb: [batches, ...cells] of two independent runs.  Returns (chi2, n_cells, z [cells])"""
    from scipy import stats
    a = xa.reshape(xa.shape[0], -1); b = xb.reshape(xb.shape[0], -1)
    keep = (a.mean(0) + b.mean(0)) / 2 >= min_mean
    a = a[:, keep]; b = b[:, keep]
    na, nb = a.shape[0], b.shape[0]
    va = a.var(0, ddof=1) / na; vb = b.var(0, ddof=1) / nb
    t = (a.mean(0) - b.mean(0)) / np.sqrt(va + vb)
    dof = (va + vb) ** 2 / (va ** 2 / (na - 1) + vb ** 2 / (nb - 1))
    z = stats.norm.isf(stats.t.sf(np.abs(t), dof)) * np.sign(t)       # same tail probability under the normal law
    return float(np.sum(z * z)), int(keep.sum()), z


def assert_parity(xa, xb, label, min_mean=40.0):
    chi2, n, z = chi2_between(xa, xb, min_mean)
    lim = n + 3 * np.sqrt(2 * n)
    assert chi2 < lim, "%s: chi2 %.1f over %d cells exceeds the 3 sigma limit %.1f (worst |z| %.1f)" % (label, chi2, n, lim, np.abs(z).max())
    return chi2, n
