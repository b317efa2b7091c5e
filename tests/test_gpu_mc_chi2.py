"""Monte-Carlo parity on the benchmark workload by the north_star's criterion: the HIP path run
with seed A against the CPU oracle run with seed B (tests/golden/g10_mc_hist_*.npz, made by tests/golden/gen_mc_hist.py) -
independent random streams and independent input rays, so the chaos of the long bounce chains is beside the point.
Per step band (0-5, 6-20, 21-99) and per (surface, orientation) cell: number of interactions and absorbed intensity,
chi-square over all cells within 3 sigma (tests/mc_hist.py: variances from the batch-to-batch scatter).  A control with
the reflector's absorption off by 2 % (0.100 -> 0.102) must fail the same test."""
import os

import numpy as np
import pytest

import mc_hist

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gpu_stats(sl, S, origin, kw, batches, per_batch, seed):
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    import sys
    sys.path.insert(0, GOLD)
    from gen_mc_hist import point_source_rays
    rng = np.random.default_rng(seed)
    H = []; A = []
    c = Context(0)
    try:
        for b in range(batches):
            c.set_option("level_min_rays", 0 if b % 2 == 0 else 1 << 15)      # even batches: level-synchronous kernels; odd: the persistent kernel
            sp, rays = point_source_rays(per_batch, rng, origin)
            ri, tab, _ = RayTracer2(sp, rays, sl, kw["K"], kw["min_travel"], kw["thr"], kw["tir"], True, True, False, compat="matlab",
                                    seed=seed + b, ray_id_base=b * per_batch, ctx=c)
            h, a = mc_hist.batch_stats(list(ri), tab, S)
            H.append(h); A.append(a)
    finally:
        c.close()
    return np.array(H), np.array(A)


@pytest.mark.parametrize("workload", ["sbc_full_mc", "sbc_full_mc_diffuse"])
def test_per_surface_histograms_agree_with_the_oracle_within_3_sigma(workload):
    import bench
    g = np.load(os.path.join(GOLD, "g10_mc_hist_%s.npz" % workload))
    sl, prog, origin, kw = bench.build_workload(workload)
    S = prog[0].shape[0]
    batches, per_batch = g["hits"].shape[0], int(g["per_batch"])
    H, A = gpu_stats(sl, S, origin, kw, batches, per_batch, seed=424242)
    for b, band in enumerate(mc_hist.BANDS):
        c1 = mc_hist.assert_parity(H[:, b], g["hits"][:, b], "%s hits, steps %d-%d" % (workload, band[0], band[1] - 1))
        c2 = mc_hist.assert_parity(A[:, b], g["absorbed"][:, b], "%s absorbed intensity, steps %d-%d" % (workload, band[0], band[1] - 1), min_mean=1.0)
        print(workload, band, "hits chi2/cells", c1, "absorbed chi2/cells", c2)
    # totals: interactions per ray
    assert abs(H.sum() - g["hits"].sum()) < 3 * np.sqrt(batches * (H.sum(axis=(1, 2, 3)).var(ddof=1) + g["hits"].sum(axis=(1, 2, 3)).var(ddof=1)))


def test_a_two_percent_error_on_one_surface_fails_the_test():
    import bench
    from raytracer_amd import geometries as G
    g = np.load(os.path.join(GOLD, "g10_mc_hist_sbc_full_mc.npz"))
    _, prog, origin, kw = bench.build_workload("sbc_full_mc")
    specs = G.sbc_full_mc_specs(); specs.reflector_absorption = 0.102
    sl = G.create_arbc_geometry(specs)
    H, A = gpu_stats(sl, prog[0].shape[0], origin, kw, g["hits"].shape[0], int(g["per_batch"]), seed=424242)
    with pytest.raises(AssertionError):
        for b in range(len(mc_hist.BANDS)):
            mc_hist.assert_parity(A[:, b], g["absorbed"][:, b], "control", min_mean=1.0)
            mc_hist.assert_parity(H[:, b], g["hits"][:, b], "control")
