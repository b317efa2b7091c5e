"""The chi-square Monte-Carlo parity test of tests/test_gpu_mc_chi2.py with the host build of the device header
(tests/hostsim) in place of the kernels: the per-ray arithmetic the GPU runs, seed A, against the oracle fixture made
with seed B - and the control: a 2 % error in the reflector's absorption must not pass."""
import os
import sys

import numpy as np
import pytest

import hostsim_util as Hs
import mc_hist

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)


def host_stats(workload, absorption=None, seed=424242):
    import bench
    from gen_mc_hist import point_source_rays
    from raytracer_amd import geometries as G
    g = np.load(os.path.join(GOLD, "g10_mc_hist_%s.npz" % workload))
    sl, prog, origin, kw = bench.build_workload(workload)
    if absorption is not None:
        specs = G.sbc_full_mc_specs(); specs.reflector_absorption = absorption; sl = G.create_arbc_geometry(specs)
    S = prog[0].shape[0]; per = int(g["per_batch"])
    rng = np.random.default_rng(seed); H = []; A = []
    for b in range(g["hits"].shape[0]):
        sp, rays = point_source_rays(per, rng, origin)
        r = Hs.trace(sp, rays, sl, kw["K"], kw["min_travel"], kw["thr"], kw["tir"], True, True, compat="matlab", seed=seed + b, ray_id_base=b * per)
        h, a = mc_hist.batch_stats(r.ray_interfaces, r.absorption_table, S); H.append(h); A.append(a)
    return np.array(H), np.array(A), g


@pytest.mark.parametrize("workload", ["sbc_full_mc", "sbc_full_mc_diffuse"])
def test_device_arithmetic_agrees_with_the_oracle_within_3_sigma(workload):
    H, A, g = host_stats(workload)
    for b, band in enumerate(mc_hist.BANDS):
        mc_hist.assert_parity(H[:, b], g["hits"][:, b], "%s hits %r" % (workload, band))
        mc_hist.assert_parity(A[:, b], g["absorbed"][:, b], "%s absorbed %r" % (workload, band), min_mean=1.0)


def test_two_percent_error_is_caught():
    H, A, g = host_stats("sbc_full_mc", absorption=0.102)
    failed = 0
    for b in range(len(mc_hist.BANDS)):
        chi2, n, z = mc_hist.chi2_between(A[:, b], g["absorbed"][:, b], 1.0)
        failed += chi2 > n + 3 * np.sqrt(2 * n)
    assert failed == len(mc_hist.BANDS)
