"""Generates tests/golden/g10_mc_hist_<workload>.npz: per-batch hit / absorbed-intensity histograms of the bench
workload from the CPU ORACLE (oracle/rt_oracle.c, the product's compiled bounds program standing in for the Python
lambdas so that 2e5 rays finish in minutes) with its own seed.  tests/test_gpu_mc_chi2.py compares the HIP path, run
with a different seed, against these by chi-square (tests/mc_hist.py).  Run here, on the CPU:

    python tests/golden/gen_mc_hist.py sbc_full_mc sbc_full_mc_diffuse
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, ".."))
import bench          # noqa: E402
import mc_hist        # noqa: E402
from oracle import oracle as O   # noqa: E402

BATCHES, PER_BATCH, SEED = 20, 10000, 777001


def point_source_rays(n, rng, origin):
    """isotropic unpolarised unit rays, polref = x^ x d (RunSBCGeometry.m:45-68), from numpy's generator: the oracle's
    inputs are independent of the device's in-kernel source"""
    ct = 1 - 2 * rng.random(n); st = np.sqrt(1 - ct * ct); ph = 2 * np.pi * rng.random(n)
    rays = np.zeros((n, 10)); rays[:, 0] = st * np.cos(ph); rays[:, 1] = st * np.sin(ph); rays[:, 2] = ct
    pol = np.cross(np.array([1.0, 0, 0]), rays[:, 0:3]); pol /= np.linalg.norm(pol, axis=1, keepdims=True)
    rays[:, 3:6] = pol; rays[:, 6] = 1
    return np.tile(np.asarray(origin, float), (n, 1)), rays


def main():
    for name in sys.argv[1:]:
        sl, prog, origin, kw = bench.build_workload(name)
        S = prog[0].shape[0]
        rng = np.random.default_rng(SEED)
        H = []; A = []; t0 = time.time()
        for b in range(BATCHES):
            sp, rays = point_source_rays(PER_BATCH, rng, origin)
            r = O.trace(sp, rays, [None] * S, kw["K"], kw["min_travel"], kw["thr"], kw["tir"], True, True, False, compat="matlab",
                        seed=SEED + b, ray_id_base=b * PER_BATCH, program=prog)
            h, a = mc_hist.batch_stats(r.ray_interfaces, r.absorption_table, S)
            H.append(h); A.append(a)
            print("%s batch %d/%d: %d interactions, %.0f s" % (name, b + 1, BATCHES, int(r.n_scat.sum()), time.time() - t0), flush=True)
        np.savez_compressed(os.path.join(HERE, "g10_mc_hist_%s.npz" % name), hits=np.array(H), absorbed=np.array(A), seed=SEED, per_batch=PER_BATCH,
                            workload=name, bands=np.array(mc_hist.BANDS))


if __name__ == "__main__":
    main()
