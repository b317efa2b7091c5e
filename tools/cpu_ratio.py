"""Build container only (the reference never travels to the GPU box): times the numpy reference RayTracer2, imported
unmodified from /root/reference/python, and the CPU oracle (oracle/rt_oracle.c) on identical seeded inputs of
SURVEY 8d configs 1 and 2 - the ratio that ties bench.py's same-node `cpu_baseline` (oracle) back to the real reference
(CPU-baseline plan).  python tools/cpu_ratio.py"""
import contextlib
import io
import os
import sys
import time

import numpy as np

np.round_ = np.round
import numpy.matlib  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, "/root/reference/python")
sys.dont_write_bytecode = True
import cases  # noqa: E402
import RayTracer2 as REF  # noqa: E402
import surface as refsurface  # noqa: E402
from oracle import oracle as O  # noqa: E402


def timed(f, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); out = f(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def main():
    n = 100000
    rows = []
    # config 1: dielectric sphere + absorbing plane, no-dice, K=10, thresholds 1e-6
    sp, rays = cases.isotropic_rays(n, 0, origin=(3, 0, 0))
    args1 = ([10], [1e-9], np.array([1e-6, 1e-6]), [], True, False, False)
    # config 2 geometry: UnifiedTest, dice mode, K=10
    sp2, rays2 = cases.unifiedtest_rays(n)
    for name, s, r, geo, args in (("config 1 (no-dice)", sp, rays, cases.sphere_plane, args1), ("config 2 geometry (dice)", sp2, rays2, cases.unifiedtest, (10,))):
        def ref():
            with contextlib.redirect_stdout(io.StringIO()):
                return REF.RayTracer2(s.copy(), r.copy(), geo(refsurface.surface), *args)
        t_ref, out = timed(ref)
        inter = sum(len(ri.ray_index) for ri in out[0])
        res = {}
        for th in (1, 8):
            os.environ["OMP_NUM_THREADS"] = str(th)
            t_or, o = timed(lambda: O.trace(s, r.copy(), geo(), *args, compat="python-strict", seed=1))
            res[th] = t_or
            inter_o = sum(len(ri.ray_index) for ri in o.ray_interfaces)
        rows.append((name, inter, inter_o, t_ref, res[1], res[8]))
    for name, inter, inter_o, t_ref, t1, t8 in rows:
        print("%-26s %8d interactions (oracle %8d): numpy reference %.3f s = %.3g /s | oracle 1 thread %.3f s = %.3g /s (x%.1f) | 8 threads %.3f s = %.3g /s (x%.1f)"
              % (name, inter, inter_o, t_ref, inter / t_ref, t1, inter_o / t1, t_ref / t1 * inter_o / inter, t8, inter_o / t8, t_ref / t8 * inter_o / inter))


if __name__ == "__main__":
    main()
