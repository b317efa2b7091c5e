"""The drop-in's default call end to end (python/RayTracer2.py:185 with full_output=True: VERDICT r2 item 5): the
UnifiedTest geometry (python/UnifiedTest.py:15-47: 45 degree rays onto a UNIFIED panel under an absorbing hemisphere),
host arrays in, every rayInterfaces array out, in the reference's row order.

    python tools/bench_dropin.py [--rays 5000000] > profiles/r3_dropin.json

Prints one JSON line: records/s of the whole call and where the time goes (normalisation of the caller's rays, the
library call = upload + kernels + record copies, ordering, splitting the rows into the 17 fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=5000000)
    ap.add_argument("--numpy-host-work", action="store_true", help="the host work as in round 2: numpy normalisation and lexsort ordering")
    a = ap.parse_args()
    import torch  # noqa: F401
    import cases
    from raytracer_amd import RayTracer2 as R
    sl = cases.unifiedtest()
    n = a.rays
    sp, rays = cases.unifiedtest_rays(n)
    t = {}

    def timed(name, fn):
        def w(*args, **kw):
            t0 = time.perf_counter(); r = fn(*args, **kw); t[name] = t.get(name, 0.0) + time.perf_counter() - t0; return r
        return w
    if a.numpy_host_work:
        R.order_groups = R.order_records
        R.NUMPY_NORMALISE = True
    R.order_groups = timed("ordering", R.order_groups)
    R.Context.interfaces = timed("fields", R.Context.interfaces)
    best = None
    for rep in range(3):
        t.clear()
        r_in = rays.copy()
        c = R.get_context(0)
        if rep == 0:
            c.L.rt_trace = timed("library call (upload, kernels, record copies)", c.L.rt_trace)
        t0 = time.perf_counter()
        ri, tab, _ = R.RayTracer2(sp, r_in, sl, seed=3 + rep)
        dt = time.perf_counter() - t0
        nrec = sum(len(x.ray_index) for x in ri)
        rec = {"rays": n, "records": nrec, "seconds": dt, "records_per_s": nrec / dt, "parts_s": dict(t)}
        rec["parts_s"]["rest (argument checks, normalisation of the caller's rays, geometry key, results)"] = dt - sum(t.values())
        if best is None or rec["seconds"] < best["seconds"]:
            best = rec
        del ri
    best["host_work"] = "numpy (round 2)" if a.numpy_host_work else "library (threaded C)"
    print(json.dumps(best))


if __name__ == "__main__":
    main()
