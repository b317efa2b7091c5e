"""Interactions/s of the bench workload against rays per step (and against an option of the context), one process, one
context: the data behind the batch size of bench.py and the per-step host sync of the level path.

    python tools/sweep_rays.py [--workload sbc_full_mc] [--lo 15] [--hi 26] [--opt seg_per_cu=6,8,9,12] > profiles/r2_sweep.json
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sbc_full_mc")
    ap.add_argument("--lo", type=int, default=15); ap.add_argument("--hi", type=int, default=26)
    ap.add_argument("--opt", default="", help="name=v1,v2,...: repeat the largest size for every value of this context option")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--set", default="", help="name=value[,name=value...]: context options for the whole sweep")
    a = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime per process: see raytracer_amd/_lib.py)
    import bench
    from raytracer_amd import _lib
    from raytracer_amd.RayTracer2 import Context
    sl, prog, origin, kw = bench.build_workload(a.workload)
    ctx = Context(0); ctx.set_program(*prog)
    for kv in filter(None, a.set.split(",")):
        ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    org = (C.c_double * 3)(*origin)
    flags = kw.get("flags", _lib.FLAG_SINGLECHILD)

    def run(n, reps):
        best = None
        for r in range(reps + 1):                       # first one warms up (buffers, code objects)
            prm = ctx.params(kw["K"], flags, 0, kw["min_travel"], kw["thr"], kw["tir"], a.seed, r * n)
            t0 = time.perf_counter()
            ctx._check(ctx.L.rt_trace_point_source(ctx.h, org, n, C.byref(prm)))
            dt = time.perf_counter() - t0
            live, scat = ctx.step_counts(kw["K"])
            ms = C.c_double(); nl = C.c_int32(); ctx.L.rt_result_timing(ctx.h, C.byref(ms), C.byref(nl))
            bk = (C.c_double * 3)(); bn = (C.c_int32 * 3)(); ctx.L.rt_result_timing_by_kernel(ctx.h, bk, bn)
            rec = {"rays": n, "interactions": int(scat.sum()), "wall_s": dt, "kernel_ms": ms.value, "launches": nl.value,
                   "interactions_per_s": scat.sum() / dt, "interactions_per_kernel_s": scat.sum() / (ms.value / 1e3) if ms.value else None,
                   "ms_by_kernel": {"k_trace": bk[0], "k_level_hit": bk[1], "k_level_phys": bk[2]}, "launches_by_kernel": [bn[0], bn[1], bn[2]]}
            if r > 0 and (best is None or rec["wall_s"] < best["wall_s"]):
                best = rec
        return best
    out = {"workload": a.workload, "sizes": [], "options": []}
    for e in range(a.lo, a.hi + 1):
        n = 1 << e
        rec = run(n, 3 if e <= 22 else 2)
        out["sizes"].append(rec)
        print("2^%d rays: %.3g interactions/s wall, %.3g by kernel time, %d launches" % (e, rec["interactions_per_s"], rec["interactions_per_kernel_s"], rec["launches"]), file=sys.stderr, flush=True)
    if a.opt:
        name, vals = a.opt.split("=")
        for v in vals.split(","):
            ctx.set_option(name, float(v))
            rec = run(1 << min(a.hi, 25), 2); rec[name] = float(v)
            out["options"].append(rec)
            print("%s=%s: %.4g interactions/s, hit %.1f ms, phys %.1f ms" % (name, v, rec["interactions_per_s"], rec["ms_by_kernel"]["k_level_hit"], rec["ms_by_kernel"]["k_level_phys"]), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
