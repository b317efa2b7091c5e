"""Cycle sums per phase of the persistent kernel k_trace (builds with -DRT_PROFILE only): python tools/phase_profile_trace.py [rays]
phases: refill, nearest hit, physics, tallies + child; then wave passes and live lanes per pass"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch  # noqa
import bench
from raytracer_amd import _lib
from raytracer_amd.RayTracer2 import Context
sl, prog, origin, kw = bench.build_workload("sbc_full_mc")
ctx = Context(0); ctx.set_program(*prog)
org = (C.c_double * 3)(*origin); n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
out = (C.c_ulonglong * 16)()
for rep in range(2):
    ctx.L.rt_debug_profile(ctx.h, out, 1)
    prm = ctx.params(100, _lib.FLAG_SINGLECHILD, 0, 1e-6, (1e-4, 1e-4), -1.0, 5, 0)
    ctx._check(ctx.L.rt_trace_point_source(ctx.h, org, n, C.byref(prm)))
    bk = (C.c_double * 3)(); bn = (C.c_int32 * 3)(); ctx.L.rt_result_timing_by_kernel(ctx.h, bk, bn)
    ctx.L.rt_debug_profile(ctx.h, out, 0)
v = [int(x) for x in out]
tot = sum(v[0:4]) or 1
print(" ".join("%s %.1f%%" % (nm, 100.0 * v[i] / tot) for i, nm in enumerate(["refill", "hit", "phys", "tally"])), "| wave passes %d, live lanes per pass %.1f, cycles per pass %.0f" % (v[5], v[6] / max(1, v[5]), tot / max(1, v[5])))
print("k_trace %.2f ms" % bk[0])
