"""Cycle sums per phase of k_level_phys (builds with -DRT_PROFILE only): python tools/phase_profile_level.py [rays]
phases: 0 set-up to first barrier, 1 sort, 2 barrier after sort, 3 loop body (issue + physics), 4 wait for the next record, 5 final barrier, 6 flush"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch  # noqa
import bench
from raytracer_amd import _lib
from raytracer_amd.RayTracer2 import Context
sl, prog, origin, kw = bench.build_workload("sbc_full_mc")
ctx = Context(0); ctx.set_program(*prog)
org = (C.c_double * 3)(*origin); n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
out = (C.c_ulonglong * 16)()
names = ["setup", "sort", "sort-barrier", "body", "wait-next", "end-barrier", "flush"]
for rep in range(2):
    ctx.L.rt_debug_profile(ctx.h, out, 1)
    prm = ctx.params(100, _lib.FLAG_SINGLECHILD, 0, 1e-6, (1e-4, 1e-4), -1.0, 5, 0)
    ctx._check(ctx.L.rt_trace_point_source(ctx.h, org, n, C.byref(prm)))
    bk = (C.c_double * 3)(); bn = (C.c_int32 * 3)(); ctx.L.rt_result_timing_by_kernel(ctx.h, bk, bn)
    ctx.L.rt_debug_profile(ctx.h, out, 0)
v = [int(x) for x in out]
for lab, o in (("PASS 1", 0), ("PASS 2", 8)):
    tot = sum(v[o:o + 7]) or 1
    print(lab, " ".join("%s %.1f%%" % (names[i], 100.0 * v[o + i] / tot) for i in range(7)), "| total wave-cycles %.3g" % tot)
print("hit %.1f ms phys %.1f ms" % (bk[1], bk[2]))
