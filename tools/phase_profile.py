import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from raytracer_amd import _lib
from raytracer_amd.RayTracer2 import Context
sl, prog, origin, kw = bench.build_workload("sbc_full_mc")
ctx = Context(0); ctx.set_program(*prog)
org = (C.c_double*3)(*origin); n = 1 << 23
out = (C.c_ulonglong * 16)()
for K in (1, 3, 100):
    ctx.L.rt_debug_profile(ctx.h, out, 1)
    prm = ctx.params(K, _lib.FLAG_SINGLECHILD, 0, 1e-6, (1e-4, 1e-4), -1.0, 5, 0)
    ctx._check(ctx.L.rt_trace_point_source(ctx.h, org, n, C.byref(prm)))
    ms = C.c_double(); nl = C.c_int32(); ctx.L.rt_result_timing(ctx.h, C.byref(ms), C.byref(nl))
    ctx.L.rt_debug_profile(ctx.h, out, 0)
    v = [int(x) for x in out]
    tot = sum(v[:4])
    print("K=%d: %.2f ms; wave passes %d, mean live lanes %.1f; cycles/pass: refill %.0f hit %.0f phys %.0f tally+out %.0f (total %.0f)" % (
        K, ms.value, v[5], v[6] / max(v[5], 1), v[0] / max(v[5], 1), v[1] / max(v[5], 1), v[2] / max(v[5], 1), v[3] / max(v[5], 1), tot / max(v[5], 1)))
    print("   torus: solver calls/pass %.2f, lanes wanting a solve per pass %.2f (of %d torus tests x live lanes)" % (v[4] / max(v[5], 1), v[7] / max(v[5], 1), 4))
    print("   candidate blocks (point + bounds program): %.1f per pass, %.0f cycles per pass (%.0f each)" % (v[9] / max(v[5], 1), v[8] / max(v[5], 1), v[8] / max(v[9], 1)))
    print("   torus prefilter+push %.0f cycles per pass, flush (solve + deferred candidates) %.0f cycles per pass" % (v[11] / max(v[5], 1), v[10] / max(v[5], 1)))
