"""Level path against the persistent kernel by run mode and size (which `level_min_rays` each mode wants): the bench
workload's jar, point source; modes: tallies only (lean persistent instances), final rows kept on the device, records to
the host, ray splitting with tallies only.  python tools/level_threshold_modes.py > profiles/r3_level_threshold.txt"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench  # noqa: E402
from raytracer_amd import _lib  # noqa: E402
from raytracer_amd.RayTracer2 import Context  # noqa: E402


def main():
    sl, prog, origin, kw = bench.build_workload("sbc_full_mc")
    org = (C.c_double * 3)(*origin)
    modes = (("tallies only", _lib.FLAG_SINGLECHILD, 100), ("final rows on the device", _lib.FLAG_SINGLECHILD | _lib.FLAG_FINAL_ROWS, 100),
             ("records", _lib.FLAG_SINGLECHILD | _lib.FLAG_FULL_OUTPUT, 20), ("ray splitting, tallies only", 0, 12), ("ray splitting, records", _lib.FLAG_FULL_OUTPUT, 8), ("ray splitting, final rows on the device", _lib.FLAG_FINAL_ROWS, 12))
    for name, flags, K in modes:
        print("== %s (K = %d)" % (name, K))
        for e in ((13, 15, 17) if (flags & _lib.FLAG_FULL_OUTPUT and not flags & _lib.FLAG_SINGLECHILD) else (13, 15, 17, 19)):
            n = 1 << e
            row = []
            for lmr in (1 << 40, 0):   # persistent kernel only / level path from the first step
                ctx = Context(0); ctx.set_program(*prog); ctx.set_option("level_min_rays", lmr); ctx.set_option("lanes", 1)
                dev = ctx.final_rows_device(n) if flags & _lib.FLAG_FINAL_ROWS else None
                best = None
                for r in range(3):
                    prm = ctx.params(K, flags, 0, kw["min_travel"], kw["thr"], kw["tir"], 5, 0)
                    t0 = time.perf_counter()
                    ctx._check(ctx.L.rt_trace_point_source(ctx.h, org, n, C.byref(prm)))
                    dt = time.perf_counter() - t0
                    live, scat = ctx.step_counts(K)
                    if r and (best is None or dt < best[0]):
                        best = (dt, int(scat.sum()))
                if dev is not None:
                    ctx.final_rows_host()
                ctx.close()
                row.append(best)
            print("  2^%d rays: persistent kernel %.3g interactions/s, level path %.3g (x %.2f)" % (e, row[0][1] / row[0][0], row[1][1] / row[1][0], (row[1][1] / row[1][0]) / (row[0][1] / row[0][0])), flush=True)


if __name__ == "__main__":
    main()
