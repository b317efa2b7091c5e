"""Record streaming rate: a full_output trace of the benchmark's jar geometry whose records
(408 B per interaction) leave through a record sink, chunk by chunk.

    python tools/bench_records.py [--rays 3000000] [--K 10] [--stage-gb 2]

Prints one JSON line: records/s and GB/s at the C boundary for
  serial      copy and sink between the kernels of two chunks (the path up to r1k, "stream_overlap" 0)
  overlapped  packed records, pinned stages, copy on a second stream ("stream_overlap" 1, the default)
each with a sink that returns at once and with one that reads every byte handed to it (numpy sum over the rows), and
for comparison the accumulating run (no sink; records fetched with rt_result_records afterwards).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=3000000)
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--stage-gb", type=float, default=2.0)
    ap.add_argument("--workload", default="sbc_full_mc_blackcones")
    a = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime per process)
    import bench
    from raytracer_amd import _lib
    from raytracer_amd.RayTracer2 import Context
    sl, prog, origin, kw = bench.build_workload(a.workload)
    ctx = Context(0)
    ctx.set_program(*prog)
    org = (C.c_double * 3)(*origin)
    flags = _lib.FLAG_SINGLECHILD | _lib.FLAG_FULL_OUTPUT
    REC = _lib.REC_COLS
    out = {"workload": a.workload, "rays": a.rays, "K": a.K, "stage_gb": a.stage_gb, "bytes_per_record": REC * 8 + 24}

    def run(overlap, touch, use_sink=True):
        state = {"n": 0, "calls": 0, "sum": 0.0}

        def cb(user, n, rec, step, ray, parent, fl, first, nrays):
            state["n"] += int(n); state["calls"] += 1
            if touch:
                state["sum"] += float(np.ctypeslib.as_array(rec, shape=(int(n), REC)).sum())
            return 0
        fn = _lib.RECORD_SINK(cb)
        ctx.set_option("stream_overlap", overlap)
        ctx.set_option("stream_stage_gb", a.stage_gb)
        # serial: the same chunk size as the overlapped run (whose stage is the smaller of a third of the workspace and stream_stage_gb)
        ctx.set_option("workspace_gb", 32.0 if (overlap or not use_sink) else a.stage_gb)
        if use_sink:
            ctx._check(ctx.L.rt_set_record_sink(ctx.h, fn, None))
        best = None
        for rep in range(2):  # the first run allocates the pinned stages and the pooled device buffers
            state.update(n=0, calls=0, sum=0.0)
            prm = ctx.params(a.K, flags, 0, kw["min_travel"], kw["thr"], kw["tir"], 11, rep * a.rays)
            t0 = time.perf_counter()
            ctx._check(ctx.L.rt_trace_point_source(ctx.h, org, a.rays, C.byref(prm)))
            if not use_sink:
                res = ctx.results(True, False, a.rays)
                state["n"] = res["rec"].shape[0]
                if touch:
                    state["sum"] = float(res["rec"].sum())
            dt = time.perf_counter() - t0
            ms = C.c_double(); nl = C.c_int32(); ctx.L.rt_result_timing(ctx.h, C.byref(ms), C.byref(nl))
            best = {"records": state["n"], "chunks": state["calls"], "seconds": dt, "records_per_s": state["n"] / dt,
                    "GB_per_s": state["n"] * (REC * 8 + 24) / dt / 1e9, "kernel_ms": ms.value}
        if use_sink:
            ctx._check(ctx.L.rt_set_record_sink(ctx.h, C.cast(None, _lib.RECORD_SINK), None))
        return best
    out["serial_noop_sink"] = run(0, False)
    out["overlapped_noop_sink"] = run(1, False)
    out["serial_reading_sink"] = run(0, True)
    out["overlapped_reading_sink"] = run(1, True)
    ctx.set_option("release_workspace", 1)
    acc_rays = min(a.rays, 3000000)
    if acc_rays == a.rays:
        out["accumulate_then_fetch"] = run(1, False, use_sink=False)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
