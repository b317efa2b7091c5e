"""bench.py -- headline benchmark: ray-surface interactions/s on the SBC/ArBC jar geometry, full Monte-Carlo mode.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch: `--rays` isotropic unit-intensity rays from a point source inside
the target fluid of the 39-surface jar geometry (MATLAB/Examples/CreateExamples/CreateArBCGeometry.m, with the bulk
Rayleigh/absorption lengths of CreateSBCGeometry.m:40-41 and a UNIFIED reflector: SURVEY.md 8d config 5), traced with
the arguments of MATLAB/Examples/RunExamples/RunSBCGeometry.m:73-77 (max_scatters 100, min_travel 1e-6, threshold
1e-4, tir -1, full_output off, single child).  Rays are generated on the device, so inputs are HBM-resident by
construction.  Rays are independent: every rank traces its own batch (weak scaling), no data-path collective; the only
collective is the reduction of the timing/interaction counters (and, in a real run, of the absorption table).
metric value = interactions (rows the reference would append to ray_interfaces) per second over all ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic flops per ray-surface test / per interaction, SURVEY.md section 8a/8d convention
# (add/mul 1, fma 2, div/sqrt/transcendental 1)
FLOP_SHAPE = {0: 23 + 10, 1: 85 + 20, 2: 130 + 20, 3: 175 + 20, 4: 560 + 40}
FLOP_PHYSICS = 300 + 40          # Fresnel + bookkeeping per interaction
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector FP64 (vendor figure: 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)


def flop_per_interaction(table):
    return float(sum(FLOP_SHAPE[int(s)] for s in table[:, 0]) + FLOP_PHYSICS)


def csrc_sha256():
    """hash of the kernel sources: profiles/counters.json is only believed if it was made from these"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "raytracer_amd", "csrc", "*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build_workload(name):
    from raytracer_amd import geometries as G
    from raytracer_amd import table as T
    if name in ("sbc_full_mc", "sbc_full_mc_blackcones", "sbc_full_mc_diffuse"):  # _diffuse: SURVEY 8d config 5 with 'diffuse' reflectors
        sl = G.create_arbc_geometry(G.sbc_full_mc_specs(black_cones=name.endswith("blackcones"), reflector="diffuse" if name.endswith("diffuse") else "unified"))
        origin = (0.0, 0.0, 10.0)
        kw = dict(K=100, min_travel=1e-6, thr=(1e-4, 1e-4), tir=-1.0)
    elif name == "arbc_camera":  # camera rays through the jar, python/BubbleImageCreator.py:129
        sl = G.create_arbc_geometry(G.ArBCSpecs())
        origin = None
        kw = dict(K=18, min_travel=1e-5, thr=(1e-5, 100.0), tir=-1.0)
    elif name == "arbc_nodice":  # ray splitting (no dice), the ray count grows ~13x before the thresholds bite
        sl = G.create_arbc_geometry(G.ArBCSpecs())
        origin = (0.0, 0.0, -5.0)
        kw = dict(K=18, min_travel=1e-5, thr=(1e-5, 1e-5), tir=-1.0, flags=0)
    elif name == "sbc_lightcollection":
        sl = G.create_sbc_geometry()
        origin = (0.0, 0.6, 1.3)
        kw = dict(K=100, min_travel=1e-6, thr=(1e-4, 1e-4), tir=-1.0)
    else:
        raise SystemExit("unknown workload " + name)
    prog = T.compile_surfaces(sl)
    return sl, prog, origin, kw


def cpu_baseline(prog, origin, kw, seed, target_seconds=12.0):
    """CPU restatement (oracle/rt_oracle.c, compiled bounds program, OpenMP) on a bounded sample of the same workload."""
    from oracle import oracle as O
    from raytracer_amd.RayTracer2 import Context
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    n = 20000
    done = 0; t_used = 0.0; inter = 0
    rng = np.random.default_rng(seed)
    while t_used < target_seconds and done < 40:
        ct = 1 - 2 * rng.random(n); st = np.sqrt(1 - ct * ct); ph = 2 * np.pi * rng.random(n)
        rays = np.zeros((n, 10)); rays[:, 0] = st * np.cos(ph); rays[:, 1] = st * np.sin(ph); rays[:, 2] = ct
        pol = np.cross(np.array([1.0, 0, 0]), rays[:, 0:3]); pol /= np.linalg.norm(pol, axis=1, keepdims=True)
        rays[:, 3:6] = pol; rays[:, 6] = 1
        sp = np.tile(np.asarray(origin, float), (n, 1))
        t0 = time.perf_counter()
        r = O.trace(sp, rays, [None] * prog[0].shape[0], kw["K"], kw["min_travel"], kw["thr"], kw["tir"], False, True, False,
                    compat="matlab", seed=seed + done, program=prog)
        t_used += time.perf_counter() - t0
        inter += int(r.n_scat.sum()); done += 1
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": inter / t_used, "unit": "interactions/s", "cores": threads, "cpu": cpu, "kind": "port",
            "sample": "%d rays (%d interactions) of the same workload through oracle/rt_oracle.c, %.1f s" % (done * n, inter, t_used),
            "note": "the builder's own C restatement (OpenMP), not a timing of the reference: the reference's numpy RayTracer2 cannot run this geometry "
                    "(torus and quadsurface are never imported, python/RayToShape.py:23,27; parity of this figure with the reference is unpinned). The only link "
                    "to the real reference is at S = 2 (SURVEY configs 1-2, tools/cpu_ratio.py, build container): there the numpy reference makes 4.9e5 and "
                    "4.8e5 interactions/s on one core and this restatement 3.9x and 5.9x that on one thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays", type=int, default=1 << 25, help="rays per step per GPU")
    ap.add_argument("--workload", default="sbc_full_mc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--host-inputs", action="store_true", help="hand the rays over as host arrays (N x 3, N x 10), as the Python boundary does: "
                    "the PCIe-inclusive rate quoted in DESIGN.md, never the headline value")
    ap.add_argument("--lanes", type=int, default=0, help="rt_set_option lanes (0: the library's default, 2): large tally-only runs are traced as two "
                    "halves at the same time on two streams that take turns with the nearest-hit launch, so one lane's nearest-hit launch (bound by "
                    "vector issue) runs beside the other's physics launches (memory, latency).  1: one lane, every launch alone on the GPU")
    ap.add_argument("--no-one-lane", "--no-two-lanes", dest="no_one_lane", action="store_true", help="skip the informational one-lane pass behind the "
                    "timed region (profiling runs: its launches would be averaged into the per-kernel statistics)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: raytracer_amd has no CPU path")
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 path on a box with fewer GPUs than ranks
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)

    import ctypes as C
    from raytracer_amd import _lib
    from raytracer_amd.RayTracer2 import Context
    sl, prog, origin, kw = build_workload(a.workload)
    ctx = Context(local)
    ctx.set_program(*prog)
    if a.lanes != 0:
        ctx.set_option("lanes", a.lanes)
    lanes = a.lanes if a.lanes != 0 else 2  # include/rt_mi355x.h: the default
    S = prog[0].shape[0]
    flags = kw.get("flags", _lib.FLAG_SINGLECHILD)
    cam = None
    if origin is None:  # one ray per pixel of a 3:4 sensor with about --rays pixels
        from raytracer_amd import geometries as Gm
        h = max(1, int(round((a.rays * 0.75) ** 0.5))); w = max(1, a.rays // h)
        cam = Gm.arbc_camera(Gm.ArBCSpecs(cam_resolution=np.array([h, w])))
        a.rays = h * w
    else:
        org = (C.c_double * 3)(*origin)

    h_sp = h_rays = None
    if a.host_inputs:
        if cam is not None:
            raise SystemExit("--host-inputs is for the point-source workloads")
        h_sp = np.zeros((a.rays, 3)); h_rays = np.zeros((a.rays, 10))
        dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        ctx._check(ctx.L.rt_generate_point_source(ctx.h, org, a.rays, C.c_uint64(a.seed), 0, dp(h_sp), dp(h_rays)))

    by_kernel = [[0.0, 0], [0.0, 0], [0.0, 0]]  # k_trace, k_level_hit, k_level_phys: ms, launches
    hit_units = [0]
    table_sum = [None]

    def step(i):
        prm = ctx.params(kw["K"], flags, 0, kw["min_travel"], kw["thr"], kw["tir"], a.seed, (rank * (a.steps + a.warmup) + i) * a.rays)
        if cam is not None:
            ctx._check(ctx.L.rt_trace_camera(ctx.h, C.byref(cam), C.byref(prm)))
        elif h_sp is not None:
            ctx._check(ctx.L.rt_trace(ctx.h, dp(h_sp), dp(h_rays), a.rays, C.byref(prm)))
        else:
            ctx._check(ctx.L.rt_trace_point_source(ctx.h, org, a.rays, C.byref(prm)))
        live, scat = ctx.step_counts(kw["K"])
        if dist is not None:  # the one collective of the path: the tallies of all ranks, all-reduced from the device table as it is (multigpu.allreduce_table does the same)
            tt = ctx.table_device(kw["K"])
            if backend != "nccl":
                tt = tt.cpu()
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            # conservation on the all-reduced table (RunSBCGeometry.m:87-92): absorbed + escaped + dropped over all steps, plus
            # what is still followed after the last one = the rays of all ranks
            rows = tt.reshape(tt.shape[0], 5, -1).sum(-1).cpu().numpy(); nz = np.nonzero(rows.any(axis=1))[0]
            table_sum[0] = float(rows[:, 0:4].sum() + (rows[nz[-1], 4] if nz.size else 0.0))
        ms = C.c_double(); nl = C.c_int32(); ctx.L.rt_result_timing(ctx.h, C.byref(ms), C.byref(nl))
        bk = (C.c_double * 3)(); bn = (C.c_int32 * 3)(); ctx.L.rt_result_timing_by_kernel(ctx.h, bk, bn)
        for j in range(3):
            by_kernel[j][0] += bk[j]; by_kernel[j][1] += bn[j]
        hit_units[0] += int(live[:bn[1]].sum())  # rays the nearest-hit launches of this trace went through
        return int(scat.sum()), ms.value

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    by_kernel[:] = [[0.0, 0], [0.0, 0], [0.0, 0]]
    hit_units[0] = 0
    barrier()
    t0 = time.perf_counter()
    inter = 0; kms = 0.0
    for i in range(a.steps):
        n, ms = step(a.warmup + i)
        inter += n; kms += ms
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt, float(inter), kms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); inter_all = float(tsum[1]); kms_max = float(tmax[2])
    else:
        inter_all = float(inter); kms_max = kms
    if rank == 0:
        fpi = flop_per_interaction(prog[0])
        fph = fpi - FLOP_PHYSICS  # nearest-hit search only: per ray and step, all surfaces
        # the whole path by the same convention: over the wall time of the timed region (the launches of two lanes overlap:
        # the sum of their event-bracketed times is longer than the step)
        path = (inter * fpi / dt) / 1e12 if dt > 0 else 0.0
        names = ("k_trace", "k_level_hit", "k_level_phys")
        hit_ms, hit_n = by_kernel[1]
        # dominant kernel: k_level_hit (nearest hit over all surfaces for the live rays of one step, one launch per step)
        if hit_n > 0 and hit_ms > 0:
            ach = (hit_units[0] * fph / (hit_ms / 1e3)) / 1e12
            dom = {"kernel": "k_level_hit", "launches": hit_n, "avg_launch_ms": hit_ms / hit_n, "units_per_launch": hit_units[0] / hit_n,
                   "flop_per_unit": fph, "unit_name": "ray-step (one live ray tested against all %d surfaces)" % S}
        else:  # level path off (RT_LEVEL_STEPS=0): the persistent kernel does everything
            ach = path
            dom = {"kernel": "k_trace", "launches": by_kernel[0][1], "avg_launch_ms": by_kernel[0][0] / max(by_kernel[0][1], 1),
                   "units_per_launch": inter / max(by_kernel[0][1], 1), "flop_per_unit": fpi, "unit_name": "interaction"}
        # Counter-measured figures of the dominant kernel come from rocprofv3 --pmc passes of this same command
        # (tools/profile_counters.sh -> profiles/counters.json); they are used only if that file was made from the kernel
        # sources as they are now (sha256 of raytracer_amd/csrc) on this workload, and are null otherwise.
        traffic = counter_tflops = valu_issue_frac = None
        cpath = os.path.join(ROOT, "profiles", "counters.json")
        if os.path.exists(cpath):
            try:
                t = json.load(open(cpath))
                if t.get("workload") == a.workload and t.get("src_sha256") == csrc_sha256():
                    kk = t["kernels"].get(dom["kernel"], {})
                    traffic = kk.get("hbm_bytes_per_unit") * dom["units_per_launch"] if kk.get("hbm_bytes_per_unit") is not None else None
                    counter_tflops = kk.get("fp64_flop_per_unit") * dom["units_per_launch"] / (dom["avg_launch_ms"] / 1e3) / 1e12 if kk.get("fp64_flop_per_unit") is not None else None
                    valu_issue_frac = kk.get("valu_issue_frac")
            except Exception:
                traffic = counter_tflops = valu_issue_frac = None
        out = {
            "metric": "ray-surface interactions/sec (node) on SBC geometry", "value": inter_all / dt, "unit": "interactions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("%s: %d-surface jar geometry, %s, " + ("MC single-child" if flags & _lib.FLAG_SINGLECHILD else "ray splitting (no dice)") + ", K=%d, thr=%g, %d rays/step/GPU, full_output off")
                                   % (a.workload, S, ("point source" if h_sp is None else "point-source rays handed over as HOST arrays (PCIe-inclusive)") if cam is None
                                      else "camera rays generated on the device", kw["K"], kw["thr"][0], a.rays),
                       "rays_per_step_per_gpu": a.rays, "lanes": lanes, "surfaces": S, "interactions_per_ray": inter / (a.steps * a.rays),
                       "ray_surface_tests_per_s": inter_all * S / dt},
            "roofline": dict({"bound": "fp64_valu", "achieved": counter_tflops if counter_tflops is not None else ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": (counter_tflops if counter_tflops is not None else ach) / FP64_VALU_PEAK_TFLOPS,
                              "basis": "executed" if counter_tflops is not None else "algorithmic (profiles/counters.json is not from these kernel sources or not of this workload: no executed-flop figure)",
                              "algorithmic_achieved": ach, "algorithmic_frac": ach / FP64_VALU_PEAK_TFLOPS,
                              "traffic": traffic, "counter_tflops": counter_tflops, "valu_issue_frac": valu_issue_frac,
                              "note": "achieved/frac = FP64 flops the kernel EXECUTED per second (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes, FMA = 2, per unit "
                                      "from the rocprofv3 --pmc passes of this command in profiles/counters.json, x units per launch / measured launch time) over "
                                      "the FP64 vector peak. algorithmic_* follow the flop CONVENTION of SURVEY.md 8a/8d (every surface counted for every ray; the "
                                      "frame program exists to skip most of that work, so it is not executed work and can exceed 1). valu_issue_frac = "
                                      "SQ_ACTIVE_INST_VALU cycles / (GRBM_GUI_ACTIVE x SIMDs), traffic = HBM bytes per launch (TCC_EA0 read/write requests). No dense "
                                      "contraction: MFMA unused. With lanes = 2 avg_launch_ms is the event-bracketed time of a launch that shares the GPU with the "
                                      "other lane's physics launches (that overlap is the point of the lanes); `one_lane` has the same figures for launches that run alone"}, **dom),
            "roofline_path": {"achieved": path, "frac": path / FP64_VALU_PEAK_TFLOPS, "basis": "algorithmic flop convention over the wall time of the timed region",
                              "flop_per_interaction": fpi, "kernel_ms_per_step": kms / a.steps, "kernel_ms_note": "sum of event-bracketed launch times; with lanes = 2 launches overlap and the sum exceeds the step",
                              "by_kernel": {n: {"ms_per_step": by_kernel[j][0] / a.steps, "launches_per_step": by_kernel[j][1] / a.steps}
                                            for j, n in enumerate(names)}},
        }
        if table_sum[0] is not None:
            out["config"]["allreduced_table_rays_accounted_last_step"] = table_sum[0]
        if world == 1 and lanes != 1 and not a.no_one_lane and cam is None and h_sp is None:
            # Not the headline value: the same K steps again with one lane, every launch alone on the GPU - the per-kernel times
            # and the roofline of the dominant kernel without the other lane's launches beside it.
            ctx.set_option("lanes", 1)
            step(a.warmup + a.steps)
            by_kernel[:] = [[0.0, 0], [0.0, 0], [0.0, 0]]; hit_units[0] = 0
            torch.cuda.synchronize(); t1 = time.perf_counter(); inter1 = 0
            for i in range(a.steps):
                inter1 += step(a.warmup + a.steps + 1 + i)[0]
            torch.cuda.synchronize(); dt1 = time.perf_counter() - t1
            ctx.set_option("lanes", lanes)
            h_ms, h_n = by_kernel[1]
            one = {"value": inter1 / dt1, "unit": "interactions/s", "ms_per_step": 1e3 * dt1 / a.steps, "steps": a.steps,
                   "by_kernel": {n: {"ms_per_step": by_kernel[j][0] / a.steps, "launches_per_step": by_kernel[j][1] / a.steps} for j, n in enumerate(names)},
                   "note": "same workload, rt_set_option(lanes, 1); measured after the timed region, not part of `value`"}
            if h_n > 0 and h_ms > 0:
                ach1 = (hit_units[0] * fph / (h_ms / 1e3)) / 1e12
                ct1 = (counter_tflops * dom["avg_launch_ms"] / dom["units_per_launch"] * (hit_units[0] / h_n) / (h_ms / h_n)) if counter_tflops is not None else None
                one["roofline"] = {"kernel": "k_level_hit", "launches": h_n, "avg_launch_ms": h_ms / h_n, "units_per_launch": hit_units[0] / h_n,
                                   "achieved": ct1 if ct1 is not None else ach1, "frac": (ct1 if ct1 is not None else ach1) / FP64_VALU_PEAK_TFLOPS,
                                   "basis": "executed" if ct1 is not None else "algorithmic", "algorithmic_achieved": ach1, "algorithmic_frac": ach1 / FP64_VALU_PEAK_TFLOPS,
                                   "counter_tflops": ct1}
            out["one_lane"] = one
        if not a.no_cpu_baseline and world == 1 and cam is None:
            out["cpu_baseline"] = cpu_baseline(prog, origin, kw, a.seed)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
