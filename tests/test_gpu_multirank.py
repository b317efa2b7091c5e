"""The multi-rank path on the HIP kernels: two fresh processes, both on GPU 0, collectives over gloo,
through multigpu.sharded_trace with its default tracer - every rank traces its shard with ray_id_base set to the
shard's start, the absorption table is all-reduced, the final ray rows are all-gathered.  The result must be the
single-process one: tallies to rounding (the order of FP64 atomic additions differs), final rows bit for bit (Philox
streams are keyed by the global ray id, the kernels are the same)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path):
    out = str(tmp_path / "mr.npz")
    env = dict(os.environ, RT_LEVEL_MIN_RAYS="0", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")   # level kernels for the shards as for the whole
    port = 29600 + (os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "mp_gpu_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    sys.path.insert(0, HERE)
    import mp_gpu_worker as W
    from raytracer_amd.RayTracer2 import Context, RayTracer2, trace_point_source, trace_camera
    sl, sp, rays, args, kw = W.scene_and_rays()
    c = Context(0)
    try:
        c.set_option("level_min_rays", 0)
        ri, tab, _, fin = RayTracer2(sp, rays.copy(), sl, *args, ctx=c, return_final=True, **kw)
        pt = trace_point_source((0.3, -0.2, 0.1), 70001, sl, 14, 1e-7, 1e-5, -1, True, ctx=c, **kw)
        csl, cam, cargs, ckw = W.camera_scene()
        ctab, cfin, _ = trace_camera(cam, csl, *cargs, ctx=c, **ckw)
    finally:
        c.close()
    # generated sources, sharded by ray id (contiguous / interleaved): the single-process tallies and final rows
    for name in ("pt0", "pt1"):
        assert z[name].shape == pt.shape and np.allclose(z[name], pt, rtol=1e-9, atol=1e-9), name
    assert abs(pt[:, 0:4].sum() + pt[-1, 4].sum() - 70001) < 1e-6 * 70001
    for t_, f_ in ((z["ctab"], z["cfin"]), (z["ctab2"], z["cfin2"])):
        assert t_.shape == ctab.shape and np.allclose(t_, ctab, rtol=1e-9, atol=1e-9)
        assert f_.shape == cfin.shape and np.array_equal(f_, cfin, equal_nan=True)
    assert z["table"].shape == tab.shape and np.allclose(z["table"], tab, rtol=1e-9, atol=1e-9)
    assert z["final"].shape == fin.shape and np.array_equal(z["final"], fin, equal_nan=True)
    assert abs(tab[:, 0:4].sum() + tab[-1, 4].sum() - sp.shape[0]) < 1e-6 * sp.shape[0]   # conservation, RunSBCGeometry.m:87-92


def test_bench_two_ranks_gloo_on_one_gpu():
    """bench.py's N > 1 branch (rank sharding of the generated source, barrier + max-over-ranks timing, the all-reduce of
    the device table) as the driver launches it, with two ranks that share GPU 0 and gloo for the collectives
    (RT_BENCH_BACKEND=gloo: one GPU cannot host two RCCL ranks).  The line must be well-formed, carry the all-reduced
    table, and count both ranks' interactions."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    port = 29700 + (os.getpid() % 1500)
    common = ["--steps", "2", "--warmup", "1", "--rays", str(1 << 20), "--no-cpu-baseline", "--no-one-lane"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2"] + common
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    two = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, env=env, capture_output=True, text=True, timeout=280, cwd=root)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    one = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["unit"] == "interactions/s" and two["steps"] == 2
    n = 1 << 20
    # every ray ends in exactly one of the tallies (RunSBCGeometry.m:87-92): the all-reduced table of the last step sums to both ranks' rays
    assert abs(two["config"]["allreduced_table_rays_accounted_last_step"] - 2 * n) < 1e-6 * 2 * n
    # weak scaling on a shared GPU: both ranks' interactions are counted (twice the rays, the same interactions per ray) ...
    assert abs(two["config"]["interactions_per_ray"] - one["config"]["interactions_per_ray"]) < 0.01 * one["config"]["interactions_per_ray"]
    # ... and the rate is that of one GPU doing both shares, not a per-rank figure: between half and twice the one-rank rate
    assert 0.5 * one["value"] < two["value"] < 2.0 * one["value"]
