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
    from raytracer_amd.RayTracer2 import Context, RayTracer2
    sl, sp, rays, args, kw = W.scene_and_rays()
    c = Context(0)
    try:
        c.set_option("level_min_rays", 0)
        ri, tab, _, fin = RayTracer2(sp, rays.copy(), sl, *args, ctx=c, return_final=True, **kw)
    finally:
        c.close()
    assert z["table"].shape == tab.shape and np.allclose(z["table"], tab, rtol=1e-9, atol=1e-9)
    assert z["final"].shape == fin.shape and np.array_equal(z["final"], fin, equal_nan=True)
    assert abs(tab[:, 0:4].sum() + tab[-1, 4].sum() - sp.shape[0]) < 1e-6 * sp.shape[0]   # conservation, RunSBCGeometry.m:87-92
