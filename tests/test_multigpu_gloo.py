"""world_size-2 gloo test of the ray sharding (runs on CPU: the per-rank tracer is the oracle, passed in as trace_fn)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

import cases


def _oracle_fn(sp, rays, sl, args, kw, base):
    from oracle import oracle as O
    r = O.trace(sp, rays, sl, *args, ray_id_base=base, **kw)
    return r.ray_interfaces, r.absorption_table


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd.multigpu import sharded_trace
    sp, rays = cases.isotropic_rays(3001, 5, origin=(3, 0, 0))
    ris, tab = sharded_trace(sp, rays, cases.sphere_plane(), 10, 1e-9, 1e-6, -1, True, True, False, trace_fn=_oracle_fn, seed=17)
    if rank == 0:
        q.put((tab, [r.ray_index for r in ris], [r.intersection_point for r in ris]))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_sharding_equals_single_rank():
    from raytracer_amd.multigpu import shard_bounds
    assert [shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_bounds(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tab, idx, pts = q.get(timeout=120)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    sp, rays = cases.isotropic_rays(3001, 5, origin=(3, 0, 0))
    ref_ris, ref_tab = _oracle_fn(sp, rays, cases.sphere_plane(), (10, 1e-9, 1e-6, -1, True, True, False), dict(seed=17), 0)
    assert np.allclose(tab, ref_tab, rtol=1e-12, atol=1e-12)          # counter-based RNG: same result for any rank count
    assert len(idx) == len(ref_ris)
    for a, b, c in zip(idx, pts, ref_ris):
        assert np.array_equal(a, c.ray_index) and np.allclose(b, c.intersection_point, rtol=0, atol=1e-12)
