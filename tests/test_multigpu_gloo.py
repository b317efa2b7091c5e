"""world_size-2 gloo test of the ray sharding (runs on CPU: the per-rank tracer is the oracle, passed in as trace_fn)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

import cases


def _oracle_fn(sp, rays, sl, args, kw, base):
    from oracle import oracle as O
    r = O.trace(sp, rays, sl, *args, ray_id_base=base, final=True, **kw)   # (kw carries ray_id_stride for an interleaved shard)
    return r.ray_interfaces, r.absorption_table, r.final


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd.multigpu import sharded_trace
    sp, rays = cases.isotropic_rays(3001, 5, origin=(3, 0, 0))
    ris, tab, fin = sharded_trace(sp, rays, cases.sphere_plane(), 10, 1e-9, 1e-6, -1, True, True, False, trace_fn=_oracle_fn, seed=17, gather_final=True)
    # the same run cut the other way: rank r takes rays r, r + 2, ... (ray ids with a stride)
    ris_i, tab_i, fin_i = sharded_trace(sp, rays, cases.sphere_plane(), 10, 1e-9, 1e-6, -1, True, True, False, trace_fn=_oracle_fn, seed=17, gather_final=True,
                                        interleave=True)
    if rank == 1:
        q.put(("final@1", fin.numpy().copy(), fin_i.numpy().copy()))     # the all-gather leaves the whole table on every rank
    if rank == 0:
        q.put((tab, [r.ray_index for r in ris], [r.intersection_point for r in ris], tab_i, [r.ray_index for r in ris_i], [r.intersection_point for r in ris_i]))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_sharding_equals_single_rank():
    from raytracer_amd.multigpu import shard_bounds, shard_ids
    assert [shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_bounds(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert [shard_ids(10, 4, r, True) for r in range(4)] == [(0, 3, 4), (1, 3, 4), (2, 2, 4), (3, 2, 4)]
    assert [shard_ids(2, 4, r, True) for r in range(4)] == [(0, 1, 4), (1, 1, 4), (2, 0, 4), (3, 0, 4)]
    assert [shard_ids(10, 4, r) for r in range(4)] == [(0, 3, 1), (3, 3, 1), (6, 3, 1), (9, 1, 1)]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120), q.get(timeout=120)]
    _, fin1, fin1_i = next(g for g in got if isinstance(g[0], str)); tab, idx, pts, tab_i, idx_i, pts_i = next(g for g in got if not isinstance(g[0], str))
    for p in procs:
        p.join(60); assert p.exitcode == 0
    sp, rays = cases.isotropic_rays(3001, 5, origin=(3, 0, 0))
    ref_ris, ref_tab, ref_fin = _oracle_fn(sp, rays, cases.sphere_plane(), (10, 1e-9, 1e-6, -1, True, True, False), dict(seed=17), 0)
    assert fin1.shape == ref_fin.shape and np.array_equal(fin1, ref_fin, equal_nan=True)   # final rows of all rays, gathered with all_gather_into_tensor
    assert np.allclose(tab, ref_tab, rtol=1e-12, atol=1e-12)          # counter-based RNG: same result for any rank count
    assert len(idx) == len(ref_ris)
    for a, b, c in zip(idx, pts, ref_ris):
        assert np.array_equal(a, c.ray_index) and np.allclose(b, c.intersection_point, rtol=0, atol=1e-12)
    # interleaved shards: the same table, the same final rows in ray order, the same records in ray order (dice mode)
    assert fin1_i.shape == ref_fin.shape and np.array_equal(fin1_i, ref_fin, equal_nan=True)
    assert np.allclose(tab_i, ref_tab, rtol=1e-12, atol=1e-12) and len(idx_i) == len(ref_ris)
    for a, b, c in zip(idx_i, pts_i, ref_ris):
        assert np.array_equal(a, c.ray_index) and np.allclose(b, c.intersection_point, rtol=0, atol=1e-12)
