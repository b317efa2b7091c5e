"""Ray-sharded tracing over several GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Rays never interact, so the path shards with no data-path collective: rank r traces the contiguous block
[r*ceil(N/G), ...) with ray_id_base set to the block start -- Philox streams are keyed by the global ray id, so the
result is the same for any number of ranks.  The only collective is the all-reduce(sum) of the absorption_table and
the per-step counters; on request the final ray rows (RT_FINAL_COLS doubles per ray, written by the kernels into a
device tensor) are all-gathered - `all_gather_into_tensor`, over RCCL/xGMI with the nccl backend, no pickling and no
trip through host memory - and the records are gathered to rank 0.  The reference has no counterpart (it is
single-process).  `trace_fn` is the per-rank tracer (default: the HIP path); tests pass the oracle there to
exercise the sharding logic on CPU with gloo.
"""
import numpy as np


def shard_bounds(n, world, rank):
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _gpu_trace(sp, rays, surfacelist, args, kw, ray_id_base, device, want_final=False):
    from .RayTracer2 import RayTracer2
    out = RayTracer2(sp, rays, surfacelist, *args, ray_id_base=ray_id_base, device=device, return_final="device" if want_final else False, **kw)
    return list(out[0]), out[1], (out[3] if want_final else None)


def gather_final_rows(local, n_total, world, rank, group=None):
    """all-gather of the per-ray final rows: `local` = this rank's [n_local][cols] rows (torch tensor on the GPU, or a numpy
    array on the CPU path), shards as shard_bounds() cuts them.  Every rank gets the [n_total][cols] table; with the nccl
    backend it stays in HBM (RCCL over xGMI), with gloo it goes through host memory.  Shards are padded to equal length
    for the collective, the padding is cut off afterwards."""
    import torch
    import torch.distributed as dist
    per = -(-n_total // world)
    nccl = dist.get_backend(group) == "nccl"
    t = local if hasattr(local, "data_ptr") else torch.from_numpy(np.ascontiguousarray(local))
    t = t if nccl else t.cpu()
    cols = t.shape[1] if t.dim() == 2 else 0
    pad = torch.zeros((per, cols), dtype=torch.float64, device=t.device)
    pad[:t.shape[0]] = t
    out = torch.empty((world * per, cols), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:n_total]


def merge_interfaces(parts, fields):
    """concatenate per-rank records step by step (rank order = ray order, which is the reference's row order in dice mode)"""
    import types
    steps = max((len(p) for p in parts), default=0)
    out = []
    for k in range(steps):
        ri = types.SimpleNamespace()
        have = [p[k] for p in parts if len(p) > k]
        for f in fields:
            setattr(ri, f, np.concatenate([np.asarray(getattr(h, f)) for h in have], axis=0))
        out.append(ri)
    return out


def sharded_trace(ray_startingpoints, rays, surfacelist, *args, trace_fn=None, group=None, gather_records=True, gather_final=False, **kw):
    """Same positional arguments as RayTracer2 after the surface list.  Returns (ray_interfaces on rank 0 or this
    rank's own, absorption_table summed over ranks) and, with gather_final=True, as a third item the final rows of all
    rays on every rank (gather_final_rows).  Must be called by every rank of `group`."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sp = np.asarray(ray_startingpoints, dtype=np.float64); ry = np.asarray(rays, dtype=np.float64)
    n = sp.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    K = int(np.atleast_1d(args[0]).ravel()[0]) if args else 10
    S = len(surfacelist)
    if hi > lo:
        if trace_fn is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
            ris, tab, fin = _gpu_trace(sp[lo:hi], ry[lo:hi].copy(), surfacelist, args, kw, lo + kw.pop("ray_id_base", 0), device, gather_final)
        else:
            res = trace_fn(sp[lo:hi], ry[lo:hi].copy(), surfacelist, args, kw, lo)
            ris, tab = res[0], res[1]; fin = res[2] if len(res) > 2 else None
    else:
        ris, tab, fin = [], np.zeros((0, 5, S, 2)), np.zeros((0, 32))
    full = np.zeros((K, 5, S, 2)); full[:tab.shape[0]] = tab
    steps = np.array([tab.shape[0]], dtype=np.int64)
    if world > 1:
        dev = "cuda" if (dist.get_backend(group) == "nccl") else "cpu"
        t = torch.from_numpy(full).to(dev); dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group); full = t.cpu().numpy()
        s = torch.from_numpy(steps).to(dev); dist.all_reduce(s, op=dist.ReduceOp.MAX, group=group); steps = s.cpu().numpy()
        if gather_records:
            parts = [None] * world if rank == 0 else None
            dist.gather_object(ris, parts, dst=0, group=group)
            if rank == 0:
                from .RayTracer2 import rayInterfaces  # noqa: F401
                fields = [f for f in vars(next((r for p in parts for r in p), None) or object()).keys()] if any(len(p) for p in parts) else []
                ris = merge_interfaces(parts, fields)
    if gather_final:
        if fin is None:
            raise ValueError("gather_final=True needs a tracer that returns final rows")
        final = gather_final_rows(fin, n, world, rank, group) if world > 1 else fin
        return ris, full[:int(steps[0])], final
    return ris, full[:int(steps[0])]
