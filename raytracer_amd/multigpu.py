"""Ray-sharded tracing over several GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Rays never interact, so the path shards with no data-path collective: rank r traces its share with the global ids of
its rays -- Philox streams are keyed by the global ray id, so the result is the same for any number of ranks and either
way of cutting:
  * contiguous (default): rank r takes the block [r*ceil(N/G), ...), `ray_id_base` = the block's start;
  * interleaved (`interleave=True`): rank r takes rays r, r + G, r + 2G, ... (`ray_id_stride` = G).  For sources that
    are spatially ordered - camera pixels, MATLAB/Examples/ImagingExamples/GenerateRaysFromCamera.m:44-46 - in
    ray-splitting mode, where the number of descendants of a ray depends on where it starts: every rank then gets the
    same mix of long and short lineages.
The only collective is the all-reduce(sum) of the absorption_table (taken from the device as it is:
rt_result_table_device); on request the final ray rows (RT_FINAL_COLS doubles per ray, written by the kernels into a
device tensor) are all-gathered - `all_gather_into_tensor`, over RCCL/xGMI with the nccl backend, no pickling and no
trip through host memory - and the records are gathered to rank 0.  Sources that are generated on the device
(`sharded_trace_point_source`, `sharded_trace_camera`) are sharded by ray id alone: no rank ever holds the rays of
another, or any host array of rays at all (1e10 rays over 8 GPUs).
The reference has no counterpart (it is single-process).  `trace_fn` is the per-rank tracer (default: the HIP path);
tests pass the oracle there to exercise the sharding logic on CPU with gloo.
"""
import numpy as np


def shard_bounds(n, world, rank):
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard_ids(n, world, rank, interleave=False):
    """(first, count, stride): rank's share of the ids 0..n-1 is first, first + stride, ... (count of them)"""
    if interleave:
        return rank, max(0, -(-(n - rank) // world)), world
    lo, hi = shard_bounds(n, world, rank)
    return lo, hi - lo, 1


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _gpu_trace(sp, rays, surfacelist, args, kw, ray_id_base, device, want_final=False):
    from .RayTracer2 import RayTracer2
    out = RayTracer2(sp, rays, surfacelist, *args, ray_id_base=ray_id_base, device=device, return_final="device" if want_final else False,
                     device_table=True, **kw)
    return list(out[0]), out[1], (out[3] if want_final else None)


def allreduce_table(tab, K, S, group=None):
    """sum over ranks of the absorption tables.  `tab`: this rank's table - a torch tensor [K][5][S][2] on the GPU
    (Context.table_device, with its `num_steps`: it goes into the collective as it is with the nccl backend) or a numpy
    array of its num_steps rows (CPU tracers), or None for a rank without rays.  Returns (numpy table [K][5][S][2], the
    reference's num_scatters = the largest over the ranks)."""
    import torch
    import torch.distributed as dist
    world, _ = _world(group)
    nccl = world > 1 and dist.get_backend(group) == "nccl"
    if tab is None:
        t = torch.zeros((K, 5, S, 2), dtype=torch.float64); ns = 0
    elif hasattr(tab, "data_ptr"):
        t = tab; ns = int(getattr(tab, "num_steps", K))
    else:
        t = torch.zeros((K, 5, S, 2), dtype=torch.float64); t[:tab.shape[0]] = torch.from_numpy(np.ascontiguousarray(tab)); ns = int(tab.shape[0])
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
        t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        s = torch.tensor([ns], dtype=torch.int64, device=dev); dist.all_reduce(s, op=dist.ReduceOp.MAX, group=group); ns = int(s[0])
    return t.cpu().numpy(), ns


def gather_final_rows(local, n_total, world, rank, group=None, interleave=False):
    """all-gather of the per-ray final rows: `local` = this rank's [n_local][cols] rows (torch tensor on the GPU, or a numpy
    array on the CPU path; None or empty for a rank without rays), shards as shard_ids() cuts them.  Every rank gets the
    [n_total][cols] table in ray order; with the nccl backend it stays in HBM (RCCL over xGMI), with gloo it goes through
    host memory.  Shards are padded to equal length for the collective, the padding is cut off afterwards."""
    import torch
    import torch.distributed as dist
    per = -(-n_total // world)
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    if local is None:
        local = np.zeros((0, 32))
    t = local if hasattr(local, "data_ptr") else torch.from_numpy(np.ascontiguousarray(local))
    t = t.to(dev)   # (a rank whose shard is empty hands over a CPU array whatever the backend)
    cols = t.shape[1] if t.dim() == 2 and t.shape[0] else 32
    pad = torch.zeros((per, cols), dtype=torch.float64, device=dev)
    if t.shape[0]:
        pad[:t.shape[0]] = t
    out = torch.empty((world * per, cols), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    if interleave:   # row i of rank r is ray r + i * world
        out = out.reshape(world, per, cols).permute(1, 0, 2).reshape(world * per, cols)
    return out[:n_total]


def merge_interfaces(parts, fields, by_ray=False):
    """concatenate per-rank records step by step.  Contiguous shards: rank order = ray order, which is the reference's row
    order in dice mode.  by_ray (interleaved shards, dice mode: at most one row per ray and step): rows sorted by ray id."""
    import types
    steps = max((len(p) for p in parts), default=0)
    out = []
    for k in range(steps):
        ri = types.SimpleNamespace()
        have = [p[k] for p in parts if len(p) > k]
        order = None
        if by_ray and "ray_index" in fields:
            order = np.argsort(np.abs(np.concatenate([np.asarray(h.ray_index).ravel() for h in have])), kind="stable")
        for f in fields:
            v = np.concatenate([np.asarray(getattr(h, f)) for h in have], axis=0)
            setattr(ri, f, v if order is None else v[order])
        out.append(ri)
    return out


def sharded_trace(ray_startingpoints, rays, surfacelist, *args, trace_fn=None, group=None, gather_records=True, gather_final=False, interleave=False, **kw):
    """Same positional arguments as RayTracer2 after the surface list.  Returns (ray_interfaces on rank 0 or this
    rank's own, absorption_table summed over ranks) and, with gather_final=True, as a third item the final rows of all
    rays on every rank (gather_final_rows).  Must be called by every rank of `group`.  interleave: see the module text;
    gathered records are then in ray order in dice mode and grouped by rank in ray-splitting mode."""
    import torch
    import torch.distributed as dist
    world, rank = _world(group)
    sp = np.asarray(ray_startingpoints, dtype=np.float64); ry = np.asarray(rays, dtype=np.float64)
    n = sp.shape[0]
    first, count, stride = shard_ids(n, world, rank, interleave)
    K = int(np.atleast_1d(args[0]).ravel()[0]) if args else 10
    S = len(surfacelist)
    base = kw.pop("ray_id_base", 0)
    if count > 0:
        mine = slice(first, first + (count - 1) * stride + 1, stride)
        if stride != 1:
            kw = dict(kw, ray_id_stride=stride)
        if trace_fn is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
            ris, tab, fin = _gpu_trace(np.ascontiguousarray(sp[mine]), np.ascontiguousarray(ry[mine]), surfacelist, args, kw, base + first, device, gather_final)
        else:
            res = trace_fn(np.ascontiguousarray(sp[mine]), np.ascontiguousarray(ry[mine]), surfacelist, args, kw, base + first)
            ris, tab = res[0], res[1]; fin = res[2] if len(res) > 2 else None
    else:
        ris, tab, fin = [], None, np.zeros((0, 32))
    full, steps = allreduce_table(tab, K, S, group)
    if world > 1 and gather_records:
        parts = [None] * world if rank == 0 else None
        dist.gather_object(ris, parts, dst=0, group=group)
        if rank == 0:
            fields = [f for f in vars(next((r for p in parts for r in p), None) or object()).keys()] if any(len(p) for p in parts) else []
            singlechild = bool(args[5]) if len(args) > 5 else True
            ris = merge_interfaces(parts, fields, by_ray=interleave and singlechild)
    if gather_final:
        if fin is None:
            raise ValueError("gather_final=True needs a tracer that returns final rows")
        final = gather_final_rows(fin, n, world, rank, group, interleave) if world > 1 else fin
        return ris, full[:steps], final
    return ris, full[:steps]


def sharded_trace_point_source(origin, n, surfacelist, *args, group=None, interleave=False, device=None, **kw):
    """n isotropic rays from `origin`, generated on the devices (RayTracer2.trace_point_source): rank r makes and traces
    the rays of its ids only, so a run of any size needs no host array and no rank holds another's rays.  `args`:
    max_scat, min_travel_len, follow_thresh, tir_handling, singlechild as in trace_point_source.  Returns the
    absorption_table summed over ranks (numpy, num_scatters rows).  Must be called by every rank of `group`."""
    import torch
    from .RayTracer2 import trace_point_source
    world, rank = _world(group)
    first, count, stride = shard_ids(int(n), world, rank, interleave)
    K = int(np.atleast_1d(args[0]).ravel()[0]) if args else 10
    base = kw.pop("ray_id_base", 0)
    tab = None
    if count > 0:
        dev = device if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        tab = trace_point_source(origin, count, surfacelist, *args, device=dev, ray_id_base=base + first, ray_id_stride=stride, device_table=True, **kw)
    full, steps = allreduce_table(tab, K, len(surfacelist), group)
    return full[:steps]


def sharded_trace_camera(cam, surfacelist, *args, group=None, interleave=True, gather_final=True, device=None, **kw):
    """One ray per pixel of `cam` (GenerateRaysFromCamera.make_camera), generated on the devices (rt_trace_camera_range):
    rank r takes its share of the pixels - interleaved by default, pixels being spatially ordered - traces them and keeps
    their final rows in HBM.  `args`: max_scat, min_travel_len, follow_thresh, tir_handling, full_output (must be False),
    singlechild as in RayTracer2.trace_camera.  Returns (absorption_table summed over ranks, final rows of all pixels in
    pixel order on every rank - a torch tensor, in HBM with the nccl backend - or None).  Must be called by every rank."""
    import torch
    from .RayTracer2 import trace_camera
    world, rank = _world(group)
    n = int(cam.resolution[0]) * int(cam.resolution[1])
    first, count, stride = shard_ids(n, world, rank, interleave)
    K = int(np.atleast_1d(args[0]).ravel()[0]) if args else 10
    if len(args) > 4 and args[4]:
        raise ValueError("sharded_trace_camera keeps final rows, not records (full_output must be False)")
    base = kw.pop("ray_id_base", 0)
    tab, fin = None, None
    if count > 0:
        dev = device if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        from .RayTracer2 import get_context
        c = kw.pop("ctx", None) or get_context(dev)
        tab_h, fin, _ = trace_camera(cam, surfacelist, *args, ctx=c, ray_id_base=base + first, ray_id_stride=stride, first_pixel=first, n_pixels=count,
                                     device_rows=True, **kw)
        tab = c.table_device(K)
    full, steps = allreduce_table(tab, K, len(surfacelist), group)
    final = None
    if gather_final:
        final = gather_final_rows(fin, n, world, rank, group, interleave) if world > 1 else fin
    return full[:steps], final
