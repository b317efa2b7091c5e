"""Worker of tests/test_gpu_multirank.py: one rank of a 2-rank job (python -m torch.distributed.run), every rank on GPU 0,
collectives over gloo (one GPU cannot host two RCCL ranks).  Traces its shard through the HIP path (default tracer of
multigpu.sharded_trace) and, on rank 0, saves the all-reduced table and the all-gathered final rows."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)


def scene_and_rays():
    import cases
    sl, origin = cases.coaxial_scene(7)
    sp, rays = cases.isotropic_rays(50001, 9, origin=tuple(origin), polarized=True)
    return sl, sp, rays, (14, 1e-7, 1e-5, -1, False, True, False), dict(seed=23, compat="matlab")


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from raytracer_amd.multigpu import sharded_trace
    sl, sp, rays, args, kw = scene_and_rays()
    ris, tab, fin = sharded_trace(sp, rays, sl, *args, gather_records=False, gather_final=True, **kw)
    if dist.get_rank() == 0:
        np.savez(sys.argv[1], table=tab, final=fin.cpu().numpy())
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
