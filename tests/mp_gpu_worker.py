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


def camera_scene():
    from raytracer_amd import geometries as G
    spec = G.ArBCSpecs(cam_resolution=np.array([60, 44]))
    return G.create_arbc_geometry(spec), G.arbc_camera(spec), (12, 1e-5, [1e-5, 100.0], -1, False, True), dict(seed=31, compat="matlab")


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from raytracer_amd.multigpu import sharded_trace
    sl, sp, rays, args, kw = scene_and_rays()
    ris, tab, fin = sharded_trace(sp, rays, sl, *args, gather_records=False, gather_final=True, **kw)
    # sources generated on the devices, sharded by ray id alone: point source (contiguous and interleaved), camera (interleaved)
    from raytracer_amd.multigpu import sharded_trace_point_source, sharded_trace_camera
    origin = (0.3, -0.2, 0.1)
    pt = [sharded_trace_point_source(origin, 70001, sl, 14, 1e-7, 1e-5, -1, True, interleave=il, **kw) for il in (False, True)]
    csl, cam, cargs, ckw = camera_scene()
    ctab, cfin = sharded_trace_camera(cam, csl, *cargs, **ckw)
    ctab2, cfin2 = sharded_trace_camera(cam, csl, *cargs, interleave=False, **ckw)
    if dist.get_rank() == 0:
        np.savez(sys.argv[1], table=tab, final=fin.cpu().numpy(), pt0=pt[0], pt1=pt[1], ctab=ctab, cfin=cfin.cpu().numpy(), ctab2=ctab2, cfin2=cfin2.cpu().numpy())
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
