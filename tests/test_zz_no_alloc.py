"""rt_trace_device_noalloc (include/rt_mi355x.h): the entry point that allocates nothing - its queues, hits, lists and
counters live in a static device array - against rt_trace_device on the same rays in HBM, through the C ABI.  Two calls
in a row on one context, the second with a smaller input (the array keeps the first call's contents).  The per-ray
arithmetic is that of the same kernels, so tallies differ only by the order of the FP64 atomic additions (1e-10) and
the step counts not at all."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _soa(sp, rays):
    import torch
    a = np.concatenate([sp.T, rays.T], axis=0)            # [13][n]: start xyz, then the ten ray columns
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _both(c, sp, rays, K, thr, seed):
    import torch
    from raytracer_amd import _lib
    n = sp.shape[0]; S = c.n_surf
    prm = c.params(K, _lib.FLAG_SINGLECHILD, 0, 1e-7, (thr, thr), 0.5, seed, 1000)
    d = _soa(sp, rays)
    torch.cuda.synchronize()
    c._check(c.L.rt_trace_device(c.h, C.c_void_p(d.data_ptr()), n, C.byref(prm)))
    ref = c.results(False, False, n)
    live, scat = c.step_counts(K)
    tab = torch.full((K, 5, S, 2), 7.0, dtype=torch.float64, device="cuda:0")      # garbage in: the call must clear its outputs
    cnt = torch.full((2, K), 9, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    c._check(c.L.rt_trace_device_noalloc(c.h, C.c_void_p(d.data_ptr()), n, C.byref(prm), C.c_void_p(tab.data_ptr()), C.c_void_p(cnt.data_ptr())))
    # what the entry point is for: the call (kernel code loaded by the one above) leaves the device's free memory where it was
    tab2 = torch.full((K, 5, S, 2), 7.0, dtype=torch.float64, device="cuda:0"); cnt2 = torch.full((2, K), 9, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    c._check(c.L.rt_trace_device_noalloc(c.h, C.c_void_p(d.data_ptr()), n, C.byref(prm), C.c_void_p(tab2.data_ptr()), C.c_void_p(cnt2.data_ptr())))
    assert torch.cuda.mem_get_info(0)[0] == free0, "rt_trace_device_noalloc changed the free device memory"
    assert torch.equal(cnt2, cnt) and np.allclose(tab2.cpu().numpy(), tab.cpu().numpy(), rtol=1e-10, atol=1e-10 * n)
    tab = tab.cpu().numpy(); cnt = cnt.cpu().numpy()
    assert np.array_equal(cnt[0], live) and np.array_equal(cnt[1], scat)
    ns = ref["num_steps"]
    assert ns == (np.nonzero(cnt[0])[0].max() + 1 if cnt[0].any() else 0)
    assert np.allclose(tab[:ns], ref["table"], rtol=1e-10, atol=1e-10 * n)
    assert not tab[ns:].any()
    return ns


@pytest.mark.parametrize("scene", ["mixed", "coaxial", "jar"])
def test_noalloc_entry_point_equals_rt_trace_device(scene):
    from raytracer_amd.RayTracer2 import Context
    from raytracer_amd import _lib
    if scene == "mixed":
        sl, origin, K = cases.mixed_geometry(), (1, 2, 3), 25                 # UNIFIED + Rayleigh: three launches per step
    elif scene == "coaxial":
        sl, origin = cases.coaxial_scene(5); K = 12                           # no UNIFIED surface: two launches per step
    else:
        from raytracer_amd import geometries as G
        sl, origin, K = G.create_arbc_geometry(G.sbc_full_mc_specs()), (0, 0, 10), 40
    c = Context(0)
    try:
        c.set_surfaces(sl)
        c.set_option("level_min_rays", 0)                                      # the reference call takes the same kernels
        sp, rays = cases.isotropic_rays(60001, 3, origin=tuple(origin), polarized=True)
        ns1 = _both(c, sp, rays, K, 1e-4, 17)
        ns2 = _both(c, sp[:7003], rays[:7003], K, 1e-4, 18)                    # smaller input into the same scratch array
        assert ns1 >= 3 and ns2 >= 3
        c2 = Context(0)                                                        # a second context on the same GPU shares the scratch array: calls one after the other
        try:
            c2.set_surfaces(sl); c2.set_option("level_min_rays", 0)
            assert _both(c2, sp[:20011], rays[:20011], K, 1e-4, 19) >= 3
            assert _both(c, sp[:5001], rays[:5001], K, 1e-4, 20) >= 3          # and the first one again behind it
        finally:
            c2.close()
        # limits: refused with a clear error, nothing run
        prm = c.params(K, _lib.FLAG_SINGLECHILD, 0, 1e-7, (1e-4, 1e-4), 0.5, 1, 0)
        assert c.L.rt_trace_device_noalloc(c.h, C.c_void_p(8), _lib.NOALLOC_MAX_RAYS + 1, C.byref(prm), C.c_void_p(8), C.c_void_p(8)) != 0
        assert b"at most" in c.L.rt_last_error(c.h)
        prm = c.params(K, _lib.FLAG_SINGLECHILD | _lib.FLAG_FULL_OUTPUT, 0, 1e-7, (1e-4, 1e-4), 0.5, 1, 0)
        assert c.L.rt_trace_device_noalloc(c.h, C.c_void_p(8), 100, C.byref(prm), C.c_void_p(8), C.c_void_p(8)) != 0
        assert b"tally-only" in c.L.rt_last_error(c.h)
    finally:
        c.close()
