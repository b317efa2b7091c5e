"""The known answers of tests/invariants.py (numpy.roots, closed forms, Beer-Lambert, Lambert, the Rayleigh dipole law,
Brewster / TIR, the UNIFIED facet-angle law) put to the HIP path through the C ABI - by the level-synchronous kernels,
by the persistent kernel, and with the frame program switched off.  Mirrors tests/test_oracle_invariants.py."""
import types

import pytest

import invariants as inv

pytestmark = pytest.mark.gpu

PATHS = {"default": {}, "level": dict(level_steps=1000, level_min_rays=0), "persistent": dict(level_steps=0),
         "level-noframes": dict(level_steps=1000, level_min_rays=0, frames=0)}


def gpu_tracer(options):
    def trace(sp, rays, sl, *args, **kw):
        from raytracer_amd.RayTracer2 import Context, RayTracer2
        c = Context(0)
        try:
            for key, v in options.items():
                c.set_option(key, v)
            out = RayTracer2(sp, rays.copy(), sl, *args, ctx=c, **kw)
        finally:
            c.close()
        return types.SimpleNamespace(ray_interfaces=list(out[0]), absorption_table=out[1], raytable=out[2])
    return trace


@pytest.mark.parametrize("path", sorted(PATHS))
@pytest.mark.parametrize("check", inv.ALL_CHECKS, ids=lambda f: f.__name__)
def test_gpu_known_answers(check, path):
    print(check.__name__, path, check(gpu_tracer(PATHS[path])))


@pytest.mark.parametrize("path", ["default", "persistent"])
def test_gpu_raytable_rule(path):
    print(inv.check_raytable_rule(gpu_tracer(PATHS[path])))
