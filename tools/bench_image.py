"""Times the whole of python/BubbleImageCreator.py:125-357 (camera trace, LED trace, end points, LED-to-camera matching)
on the jar geometry: python tools/bench_image.py [n_pixels] [n_led_rays] [on_device 0|1].  One JSON line."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from raytracer_amd import geometries as G  # noqa: E402
from raytracer_amd.BubbleImageCreator import LIGHT_SURFACES, arbc_lights, create_bubble_image  # noqa: E402
from raytracer_amd.RayTracer2 import get_context  # noqa: E402


def main():
    n_pix = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 25
    n_led = int(float(sys.argv[2])) if len(sys.argv) > 2 else 9_000_000
    on_device = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    h = max(1, int(round((n_pix * 0.75) ** 0.5))); w = max(1, n_pix // h)
    specs = G.ArBCSpecs(cam_resolution=np.array([h, w]), lights_nrays=max(1, n_led // 15))
    sl = G.create_arbc_geometry(specs); cam = G.arbc_camera(specs)
    sp, rays = arbc_lights(specs, np.random.default_rng(1))
    lit = LIGHT_SURFACES + ('reflector/diffuser topcone',)   # what this geometry's camera sees (tests/test_gpu_match.py)
    c = get_context(0)
    best = None
    for it in range(3):
        t0 = time.perf_counter()
        out = create_bubble_image(sl, cam, sp, rays.copy(), seed=5, descriptions=lit, ctx=c, on_device=on_device)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    img = out["image"]
    print(json.dumps({"pixels": h * w, "led_rays": sp.shape[0], "on_device": on_device, "wall_s": round(best, 4), "lit_pixels": int(np.count_nonzero(img)),
                      "sum_intensity": float(img.sum()), "camera_rays_per_s": round(h * w / best, 1)}))


if __name__ == "__main__":
    main()
