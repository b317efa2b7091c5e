"""Times rt_match_endpoints (LED-to-camera end point matching) on synthetic end points lying on a cylinder wall of the
jar geometry's reflector (radius 12, height 100): n_cam camera end points in the camera's field of view (a patch),
n_led LED end points all over the wall.  Prints one JSON line.  python tools/bench_match.py [n_cam] [n_led] [max_level]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from raytracer_amd.BubbleImageCreator import match_leds_to_camera  # noqa: E402
from raytracer_amd.RayTracer2 import get_context  # noqa: E402


def wall(n, rng, phi_lo, phi_hi, z_lo, z_hi):
    r = torch.zeros((n, 32), dtype=torch.float64)
    phi = torch.from_numpy(rng.uniform(phi_lo, phi_hi, n)); z = torch.from_numpy(rng.uniform(z_lo, z_hi, n))
    r[:, 0] = 12 * torch.cos(phi); r[:, 1] = 12 * torch.sin(phi); r[:, 2] = z
    d = torch.from_numpy(rng.normal(size=(n, 3))); d /= d.norm(dim=1, keepdim=True)
    nrm = torch.stack([-torch.cos(phi), -torch.sin(phi), torch.zeros(n, dtype=torch.float64)], 1)
    d = torch.where(((d * nrm).sum(1) > 0)[:, None], -d, d)
    r[:, 3:6] = d; r[:, 9] = 1.0; r[:, 13:16] = nrm; r[:, 16] = 29; r[:, 17] = 1; r[:, 19] = 3
    return r


def main():
    n_cam = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    n_led = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    lev = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    rng = np.random.default_rng(0)
    cam = wall(n_cam, rng, -1.0, 1.0, 20, 80).cuda(); led = wall(n_led, rng, -np.pi, np.pi, 0, 100).cuda()
    c = get_context(0); c.set_option("match_max_level", lev)
    best = None
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        pix, lc, ld, la = match_leds_to_camera(cam, led, [29], details=True, ctx=c)
        torch.cuda.synchronize(); wall_s = time.time() - t0
        ms = c.results(False, False, 0)["kernel_ms"]
        best = ms if best is None else min(best, ms)
    matched = int((lc >= 0).sum())
    print(json.dumps({"n_cam": n_cam, "n_led": n_led, "max_level": lev, "kernel_ms": round(best, 3), "wall_s_last": round(wall_s, 3), "matched": matched,
                      "led_rays_per_s": round(n_led / (best * 1e-3), 1), "sum_pix": float(pix.sum()), "sum_add": float(la.sum())}))


if __name__ == "__main__":
    main()
