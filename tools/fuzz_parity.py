"""Randomised parity sweep on the GPU box: random scenes (tests/cases.py coaxial_scene with random surface types,
mixed_geometry), random run modes (dice / ray splitting, records / tallies only, level path on / off / partial, UNIFIED
launch on / off, frames on / off), few steps (no chaotic growth) - device against the oracle.  Prints the cases that
differ.  Scenes stay physical (no ray inside a conductor: there the reference's own arithmetic is NaN-driven and the
two restatements order their NaNs differently).  python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import oracle as O  # noqa: E402
from raytracer_amd.RayTracer2 import Context, RayTracer2  # noqa: E402


def scene(rng):
    if rng.random() < 0.3:
        return cases.mixed_geometry(), (1, 2, 3)
    sl, origin = cases.coaxial_scene(int(rng.integers(0, 1000)))
    for s in sl:  # random optics that keep the scene physical: no ray may find itself inside a conductor
        if s.absorption == 1.0:
            continue
        backed = np.isinf(s.n_outside) or np.isinf(s.n_inside)   # already a conductor behind it
        r = rng.random()
        if r < 0.25:
            s.surface_type = 'unified'; s.unifiedparams = [float(rng.choice([0, 0.05, 0.3])), float(rng.uniform(0, 1)), 0.3, 0.2, 0.1]
        elif r < 0.4 and backed:
            s.surface_type = 'diffuse' if r < 0.33 else 'retro'
        if rng.random() < 0.3 and not np.isinf(s.n_inside):
            s.rayleigh_inside = float(rng.uniform(5, 50)); s.abslength_inside = float(rng.uniform(20, 200))
    return sl, origin


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed0)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1   # run just this case (the random draws of the others are still made)
    bad = 0; t0 = time.time()
    rng_l = np.random.default_rng(seed0 + 1)  # (a stream of its own: the cases of a seed stay what they were before lanes were drawn)
    for ci in range(n_cases):
        sl, origin = scene(rng)
        n = int(rng.choice([300, 5000, 40000]))
        sp, rays = cases.isotropic_rays(n, int(rng.integers(0, 1 << 30)), origin=origin, polarized=bool(rng.random() < 0.5))
        single = bool(rng.random() < 0.6); full = bool(rng.random() < 0.5)
        K = int(rng.choice([3, 5])); tir = float(rng.choice([-1.0, 0.5])); thr = float(rng.choice([1e-9, 1e-3]))  # not 0: children of intensity ~1e-17 (rounding noise) would be followed or not by the last bit
        opts = dict(level_steps=int(rng.choice([0, 2, 1000])), level_min_rays=0, defer_unified=int(rng.random() < 0.5), frames=int(rng.random() < 0.8))
        seed = int(rng.integers(0, 1 << 30))
        lanes = int(rng_l.choice([1, 2, 3]))
        if lanes > 1:   # tally-only runs with device final rows take the lanes; everything else ignores the option
            opts.update(lanes=lanes, lanes_min_rays=0, lanes_alternate=int(rng_l.random() < 0.7))
        args = (K, 1e-7, np.array([thr, thr]), tir, full, single, False)
        if only >= 0 and ci != only:
            continue
        ref = O.trace(sp, rays.copy(), sl, *args, seed=seed, compat="matlab", final=True)
        c = Context(0)
        try:
            for k, v in opts.items():
                c.set_option(k, v)
            laned = lanes > 1 and not full   # (the lanes take final rows only where they stay on the device)
            out = RayTracer2(sp, rays.copy(), sl, *args, seed=seed, compat="matlab", ctx=c, return_final="device" if laned else True)
            if laned:
                out = list(out); out[3] = out[3].cpu().numpy()
            live, scat = c.step_counts(K)
        finally:
            c.close()
        tab = out[1]; msg = []
        if tab.shape != ref.absorption_table.shape:
            msg.append("table shape %s vs %s" % (tab.shape, ref.absorption_table.shape))
        else:
            e = np.abs(tab - ref.absorption_table) / np.maximum(1, np.abs(ref.absorption_table))
            e[np.isnan(e)] = 0
            if e.size and e.max() > 1e-6:
                msg.append("table differs by %.3g" % e.max())
        if not np.array_equal(scat, ref.n_scat):
            msg.append("scatter counts %s vs %s" % (scat.tolist(), ref.n_scat.tolist()))
        f = out[3]; fe = np.abs(f - ref.final) / np.maximum(1, np.abs(ref.final)); fe[np.isnan(f) & np.isnan(ref.final)] = 0
        if np.nanmax(fe) > 1e-4 or not np.array_equal(np.isnan(f), np.isnan(ref.final)):
            rows_bad = np.nonzero(np.nan_to_num(fe, nan=1).max(axis=1) > 1e-4)[0]
            msg.append("final rows differ by %.3g in %d rows (first: row %d, columns %s)" % (np.nanmax(fe), rows_bad.size, rows_bad[0], np.nonzero(np.nan_to_num(fe[rows_bad[0]], nan=1) > 1e-4)[0].tolist()))
            if only >= 0:
                print("device", f[rows_bad[0]].tolist()); print("oracle", ref.final[rows_bad[0]].tolist())
        if full and len(out[0]) != len(ref.ray_interfaces):
            msg.append("record steps %d vs %d" % (len(out[0]), len(ref.ray_interfaces)))
        tag = "case %d: n=%d single=%d full=%d K=%d tir=%g thr=%g %s S=%d" % (ci, n, single, full, K, tir, thr, opts, len(sl))
        if msg:
            bad += 1; print("DIFF", tag, "|", "; ".join(msg), flush=True)
        elif ci % 10 == 0:
            print("ok  ", tag, "(%.0f s)" % (time.time() - t0), flush=True)
    print("done: %d cases, %d differ" % (n_cases, bad))


if __name__ == "__main__":
    main()
