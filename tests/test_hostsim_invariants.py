"""The known answers of tests/invariants.py put to the text of raytracer_amd/csrc/rt_device.hpp compiled for the host
(tests/hostsim: frame program, Ferrari quartic, Mueller-form Fresnel - the arithmetic the HIP kernels inline), so that a
deviation shows on a machine without a GPU.  tests/test_gpu_invariants.py asks the kernels themselves."""
import numpy as np
import pytest

import invariants as inv
import hostsim_util as H


def hostsim_trace(sp, rays, sl, K, mtl, thr, tir, full_output, singlechild, raytable, **kw):
    thr = np.atleast_1d(np.asarray(thr, float)); thr = (float(thr[0]), float(thr[-1]))
    return H.trace(sp, rays.copy(), sl, int(K), float(mtl), thr, float(tir), full_output, singlechild, **kw)


@pytest.mark.parametrize("check", inv.ALL_CHECKS, ids=lambda f: f.__name__)
def test_device_header_known_answers(check):
    print(check.__name__, check(hostsim_trace))
