"""finalRays: the per-ray end-point bundle of the reference (python/finalRays.py:2-9) and the two ways to fill it.

from_rows        device reduction: the final rows a trace returns with return_final=True / trace_camera
                 (RT_FINAL_COLS layout, include/rt_mi355x.h) - last interaction of every ray, made in the kernels.
from_interfaces  host reduction over `ray_interfaces` exactly as python/BubbleImageCreator.py:134-165 means it (its
                 numpy lines do not run; the MATLAB they were ported from is what is followed): steps walked from
                 the last to the first, a ray taken at the last step whose ray_index holds its *positive* id.
Both agree whenever no reflected ray is followed (the BubbleImageCreator call: follow_thresh = [1e-5, 100]); rays
that have reflected carry a negative ray_index, which the reference's ismember() never finds, while from_rows
keeps following them.
"""
import numpy as np


class finalRays(object):
    def __init__(self):
        # attributes (python/finalRays.py:5-9)
        self.incoming_ray = None
        self.intersection_point = None
        self.surface_normal = None
        self.ray_index = None
        self.surface_index = None


def _angles(fr):
    fr.cos_incident = -np.sum(fr.incoming_ray[:, 0:3] * fr.surface_normal, axis=1)   # BubbleImageCreator.py:163 (MATLAB 1:3)
    with np.errstate(invalid="ignore"):
        fr.sin_incident = np.sqrt(1 - fr.cos_incident ** 2)                           # :164
        fr.theta_incident = np.arccos(np.clip(fr.cos_incident, -1.0, 1.0))            # :165
    return fr


def from_rows(final_rows):
    rows = np.asarray(final_rows)
    fr = finalRays()
    hit = rows[:, 19] > 0
    fr.incoming_ray = np.where(hit[:, None], rows[:, 3:13], 0.0)
    fr.intersection_point = np.where(hit[:, None], rows[:, 0:3], 0.0)
    fr.surface_normal = np.where(hit[:, None], rows[:, 13:16], 0.0)
    fr.ray_index = np.where(hit, rows[:, 18], 0.0)
    fr.surface_index = np.where(hit, rows[:, 16], 0.0)     # already number * orientation
    fr.found = hit
    fr.n_interactions = rows[:, 19].astype(np.int64)
    return _angles(fr)


def from_interfaces(raytracer_output, n_rays, index_base=0):
    fr = finalRays()
    fr.incoming_ray = np.zeros((n_rays, 10)); fr.intersection_point = np.zeros((n_rays, 3)); fr.surface_normal = np.zeros((n_rays, 3))
    fr.ray_index = np.zeros(n_rays); fr.surface_index = np.zeros(n_rays)
    found = np.zeros(n_rays, bool)
    for n in range(len(raytracer_output) - 1, -1, -1):
        if found.all():
            break
        ri = np.asarray(raytracer_output[n].ray_index, float).ravel()
        vals, first = np.unique(ri, return_index=True)                    # ismember: lowest row of each id
        ok = (vals >= index_base) & (vals < index_base + n_rays) & (vals == np.floor(vals))
        ids = vals[ok].astype(np.int64) - index_base; rows = first[ok]
        new = ~found[ids]
        ids, rows = ids[new], rows[new]
        found[ids] = True
        fr.incoming_ray[ids] = np.asarray(raytracer_output[n].incoming_ray)[rows]
        fr.intersection_point[ids] = np.asarray(raytracer_output[n].intersection_point)[rows]
        fr.surface_normal[ids] = np.asarray(raytracer_output[n].surface_normal)[rows]
        fr.ray_index[ids] = ri[rows]
        fr.surface_index[ids] = np.asarray(raytracer_output[n].surface_index, float).ravel()[rows]
    fr.found = found
    return _angles(fr)
