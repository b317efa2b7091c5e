"""`surface` attribute bag with the fields of the reference (python/surface.py:51-62).

Optional attributes are set ad hoc by callers exactly as with the reference: `unifiedparams`
(python/TestGeometry.py:52), `abslength_outside/inside`, `rayleigh_outside/inside` (python/RayTracer2.py:232-248).
"""


class surface(object):
    def __init__(self):
        self.description = None
        self.n_outside = None
        self.n_inside = None
        self.surface_type = None
        self.absorption = None
        self.shape = None
        self.param_list = None
        self.inbounds_function = None
