"""Surface objects (python/surface.py:51-62) -> flat FP64 surface table + bounds program of include/rt_mi355x.h.

Also performs the input checks the reference makes when it first touches a surface
(python/RayToShape.py:29 'Geometry has unrecognized shape', RayToPlane.py:15-16,30, RayToCylinder.py:61-62,76,
UnifiedReflectorModel.py:112-114), before any device work.
"""
import numpy as np

from . import bounds as _bounds

ROW = 32
SHAPES = {"plane": 0, "sphere": 1, "cylinder": 2, "quadsurface": 3, "torus": 4}
STYPES = {"normal": 0, "diffuse": 1, "unified": 2, "retro": 3}
BULK_PROPS = ("abslength_outside", "abslength_inside", "rayleigh_outside", "rayleigh_inside")


def _vec3(v, what):
    a = np.asarray(v, dtype=np.float64).ravel()
    if a.size != 3:
        raise Exception('Improper input to %s' % what)
    return a


def _unit(v, msg):
    n2 = float(np.sum(v * v))
    if not n2 > 0:
        raise Exception(msg)
    return v / np.abs(np.sqrt(n2))


def geom_row(shape, pl):
    """13 geometry doubles; axis/normal vectors are normalised here exactly as the RayToX functions do per call."""
    g = np.zeros(13)
    if shape == "plane":
        if len(pl) < 2:
            raise Exception('Improper input to RayToPlane')
        g[0:3] = _vec3(pl[0], "RayToPlane")
        g[3:6] = _unit(_vec3(pl[1], "RayToPlane"), 'Invalid plane for RayToPlane')
        scale = np.max(np.abs(g[0:3]))
    elif shape == "sphere":
        if len(pl) < 2 or np.size(pl[1]) != 1:
            raise Exception('Improper input to RayToSphere')
        g[0:3] = _vec3(pl[0], "RayToSphere"); g[3] = float(pl[1])
        scale = np.max(np.abs(g[0:3])) + abs(g[3])
    elif shape == "cylinder":
        if len(pl) < 3 or float(pl[2]) == 0:
            raise Exception('Improper input to RayToCylinder')
        g[0:3] = _vec3(pl[0], "RayToCylinder")
        g[3:6] = _unit(_vec3(pl[1], "RayToCylinder"), 'Invalid cylinder for RayToCylinder')
        g[6] = float(pl[2])
        scale = np.max(np.abs(g[0:3])) + abs(g[6])
    elif shape == "quadsurface":
        if len(pl) < 3 or np.size(pl[0]) != 9 or np.size(pl[1]) != 3 or np.size(pl[2]) != 1:
            raise Exception('Improper input to RayToQuadsurface')
        g[0:9] = np.asarray(pl[0], dtype=np.float64).reshape(9)
        g[9:12] = _vec3(pl[1], "RayToQuadsurface"); g[12] = float(np.asarray(pl[2]).ravel()[0])
        scale = 10.0
    elif shape == "torus":
        if len(pl) < 4:
            raise Exception('Improper input to RayToTorus')
        g[0:3] = _vec3(pl[0], "RayToTorus")
        g[3:6] = _unit(_vec3(pl[1], "RayToTorus"), 'Invalid torus for RayToTorus')
        g[6] = float(pl[2]); g[7] = float(pl[3])
        scale = np.max(np.abs(g[0:3])) + abs(g[6]) + abs(g[7])
    else:
        raise Exception('Geometry has unrecognized shape')
    return g, max(float(scale), 1.0)


def compile_surfaces(surfacelist, nobulk=False, check_bounds=True):
    """-> (table[S,32], atoms[A,12], code[C] uint32).  `nobulk` reproduces SURVEY quirk Q3 of the numpy port
    (python/RayTracer2.py:238-239 overwrites every bulk property with inf)."""
    S = len(surfacelist)
    table = np.zeros((S, ROW))
    atoms, code = [], []
    cache = {}
    for i, s in enumerate(surfacelist):
        shape = getattr(s, "shape", None)
        if shape not in SHAPES:
            raise Exception('Geometry has unrecognized shape')
        g, scale = geom_row(shape, s.param_list)
        table[i, 0] = SHAPES[shape]
        table[i, 1] = STYPES.get(getattr(s, "surface_type", None), 0)  # anything else behaves as 'normal'
        table[i, 2:15] = g
        table[i, 15] = float(s.n_outside); table[i, 16] = float(s.n_inside); table[i, 17] = float(s.absorption)
        for j, name in enumerate(BULK_PROPS):  # RayTracer2.m:263-269
            v = getattr(s, name, None)
            table[i, 18 + j] = np.inf if (nobulk or v is None or np.size(v) == 0) else float(np.asarray(v).ravel()[0])
        up = getattr(s, "unifiedparams", None)
        up = np.array([0, 1, 0, 1, 0], float) if (up is None or np.size(up) == 0) else np.asarray(up, dtype=np.float64).ravel()
        if up.size != 5:
            raise Exception('improper input to UnifiedReflectorModel')
        if table[i, 1] == STYPES["unified"] and np.any(np.cumsum(up[2:5]) > 1):
            raise Exception('impropper reflection parameters to UnifiedReflectorModel')
        table[i, 22:27] = up
        fn = getattr(s, "inbounds_function", None)
        key = id(fn)
        if key in cache:
            off, ln = cache[key]
        else:
            a, c = _bounds.compile_bounds(fn, scale=3 * scale, check=check_bounds)
            off = len(code); ln = len(c)
            base = len(atoms)
            for ins in c:
                op, arg = ins >> 24, ins & 0xFFFFFF
                code.append((op << 24) | (arg + base if op == _bounds.OP_ATOM else 0))
            atoms.extend(a)
            cache[key] = (off, ln)
        table[i, 27] = off; table[i, 28] = ln
    atoms_arr = np.zeros((max(len(atoms), 1), _bounds.ATOM))
    for k, a in enumerate(atoms):
        atoms_arr[k] = a
    code_arr = np.array(code if code else [_bounds.OP_TRUE << 24], dtype=np.uint32)
    return table, atoms_arr, code_arr


def rayleigh_lut(precision=100000):
    """rayleigh_azimuth of MATLAB/Physics/RayleighScattering.m:55-60 (interp1 == np.interp)."""
    cumd = np.linspace(0, 1, precision + 1)
    phid = np.linspace(0, 2 * np.pi, precision * 10)
    cumd_phi = (phid - .25 * np.sin(2 * phid)) / (2 * np.pi)
    return np.interp(cumd, cumd_phi, phid)
