"""Compile a reference-style `inbounds_function` (python/surface.py:19-32: callable p[N,3,M] -> bool[N,M]) into the
bounds program of include/rt_mi355x.h, which the HIP kernel evaluates per candidate hit point.

The callable is traced once with a symbolic stand-in for `p`.  Supported inside the callable (this covers every
predicate of python/TestGeometry.py, SphereTest.py, UnifiedTest.py, SBCGeometry, createGeometry.py and of the MATLAB
geometry files CreateArBCGeometry.m / CreateSBCGeometry.m when written in numpy):
  p[:, k, :], p[:, k]          coordinate k
  p +- v, p * v, (..)**2       whole-point arithmetic with 3-vectors (broadcast along axis 1), np.sum(.., axis=1)
  + - * / ** of expressions whose result stays a polynomial of degree <= 2 in (x, y, z)
  np.sqrt(e) <cmp> c, np.abs(e) <cmp> c, np.arctan2(lin, lin) <cmp> c
  < <= > >= == != ; boolean * & | ^ ~ np.logical_and/or/not/xor ; np.reshape / np.squeeze / .reshape (no-ops)
After compilation the program is verified against the callable on random points and refused on any mismatch.
"""
import numpy as np

OP_ATOM, OP_AND, OP_OR, OP_NOT, OP_TRUE, OP_FALSE, OP_XOR = 1, 2, 3, 4, 5, 6, 7
K_QUAD, K_LIN, K_ATAN2, K_COORD = 0, 1, 2, 3
CMP = {"<": 0, "<=": 1, ">": 2, ">=": 3, "==": 4, "!=": 5}
_FLIP = {"<": ">", "<=": ">=", ">": "<", ">=": "<=", "==": "==", "!=": "!="}
ATOM = 12


class BoundsCompileError(Exception):
    pass


def _is_num(x):
    return isinstance(x, (int, float, np.integer, np.floating)) or (isinstance(x, np.ndarray) and x.size == 1)


class _Base(object):
    __array_priority__ = 1000

    # numpy functions applied to symbolic objects
    def __array_function__(self, func, types, args, kwargs):
        name = func.__name__
        if name in ("reshape", "squeeze", "asarray", "array", "ascontiguousarray", "atleast_2d", "transpose"):
            return args[0]
        if name == "sum":
            return args[0]._sum(*args[1:], **kwargs)
        if name == "all" and isinstance(args[0], Bool):
            return args[0]
        if name == "size":   # `np.size(p, 0)` as the second argument of reshape (python/createGeometry.py:1105 and every surface of that file)
            axis = args[1] if len(args) > 1 else kwargs.get("axis")
            return 3 if (isinstance(args[0], Vec) and axis == 1) else 1
        if name == "shape":
            return args[0].shape
        raise BoundsCompileError("numpy function %s is not supported inside inbounds_function" % name)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":
            raise BoundsCompileError("ufunc method %s not supported" % method)
        n = ufunc.__name__
        a = inputs[0]; b = inputs[1] if len(inputs) > 1 else None
        table = {"add": lambda: a + b, "subtract": lambda: a - b, "multiply": lambda: a * b, "true_divide": lambda: a / b,
                 "divide": lambda: a / b, "negative": lambda: -a, "power": lambda: a ** b, "square": lambda: a * a,
                 "less": lambda: a < b, "less_equal": lambda: a <= b, "greater": lambda: a > b, "greater_equal": lambda: a >= b,
                 "equal": lambda: a == b, "not_equal": lambda: a != b,
                 "logical_and": lambda: _as_bool(a) & _as_bool(b), "logical_or": lambda: _as_bool(a) | _as_bool(b),
                 "logical_xor": lambda: _as_bool(a) ^ _as_bool(b), "logical_not": lambda: ~_as_bool(a),
                 "bitwise_and": lambda: _as_bool(a) & _as_bool(b), "bitwise_or": lambda: _as_bool(a) | _as_bool(b),
                 "bitwise_xor": lambda: _as_bool(a) ^ _as_bool(b), "invert": lambda: ~_as_bool(a),
                 "sqrt": lambda: Func("sqrt", _as_poly(a)), "absolute": lambda: Func("abs", _as_poly(a)),
                 "fabs": lambda: Func("abs", _as_poly(a)), "arctan2": lambda: Func("atan2", _as_poly(a), _as_poly(b))}
        if n not in table:
            raise BoundsCompileError("ufunc %s is not supported inside inbounds_function" % n)
        return table[n]()

    def reshape(self, *a, **k):
        return self

    def flatten(self):
        return self

    def astype(self, *a, **k):
        return self

    @property
    def shape(self):
        return (1, 1)


def _as_poly(x):
    if isinstance(x, Poly):
        return x
    if _is_num(x):
        return Poly({(0, 0, 0): float(np.asarray(x).ravel()[0])})
    raise BoundsCompileError("expected a polynomial expression, got %r" % (type(x),))


def _as_bool(x):
    if isinstance(x, Bool):
        return x
    if isinstance(x, (bool, np.bool_)):
        return Bool(("const", bool(x)))
    if isinstance(x, np.ndarray) and x.dtype == bool and x.size == 1:
        return Bool(("const", bool(x.ravel()[0])))
    raise BoundsCompileError("expected a boolean expression, got %r" % (type(x),))


class Poly(_Base):
    """polynomial in x, y, z; terms {(i,j,k): coef}"""

    def __init__(self, terms):
        self.t = {k: float(v) for k, v in terms.items() if v != 0 or k == (0, 0, 0)}

    def degree(self):
        return max([sum(k) for k, v in self.t.items() if v != 0] + [0])

    def const(self):
        return self.t.get((0, 0, 0), 0.0)

    def __add__(self, o):
        o = _as_poly(o); t = dict(self.t)
        for k, v in o.t.items():
            t[k] = t.get(k, 0.0) + v
        return Poly(t)
    __radd__ = __add__

    def __neg__(self):
        return Poly({k: -v for k, v in self.t.items()})

    def __sub__(self, o):
        return self + (-_as_poly(o))

    def __rsub__(self, o):
        return _as_poly(o) + (-self)

    def __mul__(self, o):
        if isinstance(o, Bool):
            raise BoundsCompileError("polynomial * boolean")
        o = _as_poly(o); t = {}
        for k1, v1 in self.t.items():
            for k2, v2 in o.t.items():
                k = (k1[0] + k2[0], k1[1] + k2[1], k1[2] + k2[2])
                t[k] = t.get(k, 0.0) + v1 * v2
        r = Poly(t)
        if r.degree() > 2:
            raise BoundsCompileError("inbounds_function needs a polynomial of degree > 2 in p")
        return r
    __rmul__ = __mul__

    def __truediv__(self, o):
        if not _is_num(o):
            raise BoundsCompileError("division by a p-dependent expression")
        return self * (1.0 / float(np.asarray(o).ravel()[0]))

    def __pow__(self, e):
        e = float(np.asarray(e).ravel()[0])
        if e == 2:
            return self * self
        if e == 1:
            return self
        if e == 0.5:
            return Func("sqrt", self)
        raise BoundsCompileError("power %r not supported" % e)

    def _cmp(self, o, op):
        if isinstance(o, Func):
            return o._cmp(self, _FLIP[op])
        d = self - _as_poly(o)
        return Bool(("poly", d, op))

    def __lt__(self, o): return self._cmp(o, "<")
    def __le__(self, o): return self._cmp(o, "<=")
    def __gt__(self, o): return self._cmp(o, ">")
    def __ge__(self, o): return self._cmp(o, ">=")
    def __eq__(self, o): return self._cmp(o, "==")
    def __ne__(self, o): return self._cmp(o, "!=")
    __hash__ = None

    def _sum(self, *a, **k):
        return self


class Func(_Base):
    """sqrt(poly), abs(poly) or atan2(lin, lin): only comparable with a constant"""

    def __init__(self, kind, a, b=None):
        self.kind, self.a, self.b = kind, a, b

    def _cmp(self, o, op):
        if not _is_num(o) and not (isinstance(o, Poly) and o.degree() == 0):
            raise BoundsCompileError("%s(..) can only be compared with a constant" % self.kind)
        c = o.const() if isinstance(o, Poly) else float(np.asarray(o).ravel()[0])
        if self.kind == "atan2":
            return Bool(("atan2", self.a, self.b, op, c))
        if self.kind == "sqrt":  # sqrt(e) cmp c  <=>  e cmp c^2 for c >= 0 (e >= 0 assumed, NaN otherwise)
            if c < 0:
                return Bool(("const", op in (">", ">=", "!=")))
            return Bool(("poly", self.a - c * c, op))
        # abs(e) cmp c
        if op in ("<", "<="):
            return Bool(("poly", self.a - c, op)) & Bool(("poly", self.a + c, _FLIP[op]))
        if op in (">", ">="):
            return Bool(("poly", self.a - c, op)) | Bool(("poly", self.a + c, _FLIP[op]))
        if op == "==":
            return Bool(("poly", self.a - c, "==")) | Bool(("poly", self.a + c, "=="))
        return Bool(("poly", self.a - c, "!=")) & Bool(("poly", self.a + c, "!="))

    def __lt__(self, o): return self._cmp(o, "<")
    def __le__(self, o): return self._cmp(o, "<=")
    def __gt__(self, o): return self._cmp(o, ">")
    def __ge__(self, o): return self._cmp(o, ">=")
    def __eq__(self, o): return self._cmp(o, "==")
    def __ne__(self, o): return self._cmp(o, "!=")
    __hash__ = None

    def __pow__(self, e):
        if self.kind == "sqrt" and float(e) == 2:
            return self.a
        if self.kind == "abs" and float(e) == 2:
            return self.a * self.a
        raise BoundsCompileError("power of %s" % self.kind)


class Bool(_Base):
    def __init__(self, node):
        self.node = node

    def __and__(self, o): return Bool(("and", self.node, _as_bool(o).node))
    def __or__(self, o): return Bool(("or", self.node, _as_bool(o).node))
    def __xor__(self, o): return Bool(("xor", self.node, _as_bool(o).node))
    def __invert__(self): return Bool(("not", self.node))
    __rand__ = __and__; __ror__ = __or__; __rxor__ = __xor__
    __mul__ = __and__; __rmul__ = __and__   # the reference writes AND as `*` (python/TestGeometry.py:48)
    __add__ = __or__; __radd__ = __or__

    def __eq__(self, o):
        if isinstance(o, (bool, np.bool_, int)):
            return self if bool(o) else ~self
        return ~(self ^ _as_bool(o))
    __hash__ = None

    def __bool__(self):
        raise BoundsCompileError("inbounds_function uses Python `and`/`or`/`not`/`if` on arrays; use & | ~")

    def _sum(self, *a, **k):
        raise BoundsCompileError("sum of booleans")


class Vec(_Base):
    """the whole point array p (or p combined with 3-vectors): three polynomials along axis 1"""

    def __init__(self, c):
        self.c = list(c)

    @property
    def shape(self):
        return (1, 3, 1)

    @staticmethod
    def _vec3(o):
        if isinstance(o, Vec):
            return o.c
        if _is_num(o):
            v = float(np.asarray(o).ravel()[0]); return [v, v, v]
        a = np.asarray(o, float)
        if a.size == 3:
            return [float(x) for x in a.ravel()]
        # repmat-style arrays of shape (N,3) / (N,3,M) with identical rows
        if a.ndim >= 2 and a.shape[1] == 3:
            a2 = np.moveaxis(a, 1, 0).reshape(3, -1)
            if np.all(a2 == a2[:, :1]):
                return [float(x) for x in a2[:, 0]]
        raise BoundsCompileError("cannot broadcast %r against p" % (a.shape,))

    def _bin(self, o, f):
        v = self._vec3(o)
        return Vec([f(_as_poly(a), b) for a, b in zip(self.c, v)])

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    __radd__ = __add__
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    __rmul__ = __mul__
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __neg__(self): return Vec([-a for a in self.c])
    def __pow__(self, e): return Vec([a ** e for a in self.c])

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) >= 2 and isinstance(key[1], (int, np.integer)):
            return self.c[int(key[1])]
        if len(key) >= 2 and isinstance(key[1], slice):
            idx = range(3)[key[1]]
            if len(idx) == 1:
                return self.c[idx[0]]
            if len(idx) == 3:
                return self
        if len(key) == 1 or (len(key) >= 2 and key[1] == slice(None)):
            return self
        raise BoundsCompileError("unsupported index %r into p" % (key,))

    def _sum(self, *a, **k):
        axis = k.get("axis", a[0] if a else None)
        if axis not in (1, -2, None):
            raise BoundsCompileError("np.sum over axis %r of p" % (axis,))
        return self.c[0] + self.c[1] + self.c[2]

    def sum(self, *a, **k):
        return self._sum(*a, **k)


# ------------------------------------------------------------------------------------------------ code generation
def _poly_atom(d, op):
    """atom for  d(x,y,z) <op> 0, constant moved to the right-hand side"""
    rhs = -d.const()
    t = {k: v for k, v in d.t.items() if k != (0, 0, 0) and v != 0}
    a = np.zeros(ATOM)
    a[1] = CMP[op]; a[2] = rhs
    if not t:  # constant predicate
        v = 0.0
        return ("const", {"<": v < rhs, "<=": v <= rhs, ">": v > rhs, ">=": v >= rhs, "==": v == rhs, "!=": v != rhs}[op])
    lin = {(1, 0, 0): 9, (0, 1, 0): 10, (0, 0, 1): 11}
    quad = {(2, 0, 0): 3, (0, 2, 0): 4, (0, 0, 2): 5, (1, 1, 0): 6, (1, 0, 1): 7, (0, 1, 1): 8}
    if len(t) == 1 and list(t.keys())[0] in lin and list(t.values())[0] == 1.0:
        a[0] = K_COORD; a[3] = lin[list(t.keys())[0]] - 9
        return ("atom", a)
    if all(k in lin for k in t):
        a[0] = K_LIN
    else:
        a[0] = K_QUAD
    for k, v in t.items():
        a[lin[k] if k in lin else quad[k]] = v
    return ("atom", a)


def _lin4(p):
    if p.degree() > 1:
        raise BoundsCompileError("arctan2 arguments must be linear in p")
    return [p.t.get((1, 0, 0), 0.0), p.t.get((0, 1, 0), 0.0), p.t.get((0, 0, 1), 0.0), p.const()]


def _emit(node, atoms, code):
    kind = node[0]
    if kind == "const":
        code.append((OP_TRUE if node[1] else OP_FALSE) << 24)
    elif kind == "poly":
        r = _poly_atom(node[1], node[2])
        if r[0] == "const":
            code.append((OP_TRUE if r[1] else OP_FALSE) << 24)
        else:
            atoms.append(r[1]); code.append((OP_ATOM << 24) | (len(atoms) - 1))
    elif kind == "atan2":
        a = np.zeros(ATOM); a[0] = K_ATAN2; a[1] = CMP[node[3]]; a[2] = node[4]
        a[3:7] = _lin4(node[1]); a[7:11] = _lin4(node[2])
        atoms.append(a); code.append((OP_ATOM << 24) | (len(atoms) - 1))
    elif kind in ("and", "or", "xor"):
        _emit(node[1], atoms, code); _emit(node[2], atoms, code)
        code.append({"and": OP_AND, "or": OP_OR, "xor": OP_XOR}[kind] << 24)
    elif kind == "not":
        _emit(node[1], atoms, code); code.append(OP_NOT << 24)
    else:
        raise BoundsCompileError("bad node %r" % (kind,))


def _depth(code):
    d = m = 0
    for ins in code:
        op = ins >> 24
        if op in (OP_ATOM, OP_TRUE, OP_FALSE):
            d += 1
        elif op in (OP_AND, OP_OR, OP_XOR):
            d -= 1
        m = max(m, d)
    return m


def trace_bounds(fn):
    """-> (list of atom rows, list of uint32 instructions) for one inbounds_function"""
    if fn is None:
        return [], [OP_TRUE << 24]
    p = Vec([Poly({(1, 0, 0): 1.0}), Poly({(0, 1, 0): 1.0}), Poly({(0, 0, 1): 1.0})])
    try:
        r = fn(p)
    except BoundsCompileError:
        raise
    except Exception as e:
        raise BoundsCompileError("could not trace inbounds_function: %s: %s" % (type(e).__name__, e))
    if isinstance(r, (bool, np.bool_)):
        r = Bool(("const", bool(r)))
    if not isinstance(r, Bool):
        raise BoundsCompileError("inbounds_function did not return a boolean expression of p (got %r)" % (type(r),))
    atoms, code = [], []
    _emit(r.node, atoms, code)
    if _depth(code) > 32:
        raise BoundsCompileError("bounds program deeper than 32")
    return atoms, code


def eval_program(atoms, code, pts):
    """numpy evaluator of a bounds program (host-side verification only)"""
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    st = []
    with np.errstate(all="ignore"):
        for ins in code:
            op, arg = ins >> 24, ins & 0xFFFFFF
            if op == OP_ATOM:
                a = atoms[arg]; k = int(a[0])
                if k == K_COORD:
                    v = pts[:, int(a[3])]
                elif k == K_LIN:
                    v = a[9] * x + a[10] * y + a[11] * z
                elif k == K_QUAD:
                    v = a[3] * x * x + a[4] * y * y + a[5] * z * z + a[6] * x * y + a[7] * x * z + a[8] * y * z + a[9] * x + a[10] * y + a[11] * z
                else:
                    v = np.arctan2(a[3] * x + a[4] * y + a[5] * z + a[6], a[7] * x + a[8] * y + a[9] * z + a[10])
                c = int(a[1]); rhs = a[2]
                st.append([v < rhs, v <= rhs, v > rhs, v >= rhs, v == rhs, v != rhs][c])
            elif op == OP_AND:
                b = st.pop(); st[-1] = st[-1] & b
            elif op == OP_OR:
                b = st.pop(); st[-1] = st[-1] | b
            elif op == OP_XOR:
                b = st.pop(); st[-1] = st[-1] ^ b
            elif op == OP_NOT:
                st[-1] = ~st[-1]
            elif op == OP_TRUE:
                st.append(np.ones(len(pts), bool))
            elif op == OP_FALSE:
                st.append(np.zeros(len(pts), bool))
    return st[-1] if st else np.ones(len(pts), bool)


def verify(fn, atoms, code, scale=10.0, n=4096, seed=1234):
    """Refuse a program that disagrees with the callable on random points (both the N x 3 x 1 and N x 3 x 2 layouts)."""
    if fn is None:
        return
    rng = np.random.default_rng(seed)
    pts = np.concatenate([rng.uniform(-s, s, (n, 3)) for s in (0.1 * scale, scale, 3 * scale)])
    # margin: skip points that sit within rounding of a threshold of any atom
    got = eval_program(atoms, code, pts)
    for M in (1, 2):
        arr = np.repeat(pts[:, :, None], M, axis=2)
        with np.errstate(all="ignore"):
            want = np.asarray(fn(arr))
        want = np.reshape(want, (len(pts), -1))[:, 0].astype(bool)
        bad = np.nonzero(want != got)[0]
        if len(bad):
            # tolerate points numerically on a boundary
            d = 1e-9 * max(scale, 1.0)
            still = [i for i in bad if all(eval_program(atoms, code, (pts[i] + d * e)[None, :])[0] == got[i] for e in np.vstack([np.eye(3), -np.eye(3)]))]
            if still:
                raise BoundsCompileError("compiled bounds program disagrees with inbounds_function at %d of %d test points, e.g. p=%r" % (len(still), len(pts), pts[still[0]]))


def compile_bounds(fn, scale=10.0, check=True):
    atoms, code = trace_bounds(fn)
    if check:
        verify(fn, atoms, code, scale)
    return atoms, code
