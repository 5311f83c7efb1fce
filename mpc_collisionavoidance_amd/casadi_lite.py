"""A small CasADi-compatible symbolic layer: just enough of `from casadi import *` for the reference's
model files (catkin_ws/src/nmpc_ca/scripts/*/usv_model.py: `MX.sym`, `vertcat`, arithmetic, `sin cos tan
atan2 sqrt fabs tanh exp log if_else`, comparisons, and the names `np`, `types`, `pi` that star-import
brings along) to run unchanged and to hand their expression graphs to the device-code generator
(codegen.py).  Scalar expression DAG with structural interning (common sub-expressions are shared);
vectors are lists of scalars.  Derivative conventions are CasADi's: d|a| = sign(a), if_else branches
differentiate separately, d sqrt(a) = 1/(2 sqrt(a)) (0/0 -> NaN at the origin).

`install()` registers this module as `casadi` (and the look-alike acados_template as `acados_template`)
in sys.modules so that `from casadi import *` / `from acados_template import ...` resolve to it.
"""
import itertools
import math
import sys
import types  # noqa: F401  (re-exported: the reference's model files use `types.SimpleNamespace` from the star import)

import numpy as _numpy


class _NumpyCompat(types.ModuleType):
    """What the star import hands the model files as `np`: numpy, plus the `np.math` alias that numpy < 1.25
    had and some of the reference's files use (scripts/usv_guidance/usv_model.py:163)."""
    math = math

    def __getattr__(self, name):
        return getattr(_numpy, name)


np = _NumpyCompat("numpy_compat")

pi = math.pi
inf = math.inf

_UNARY = ("neg", "sin", "cos", "tan", "sqrt", "fabs", "exp", "log", "tanh", "sign", "not")
_BINARY = ("add", "sub", "mul", "div", "pow", "atan2", "fmin", "fmax", "lt", "le", "gt", "ge", "eq", "ne", "and", "or")


class MX:
    """Scalar expression node. kind: 'sym' | 'const' | op name; args: child nodes."""
    __array_priority__ = 1000
    __array_ufunc__ = None  # numpy scalars defer to the reflected operators below
    _intern = {}
    _symid = itertools.count()
    __slots__ = ("kind", "args", "value", "name", "key")

    def __new__(cls, kind, args=(), value=None, name=None):
        if kind == "sym":
            key = ("sym", next(cls._symid))  # every MX.sym is a distinct symbol, whatever its name
        elif kind == "const":
            key = ("const", float(value))
        else:
            key = (kind,) + tuple(a.key for a in args)
        hit = cls._intern.get(key)
        if hit is not None:
            return hit
        self = object.__new__(cls)
        self.kind, self.args, self.value, self.name, self.key = kind, tuple(args), value, name, key
        cls._intern[key] = self
        return self

    # ---- construction
    @staticmethod
    def sym(name, n=1, m=1):
        if m != 1:
            raise NotImplementedError("matrix symbols are not needed by the USV models")
        if n == 1:
            return MX("sym", name=name)
        return MXVec([MX("sym", name="%s_%d" % (name, i)) for i in range(n)])

    @staticmethod
    def const(v):
        return MX("const", value=float(v))

    # ---- shape protocol (scalars are 1x1)
    def size(self):
        return (1, 1)

    @property
    def shape(self):
        return (1, 1)

    def __getitem__(self, i):
        if i in (0, -1, (0, 0)):
            return self
        raise IndexError("index %r out of range for a 1x1 expression" % (i,))

    def __len__(self):
        return 1

    def is_constant(self):
        return self.kind == "const"

    def __repr__(self):
        if self.kind == "sym":
            return self.name
        if self.kind == "const":
            return repr(self.value)
        return "%s(%s)" % (self.kind, ", ".join(map(repr, self.args)))

    def __hash__(self):
        return hash(self.key)

    def __bool__(self):
        raise TypeError("the truth value of a symbolic expression is undefined; use if_else")

    # ---- arithmetic with light constant folding (keeps generated code small)
    def __add__(self, o): return _bin("add", self, o)
    def __radd__(self, o): return _bin("add", o, self)
    def __sub__(self, o): return _bin("sub", self, o)
    def __rsub__(self, o): return _bin("sub", o, self)
    def __mul__(self, o): return _bin("mul", self, o)
    def __rmul__(self, o): return _bin("mul", o, self)
    def __truediv__(self, o): return _bin("div", self, o)
    def __rtruediv__(self, o): return _bin("div", o, self)
    def __pow__(self, o): return _bin("pow", self, o)
    def __rpow__(self, o): return _bin("pow", o, self)
    def __neg__(self): return _un("neg", self)
    def __pos__(self): return self
    def __lt__(self, o): return _bin("lt", self, o)
    def __le__(self, o): return _bin("le", self, o)
    def __gt__(self, o): return _bin("gt", self, o)
    def __ge__(self, o): return _bin("ge", self, o)
    # == / != keep Python semantics (identity) so that nodes stay usable as dict keys


class MXVec:
    """Column vector of scalar expressions (what vertcat returns)."""
    __array_priority__ = 1000
    __array_ufunc__ = None

    def __init__(self, items):
        self.items = list(items)

    def size(self):
        return (len(self.items), 1)

    @property
    def shape(self):
        return (len(self.items), 1)

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)

    def __getitem__(self, i):
        r = self.items[i]
        return MXVec(r) if isinstance(r, list) else r

    def _zip(self, o, op, swap=False):
        if isinstance(o, MXVec):
            if len(o) != len(self):
                raise ValueError("dimension mismatch: %d vs %d" % (len(self), len(o)))
            pairs = zip(self.items, o.items)
        else:
            pairs = ((a, o) for a in self.items)
        return MXVec([_bin(op, b, a) if swap else _bin(op, a, b) for a, b in pairs])

    def __add__(self, o): return self._zip(o, "add")
    def __radd__(self, o): return self._zip(o, "add", True)
    def __sub__(self, o): return self._zip(o, "sub")
    def __rsub__(self, o): return self._zip(o, "sub", True)
    def __mul__(self, o): return self._zip(o, "mul")
    def __rmul__(self, o): return self._zip(o, "mul", True)
    def __truediv__(self, o): return self._zip(o, "div")
    def __neg__(self): return MXVec([-a for a in self.items])

    def __repr__(self):
        return "vertcat(%s)" % ", ".join(map(repr, self.items))


def _wrap(v):
    if isinstance(v, MX):
        return v
    if isinstance(v, MXVec):
        if len(v) == 1:
            return v.items[0]
        raise TypeError("vector where a scalar expression is expected")
    if isinstance(v, (bool, _numpy.bool_)):
        return MX.const(1.0 if v else 0.0)
    return MX.const(float(v))


_FOLD1 = {"neg": lambda a: -a, "sin": math.sin, "cos": math.cos, "tan": math.tan, "sqrt": math.sqrt, "fabs": abs,
          "exp": math.exp, "log": math.log, "tanh": math.tanh, "sign": lambda a: (a > 0) - (a < 0),
          "not": lambda a: 0.0 if a else 1.0}
_FOLD2 = {"add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b, "div": lambda a, b: a / b,
          "pow": lambda a, b: a ** b, "atan2": math.atan2, "fmin": min, "fmax": max,
          "lt": lambda a, b: float(a < b), "le": lambda a, b: float(a <= b), "gt": lambda a, b: float(a > b),
          "ge": lambda a, b: float(a >= b), "eq": lambda a, b: float(a == b), "ne": lambda a, b: float(a != b),
          "and": lambda a, b: float(bool(a) and bool(b)), "or": lambda a, b: float(bool(a) or bool(b))}


def _un(op, a):
    if isinstance(a, MXVec):
        return MXVec([_un(op, x) for x in a.items])
    a = _wrap(a)
    if a.kind == "const":
        try:
            return MX.const(_FOLD1[op](a.value))
        except (ValueError, ZeroDivisionError):
            pass
    if op == "neg" and a.kind == "neg":
        return a.args[0]
    return MX(op, (a,))


def _bin(op, a, b):
    if isinstance(a, MXVec) or isinstance(b, MXVec):
        va = a if isinstance(a, MXVec) else None
        vb = b if isinstance(b, MXVec) else None
        n = len(va) if va is not None else len(vb)
        return MXVec([_bin(op, va.items[i] if va is not None else a, vb.items[i] if vb is not None else b) for i in range(n)])
    a, b = _wrap(a), _wrap(b)
    if a.kind == "const" and b.kind == "const":
        try:
            return MX.const(_FOLD2[op](a.value, b.value))
        except (ValueError, ZeroDivisionError, OverflowError):
            pass
    # identities that keep the graph (and the generated code) free of trivial operations
    if op == "add":
        if a.kind == "const" and a.value == 0.0: return b
        if b.kind == "const" and b.value == 0.0: return a
    elif op == "sub":
        if b.kind == "const" and b.value == 0.0: return a
        if a.kind == "const" and a.value == 0.0: return _un("neg", b)
    elif op == "mul":
        for p, q in ((a, b), (b, a)):
            if p.kind == "const":
                if p.value == 0.0: return MX.const(0.0)
                if p.value == 1.0: return q
                if p.value == -1.0: return _un("neg", q)
    elif op == "div":
        if b.kind == "const" and b.value == 1.0: return a
        if a.kind == "const" and a.value == 0.0: return MX.const(0.0)
    elif op == "pow":
        if b.kind == "const" and b.value == 1.0: return a
        if b.kind == "const" and b.value == 2.0: return MX("mul", (a, a))
    return MX(op, (a, b))


# ---- the casadi free functions the model files call
def vertcat(*args):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        args = tuple(args[0])  # vertcat([]) / vertcat([a, b])
    out = []
    for a in args:
        if isinstance(a, MXVec):
            out.extend(a.items)
        else:
            out.append(_wrap(a))
    return MXVec(out)


def sin(a): return _un("sin", a)
def cos(a): return _un("cos", a)
def tan(a): return _un("tan", a)
def sqrt(a): return _un("sqrt", a)
def fabs(a): return _un("fabs", a)
def exp(a): return _un("exp", a)
def log(a): return _un("log", a)
def tanh(a): return _un("tanh", a)
def sign(a): return _un("sign", a)
def atan2(a, b): return _bin("atan2", a, b)
def fmin(a, b): return _bin("fmin", a, b)
def fmax(a, b): return _bin("fmax", a, b)
def power(a, b): return _bin("pow", a, b)
def logic_and(a, b): return _bin("and", a, b)
def logic_or(a, b): return _bin("or", a, b)
def logic_not(a): return _un("not", a)


def if_else(c, a, b):
    c, a, b = _wrap(c), _wrap(a), _wrap(b)
    if c.kind == "const":
        return a if c.value != 0.0 else b
    if a is b:
        return a
    return MX("if_else", (c, a, b))


def Function(*a, **k):
    raise NotImplementedError("casadi.Function is not part of the supported subset")


def interpolant(*a, **k):
    raise NotImplementedError("casadi.interpolant is not part of the supported subset (race-car example is out of scope)")


SX = MX  # the reference only uses MX; SX.sym behaves the same here


# ---------------------------------------------------------------------------- graph utilities
def scalars(expr):
    """List of scalar nodes of a scalar / vector / None."""
    if expr is None:
        return []
    if isinstance(expr, MXVec):
        return list(expr.items)
    if isinstance(expr, (list, tuple)):
        return [_wrap(e) for e in expr]
    return [_wrap(expr)]


def topo(outputs):
    """Nodes reachable from `outputs`, children first (iterative DFS)."""
    seen, order = set(), []
    for root in outputs:
        stack = [(root, False)]
        while stack:
            n, done = stack.pop()
            if done:
                order.append(n)
                continue
            if n.key in seen:
                continue
            seen.add(n.key)
            stack.append((n, True))
            for a in n.args:
                if a.key not in seen:
                    stack.append((a, False))
    return order


def depends_on(outputs):
    """Set of symbol nodes every output depends on: {output index: set(sym nodes)}."""
    dep = {}
    for n in topo(outputs):
        if n.kind == "sym":
            dep[n.key] = frozenset([n])
        else:
            s = frozenset()
            for a in n.args:
                s = s | dep[a.key]
            dep[n.key] = s
    return [dep[o.key] for o in outputs]


def evaluate(outputs, values):
    """Numeric evaluation of scalar nodes; values: {sym node: float}."""
    val = {}
    for n in topo(outputs):
        k = n.kind
        if k == "sym":
            val[n.key] = float(values[n])
        elif k == "const":
            val[n.key] = n.value
        elif k == "if_else":
            c, a, b = (val[x.key] for x in n.args)
            val[n.key] = a if c != 0.0 else b
        elif k in _FOLD1:
            a = val[n.args[0].key]
            try:
                val[n.key] = float(_FOLD1[k](a))
            except ValueError:
                val[n.key] = float("nan")
        else:
            a, b = val[n.args[0].key], val[n.args[1].key]
            try:
                val[n.key] = float(_FOLD2[k](a, b))
            except (ValueError, ZeroDivisionError):
                val[n.key] = float("nan")
    return [val[o.key] for o in outputs]


def install():
    """Make `import casadi` / `import acados_template` resolve to this package's look-alikes."""
    from . import acados_template as _at
    sys.modules["casadi"] = sys.modules[__name__]
    sys.modules["acados_template"] = _at


__all__ = ["MX", "SX", "MXVec", "vertcat", "sin", "cos", "tan", "sqrt", "fabs", "exp", "log", "tanh", "sign", "atan2",
           "fmin", "fmax", "power", "if_else", "logic_and", "logic_or", "logic_not", "Function", "interpolant",
           "pi", "inf", "np", "types"]
