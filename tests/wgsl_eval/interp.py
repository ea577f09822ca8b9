"""TEST INFRASTRUCTURE: an interpreter for the WGSL subset the reference's effect compiler emits, vectorised over particles with numpy.

The reference defines an effect by the WGSL text `EffectShaderSources::generate` (src/lib.rs:805-1336) pastes into vfx_init.wgsl /
vfx_update.wgsl. The product lowers the same modifier / expression graph to a bytecode program (host/lowering.cpp), and the CPU oracle
(oracle/hanabi_oracle.c) walks the graph itself: two readings of the reference by one author. This file is a third one that shares nothing
with either: it reads the generated TEXT (host/wgsl.cpp, `generate_wgsl`) with a tokenizer, a Pratt parser and a numpy evaluator written
against the WGSL specification - IEEE binary32 arithmetic by numpy, transcendental builtins by numpy's float32 routines (NOT hanabi-math) -,
so agreement between the three pins the float semantics independently (tests/test_wgsl_eval.py: lists and counters exact, floats 1e-5).

Scope: statements `let` / `var` / assignment (= += -= *= /=) / `if` (+ else) / `return;` / call statements; expressions with the WGSL
operators, swizzles, constructors, casts and the builtin functions the reference's expression graph can name; user functions with
`ptr<function, Particle>` and `mat4x4<f32>` parameters (the modifiers' helpers). One lane per particle; control flow by lane masks.

WGSL abstract numerics are modelled: `1.` / `3` are AbstractFloat / AbstractInt (python float / int here) until they meet a concrete type.
"""
import re

import numpy as np

F32, I32, U32, BOOL = np.float32, np.int32, np.uint32, np.bool_


# ---- values --------------------------------------------------------------------------------------------------------------------------
class AF(float):
    """AbstractFloat literal or constant expression (evaluated in binary64)."""


class AI(int):
    """AbstractInt literal or constant expression (64-bit in WGSL; python int here)."""


def is_abstract(v):
    return isinstance(v, (AF, AI))


def concretise(v, like=None):
    """An abstract value takes the element type of the concrete value it meets (AbstractInt -> i32 / u32 / f32, AbstractFloat -> f32);
    on its own (a `let`) it becomes i32 / f32."""
    if not is_abstract(v):
        return v
    if like is not None and not is_abstract(like):
        dt = like.dtype
        if isinstance(v, AF) and dt != F32:
            raise WgslError("an AbstractFloat cannot become " + str(dt))
        if dt == BOOL:
            raise WgslError("a number cannot become bool")
    else:
        dt = F32 if isinstance(v, AF) else I32
    with np.errstate(over="ignore"):
        if dt == F32:
            return np.array([float(v)], dtype=F32)
        if not (np.iinfo(dt).min <= int(v) <= np.iinfo(dt).max):
            raise WgslError(f"{int(v)} is not representable as {dt.__name__}")
        return np.array([int(v)], dtype=dt)


class WgslError(Exception):
    pass


def comps(v):
    """number of components: 1 for a scalar"""
    return 1 if v.ndim == 1 else v.shape[1]


def lanes(v):
    return v.shape[0]


def align(a, b):
    """Broadcast a scalar against a vector (N,) vs (M,k)."""
    if a.ndim == 1 and b.ndim == 2:
        a = a[:, None]
    elif a.ndim == 2 and b.ndim == 1:
        b = b[:, None]
    elif a.ndim == 2 and b.ndim == 2 and a.shape[1] != b.shape[1] and 1 not in (a.shape[1], b.shape[1]):   # (a column of 1: a scalar aligned earlier)
        raise WgslError(f"vector size mismatch: {a.shape[1]} vs {b.shape[1]}")
    return a, b


def unify(a, b):
    """operands of a binary operator / builtin: resolve abstract numerics, check element types"""
    if is_abstract(a) and is_abstract(b):
        return a, b
    a, b = concretise(a, b), concretise(b, a)
    if a.dtype != b.dtype:
        raise WgslError(f"type mismatch: {a.dtype} vs {b.dtype}")
    return a, b


# ---- tokenizer -----------------------------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*)
  | (?P<num>(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fu]?|0x[0-9a-fA-F]+u?)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>\+=|-=|\*=|/=|<=|>=|==|!=|&&|\|\||->|[-+*/%<>=!&|^~.,;:(){}\[\]])
""", re.X)


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = TOKEN.match(src, pos)
        if not m:
            raise WgslError(f"cannot tokenize at {src[pos:pos + 30]!r}")
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group()))
    out.append(("eof", ""))
    return out


# ---- parser (AST as tuples) ------------------------------------------------------------------------------------------------------------
BINARY_PRECEDENCE = {"||": 1, "&&": 2, "|": 3, "^": 4, "&": 5, "==": 6, "!=": 6, "<": 7, "<=": 7, ">": 7, ">=": 7, "+": 9, "-": 9, "*": 10, "/": 10, "%": 10}
TYPE_NAMES = {"f32", "i32", "u32", "bool", "vec2", "vec3", "vec4", "vec2f", "vec3f", "vec4f", "vec2i", "vec3i", "vec4i", "vec2u", "vec3u", "vec4u", "mat4x4", "ptr"}


class Parser:
    def __init__(self, src):
        self.t = tokenize(src)
        self.i = 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def eat(self, text):
        if self.peek()[1] == text:
            self.i += 1
            return True
        return False

    def expect(self, text):
        if not self.eat(text):
            raise WgslError(f"expected {text!r}, found {self.peek()[1]!r}")

    # -- types: f32, vec3<f32>, ptr<function, Particle>, mat4x4<f32>
    def parse_type(self):
        name = self.next()[1]
        params = []
        if self.peek()[1] == "<" and name in TYPE_NAMES:
            self.next()
            while True:
                params.append(self.parse_type())
                if not self.eat(","):
                    break
            self.expect(">")
        return (name, tuple(params))

    # -- expressions
    def parse_expr(self, min_prec=1):
        lhs = self.parse_unary()
        while True:
            op = self.peek()[1]
            prec = BINARY_PRECEDENCE.get(op) if self.peek()[0] == "op" else None
            if prec is None or prec < min_prec:
                return lhs
            self.next()
            rhs = self.parse_expr(prec + 1)
            lhs = ("bin", op, lhs, rhs)

    def parse_unary(self):
        kind, text = self.peek()
        if kind == "op" and text in ("-", "!", "*", "&", "~"):
            self.next()
            return ("un", text, self.parse_unary())
        return self.parse_postfix(self.parse_primary())

    def parse_primary(self):
        kind, text = self.next()
        if kind == "num":
            return ("num", text)
        if kind == "op" and text == "(":
            e = self.parse_expr()
            self.expect(")")
            return ("paren", e)
        if kind == "id":
            if text in ("true", "false"):
                return ("bool", text == "true")
            if text in TYPE_NAMES and (self.peek()[1] == "<" or self.peek()[1] == "("):
                self.i -= 1
                ty = self.parse_type()
                return ("ctor", ty, self.parse_args())
            if text == "bitcast":
                self.expect("<")
                ty = self.parse_type()
                self.expect(">")
                return ("bitcast", ty, self.parse_args())
            if self.peek()[1] == "(":
                return ("call", text, self.parse_args())
            return ("var", text)
        raise WgslError(f"unexpected token {text!r}")

    def parse_args(self):
        self.expect("(")
        args = []
        if not self.eat(")"):
            while True:
                args.append(self.parse_expr())
                if self.eat(")"):
                    break
                self.expect(",")
        return args

    def parse_postfix(self, e):
        while True:
            if self.eat("."):
                e = ("member", e, self.next()[1])
            elif self.eat("["):
                idx = self.parse_expr()
                self.expect("]")
                e = ("index", e, idx)
            else:
                return e

    # -- statements
    def parse_block(self):
        self.expect("{")
        out = []
        while not self.eat("}"):
            out.append(self.parse_stmt())
        return out

    def parse_stmts_until_eof(self):
        out = []
        while self.peek()[0] != "eof":
            out.append(self.parse_stmt())
        return out

    def parse_stmt(self):
        kind, text = self.peek()
        if text == "{":
            return ("block", self.parse_block())
        if text in ("let", "var"):
            self.next()
            name = self.next()[1]
            ty = None
            if self.eat(":"):
                ty = self.parse_type()
            init = None
            if self.eat("="):
                init = self.parse_expr()
            self.expect(";")
            return ("decl", text, name, ty, init)
        if text == "if":
            self.next()
            cond = self.parse_expr()
            then = self.parse_block()
            other = None
            if self.eat("else"):
                other = [self.parse_stmt()] if self.peek()[1] == "if" else self.parse_block()
            return ("if", cond, then, other)
        if text == "return":
            self.next()
            value = None if self.peek()[1] == ";" else self.parse_expr()
            self.expect(";")
            return ("return", value)
        e = self.parse_expr()
        op = self.peek()[1]
        if op in ("=", "+=", "-=", "*=", "/="):
            self.next()
            rhs = self.parse_expr()
            self.eat(";")
            return ("assign", op, e, rhs)
        self.eat(";")    # (the reference's EmitSpawnEvent statement ends with `}` and no semicolon)
        return ("expr", e)

    def parse_functions(self):
        fns = {}
        while self.peek()[0] != "eof":
            self.expect("fn")
            name = self.next()[1]
            self.expect("(")
            params = []
            if not self.eat(")"):
                while True:
                    pname = self.next()[1]
                    self.expect(":")
                    params.append((pname, self.parse_type()))
                    if self.eat(")"):
                        break
                    self.expect(",")
            if self.eat("->"):
                self.parse_type()
            fns[name] = (params, self.parse_block())
        return fns


# ---- the builtin functions of WGSL (numpy, IEEE binary32) -----------------------------------------------------------------------------
def _f(v):
    v = concretise(v)
    if v.dtype != F32:
        raise WgslError(f"expected f32, got {v.dtype}")
    return v


def _dot(a, b):
    a, b = unify(a, b)
    a, b = concretise(a), concretise(b)
    if a.ndim != 2 or b.ndim != 2:
        raise WgslError("dot of non-vectors")
    s = a[:, 0] * b[:, 0]
    for k in range(1, a.shape[1]):
        s = s + a[:, k] * b[:, k]
    return s


def _length(v):
    v = _f(v)
    return np.abs(v) if v.ndim == 1 else np.sqrt(_dot(v, v))


def _unpack(u, signed):
    u = concretise(u)
    out = np.zeros((lanes(u), 4), F32)
    for k in range(4):
        b = (u >> U32(8 * k)) & U32(0xFF)
        if signed:
            sb = b.astype(np.int32)
            sb = np.where(sb > 127, sb - 256, sb)
            out[:, k] = np.maximum(sb.astype(F32) / F32(127.0), F32(-1.0))
        else:
            out[:, k] = b.astype(F32) / F32(255.0)
    return out


def _pack(v, signed):
    v = _f(v)
    out = np.zeros(lanes(v), U32)
    for k in range(4):
        c = v[:, k]
        if signed:
            q = np.floor(F32(0.5) + F32(127.0) * np.minimum(F32(1.0), np.maximum(F32(-1.0), c))).astype(np.int32).astype(U32) & U32(0xFF)
        else:
            q = np.floor(F32(0.5) + F32(255.0) * np.minimum(F32(1.0), np.maximum(F32(0.0), c))).astype(U32)
        out = out | (q << U32(8 * k))
    return out


def _wmin(a, b):
    a, b = unify(a, b)
    if is_abstract(a):
        return type(a)(b if b < a else a)
    a, b = align(a, b)
    return np.where(b < a, b, a)            # WGSL: min(e1, e2) = e2 < e1 ? e2 : e1


def _wmax(a, b):
    a, b = unify(a, b)
    if is_abstract(a):
        return type(a)(b if a < b else a)
    a, b = align(a, b)
    return np.where(a < b, b, a)


def _clamp(x, lo, hi):
    return _wmin(_wmax(x, lo), hi)


def _mix(a, b, t):
    a, b = unify(a, b)
    a, t = unify(a, t)
    b = concretise(b, a)
    a, b, t = concretise(a), concretise(b), concretise(t)
    a, b = align(a, b)
    a, t = align(a, t)
    b, t = align(b, t)
    return a * (F32(1.0) - t) + b * t


def _smoothstep(lo, hi, x):
    lo, hi, x = (_f(concretise(v, np.zeros(1, F32))) for v in (lo, hi, x))
    lo, x = align(lo, x)
    hi, x = align(hi, x)
    lo, hi = align(lo, hi)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = _clamp((x - lo) / (hi - lo), np.zeros(1, F32), np.ones(1, F32))
    return t * t * (F32(3.0) - F32(2.0) * t)


def _cross(a, b):
    a, b = _f(a), _f(b)
    n = max(lanes(a), lanes(b))
    a, b = np.broadcast_to(a, (n, 3)), np.broadcast_to(b, (n, 3))
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def _normalize(v):
    v = _f(v)
    with np.errstate(divide="ignore", invalid="ignore"):
        return v / _length(v)[:, None]


def _unary_float(fn):
    def g(x):
        with np.errstate(all="ignore"):
            return fn(_f(x)).astype(F32)
    return g


def _sign(x):
    x = concretise(x)
    return np.sign(x).astype(x.dtype)


def _abs(x):
    x = concretise(x)
    if x.dtype == U32:
        return x
    with np.errstate(over="ignore"):
        return np.abs(x)


def _step(edge, x):
    edge, x = unify(edge, x)
    edge, x = align(concretise(edge), concretise(x))
    return np.where(edge <= x, F32(1.0), F32(0.0)).astype(F32)


def _pow(x, y):
    x, y = unify(x, y)
    x, y = align(_f(x), _f(y))
    with np.errstate(all="ignore"):
        return np.power(x, y).astype(F32)


def _atan2(y, x):
    y, x = unify(y, x)
    y, x = align(_f(y), _f(x))
    return np.arctan2(y, x).astype(F32)


def _distance(a, b):
    a, b = unify(a, b)
    a, b = align(_f(a), _f(b))
    return _length(a - b)


def _all(v):
    v = concretise(v)
    return v if v.ndim == 1 else np.all(v, axis=1)


def _any(v):
    v = concretise(v)
    return v if v.ndim == 1 else np.any(v, axis=1)


BUILTINS = {
    "abs": _abs, "acos": _unary_float(np.arccos), "asin": _unary_float(np.arcsin), "atan": _unary_float(np.arctan), "atan2": _atan2,
    "all": _all, "any": _any, "ceil": _unary_float(np.ceil), "cos": _unary_float(np.cos), "cross": _cross, "distance": _distance, "dot": _dot,
    "exp": _unary_float(np.exp), "exp2": _unary_float(np.exp2), "floor": _unary_float(np.floor), "fract": _unary_float(lambda x: x - np.floor(x)),
    "inverseSqrt": _unary_float(lambda x: F32(1.0) / np.sqrt(x)), "length": _length, "log": _unary_float(np.log), "log2": _unary_float(np.log2),
    "max": _wmax, "min": _wmin, "mix": _mix, "clamp": _clamp, "smoothstep": _smoothstep, "normalize": _normalize,
    "pack4x8snorm": lambda v: _pack(v, True), "pack4x8unorm": lambda v: _pack(v, False),
    "unpack4x8snorm": lambda u: _unpack(u, True), "unpack4x8unorm": lambda u: _unpack(u, False),
    "pow": _pow, "round": _unary_float(np.rint), "saturate": lambda x: _clamp(x, np.zeros(1, F32), np.ones(1, F32)), "sign": _sign,
    "sin": _unary_float(np.sin), "sqrt": _unary_float(np.sqrt), "step": _step, "tan": _unary_float(np.tan),
}


# ---- value conversion (WGSL value constructors) ---------------------------------------------------------------------------------------
def convert_scalar(v, dt):
    if is_abstract(v):
        if dt == BOOL:
            return np.array([bool(v)])
        if dt == F32:
            return np.array([float(v)], F32)
        return np.array([int(v)]).astype(np.int64).astype(dt)
    if v.dtype == dt:
        return v
    if dt == BOOL:
        return v != 0
    if v.dtype == BOOL:
        return v.astype(dt)
    if v.dtype == F32 and dt in (I32, U32):   # truncation towards zero, saturating; NaN -> 0
        info = np.iinfo(dt)
        with np.errstate(invalid="ignore"):
            t = np.trunc(np.nan_to_num(v.astype(np.float64), nan=0.0, posinf=float(info.max), neginf=float(info.min)))
        return np.clip(t, info.min, info.max).astype(np.int64).astype(dt)
    if dt == F32:
        return v.astype(F32)
    return v.astype(np.int64).astype(dt) if v.dtype == I32 else v.astype(dt)   # i32 <-> u32: reinterpretation of the bits


ELEM = {"f32": F32, "i32": I32, "u32": U32, "bool": BOOL}
VEC_ALIAS = {"f": "f32", "i": "i32", "u": "u32"}


def type_of(ty):
    """(elem dtype or None, component count) of a parsed type"""
    name, params = ty
    if name in ELEM:
        return ELEM[name], 1
    m = re.fullmatch(r"vec([234])([fiu]?)", name)
    if m:
        n = int(m.group(1))
        if m.group(2):
            return ELEM[VEC_ALIAS[m.group(2)]], n
        return (ELEM[params[0][0]] if params else None), n
    raise WgslError(f"unsupported type {name}")


def construct(ty, args):
    dt, n = type_of(ty)
    if n == 1:
        if len(args) != 1:
            raise WgslError("scalar constructor takes one value")
        a = args[0]
        if not is_abstract(a) and a.ndim != 1:
            raise WgslError("scalar constructor of a vector")
        return convert_scalar(a, dt)
    # vectors: splat, component-wise, or a mix of vectors and scalars whose component counts add up
    if dt is None:   # `vec4(xyz, w)`: element type inferred
        conc = [a for a in args if not is_abstract(a)]
        dt = conc[0].dtype if conc else (F32 if any(isinstance(a, AF) for a in args) else I32)
    parts = []
    for a in args:
        a = convert_scalar(a, dt) if (is_abstract(a) or a.ndim == 1) else (a if a.dtype == dt else np.stack([convert_scalar(a[:, k], dt) for k in range(a.shape[1])], axis=1))
        parts.append(a[:, None] if a.ndim == 1 else a)
    total = sum(p.shape[1] for p in parts)
    nl = max(p.shape[0] for p in parts)
    if len(parts) == 1 and total == 1:
        return np.broadcast_to(parts[0], (nl, n)).copy()
    if len(parts) == 1 and total == n:
        return parts[0]
    if total != n:
        raise WgslError(f"vec{n} constructor with {total} components")
    return np.concatenate([np.broadcast_to(p, (nl, p.shape[1])) for p in parts], axis=1)


SWIZZLE = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


# ---- evaluator ---------------------------------------------------------------------------------------------------------------------------
class Particle:
    """`var particle: Particle`: a struct of per-lane arrays."""

    def __init__(self, fields):
        self.fields = fields


class Struct:
    def __init__(self, **fields):
        self.fields = fields


class Ptr:
    def __init__(self, target):
        self.target = target


class Mat4:
    """column-major mat4x4<f32>; cols[j] is an (1 or N, 4) array"""

    def __init__(self, cols):
        self.cols = cols


class Interp:
    def __init__(self, n_lanes, globals_, functions, hooks=None):
        self.n = n_lanes
        self.globals = globals_          # name -> value (arrays, Struct, Particle, Mat4, python callables for native functions)
        self.functions = functions       # name -> (params, body)
        self.hooks = hooks or {}         # native functions that need the interpreter (frand & co: they update `seed` under the lane mask)
        self.scopes = [dict()]
        self.mask = np.ones(n_lanes, bool)
        self.returned = [np.zeros(n_lanes, bool)]

    # -- scopes
    def lookup(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s, name
        if name in self.globals:
            return self.globals, name
        raise WgslError(f"unknown identifier {name}")

    def active(self):
        return self.mask & ~self.returned[-1]

    # -- expressions
    def number(self, text):
        if text.startswith("0x"):
            u = text.endswith("u")
            v = int(text.rstrip("u"), 16)
            return np.array([v], U32) if u else AI(v)
        if text.endswith("u"):
            return np.array([int(text[:-1])], U32)
        if text.endswith("f"):
            return np.array([float(text[:-1])], F32)
        if any(c in text for c in ".eE"):
            return AF(float(text))
        return AI(int(text))

    def ev(self, e):
        k = e[0]
        if k == "num":
            return self.number(e[1])
        if k == "bool":
            return np.array([e[1]])
        if k == "paren":
            return self.ev(e[1])
        if k == "var":
            s, n = self.lookup(e[1])
            return s[n]
        if k == "un":
            return self.unary(e[1], e[2])
        if k == "bin":
            return self.binary(e[1], self.ev(e[2]), self.ev(e[3]))
        if k == "member":
            return self.member(self.ev(e[1]), e[2])
        if k == "index":
            base, idx = self.ev(e[1]), self.ev(e[2])
            i = int(idx) if is_abstract(idx) else int(idx[0])
            if isinstance(base, Mat4):
                return base.cols[i]
            if isinstance(base, list):
                return base[i]
            return base[:, i]
        if k == "ctor":
            return construct(e[1], [self.ev(a) for a in e[2]])
        if k == "bitcast":
            v = concretise(self.ev(e[2][0]))
            return v.view(type_of(e[1])[0])
        if k == "call":
            return self.call(e[1], e[2])
        raise WgslError(f"cannot evaluate {k}")

    def unary(self, op, operand):
        if op == "&":
            v = self.ev(operand)
            return Ptr(v)
        v = self.ev(operand)
        if op == "*":
            return v.target
        if op == "-":
            if is_abstract(v):
                return type(v)(-v)
            with np.errstate(over="ignore"):
                return (-v).astype(v.dtype)
        if op == "!":
            return ~concretise(v)
        raise WgslError(f"unary {op}")

    def binary(self, op, a, b):
        if isinstance(a, Mat4):   # mat4x4 * vec4
            b = concretise(b)
            out = None
            for j in range(4):
                term = a.cols[j] * b[:, j:j + 1]
                out = term if out is None else out + term
            return out
        a, b = unify(a, b)
        if is_abstract(a):
            return self.abstract_binary(op, a, b)
        a, b = align(a, b)
        with np.errstate(all="ignore"):
            if op in ("&&", "||"):
                return (a & b) if op == "&&" else (a | b)
            if op in ("<", "<=", ">", ">=", "==", "!="):
                return {"<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal, "==": np.equal, "!=": np.not_equal}[op](a, b)
            dt = a.dtype
            if op == "+":
                return (a + b).astype(dt)
            if op == "-":
                return (a - b).astype(dt)
            if op == "*":
                return (a * b).astype(dt)
            if op == "/":
                if dt == F32:
                    return (a / b).astype(F32)
                return int_div(a, b)
            if op == "%":
                if dt == F32:
                    return (a - b * np.trunc(a / b)).astype(F32)
                return int_rem(a, b)
            if op in ("&", "|", "^"):
                return {"&": np.bitwise_and, "|": np.bitwise_or, "^": np.bitwise_xor}[op](a, b)
        raise WgslError(f"binary {op}")

    def abstract_binary(self, op, a, b):
        both_int = isinstance(a, AI) and isinstance(b, AI)
        if op in ("<", "<=", ">", ">=", "==", "!="):
            return np.array([{"<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b, "==": a == b, "!=": a != b}[op]])
        if both_int:
            x, y = int(a), int(b)
            if op == "/":
                if y == 0:
                    raise WgslError("constant division by zero")
                q = abs(x) // abs(y)
                return AI(q if (x < 0) == (y < 0) else -q)
            if op == "%":
                if y == 0:
                    raise WgslError("constant remainder by zero")
                q = abs(x) // abs(y)
                q = q if (x < 0) == (y < 0) else -q
                return AI(x - y * q)
            return AI({"+": x + y, "-": x - y, "*": x * y}[op])
        x, y = float(a), float(b)
        if op == "/":
            return AF(x / y if y != 0 else float("inf") if x > 0 else float("-inf") if x < 0 else float("nan"))
        if op == "%":
            return AF(x - y * float(int(x / y)))
        return AF({"+": x + y, "-": x - y, "*": x * y}[op])

    def member(self, base, name):
        if isinstance(base, Ptr):    # `p.field` on a pointer is `(*p).field` (WGSL's automatic dereference; the reference's TangentAccel relies on it)
            base = base.target
        if isinstance(base, (Particle, Struct)):
            return base.fields[name]
        if isinstance(base, np.ndarray) and base.ndim == 2:
            idx = [SWIZZLE[c] for c in name]
            return base[:, idx[0]] if len(idx) == 1 else base[:, idx]
        raise WgslError(f"no member {name}")

    def call(self, name, arg_exprs):
        if name in self.hooks:
            return self.hooks[name](self, *[self.ev(a) for a in arg_exprs])
        if name in self.functions:
            return self.call_user(name, [self.ev(a) for a in arg_exprs])
        if name in BUILTINS:
            args = [self.ev(a) for a in arg_exprs]
            if all(is_abstract(a) for a in args):     # a builtin of constants: evaluated in f32 here (the front end would use f64; unpinned)
                args = [concretise(AF(float(a))) for a in args]
            return BUILTINS[name](*args)
        raise WgslError(f"unknown function {name}")

    def call_user(self, name, args):
        params, body = self.functions[name]
        if len(params) != len(args):
            raise WgslError(f"{name}: {len(args)} arguments for {len(params)} parameters")
        scope = {}
        for (pname, _pty), a in zip(params, args):
            scope[pname] = a
        saved_scopes, saved_mask = self.scopes, self.mask
        self.scopes = [scope]
        self.mask = self.active()
        self.returned.append(np.zeros(self.n, bool))
        try:
            self.exec_block(body)
        finally:
            self.returned.pop()
            self.scopes, self.mask = saved_scopes, saved_mask
        return None

    # -- statements
    def exec_block(self, stmts):
        self.scopes.append({})
        try:
            for s in stmts:
                self.exec(s)
        finally:
            self.scopes.pop()

    def exec(self, s):
        k = s[0]
        if k == "block":
            self.exec_block(s[1])
        elif k == "decl":
            _, _kw, name, ty, init = s
            v = self.ev(init) if init is not None else None
            if isinstance(v, (AF, AI)):
                v = concretise(v) if ty is None else convert_scalar(v, type_of(ty)[0])
            if isinstance(v, np.ndarray):
                v = np.array(v)    # a copy: `var` bindings are assigned through lane masks later
            self.scopes[-1][name] = v
        elif k == "assign":
            self.assign(s[1], s[2], self.ev(s[3]))
        elif k == "if":
            cond = concretise(self.ev(s[1]))
            cond = np.broadcast_to(cond, (self.n,))
            saved = self.mask
            self.mask = saved & cond
            if self.active().any():
                self.exec_block(s[2])
            if s[3] is not None:
                self.mask = saved & ~cond
                if self.active().any():
                    self.exec_block(s[3])
            self.mask = saved
        elif k == "return":
            self.returned[-1] = self.returned[-1] | self.mask
        elif k == "expr":
            self.ev(s[1])
        else:
            raise WgslError(f"statement {k}")

    def assign(self, op, target, value):
        # resolve the place: a variable, a struct field (particle.x / (*particle).x), or a vector component
        def place(t):
            if t[0] == "paren":
                return place(t[1])
            if t[0] == "var":
                s, n = self.lookup(t[1])
                return s, n
            if t[0] == "member":
                base = self.ev(t[1])
                if isinstance(base, Ptr):
                    base = base.target
                if isinstance(base, (Particle, Struct)):
                    return base.fields, t[2]
                raise WgslError("assignment to a vector component is not emitted by the reference")
            raise WgslError(f"cannot assign to {t[0]}")
        container, key = place(target)
        old = container[key]
        if op != "=":
            value = self.binary(op[0], old, value)
        value = concretise(value, old)
        if value.dtype != old.dtype:
            raise WgslError(f"assignment of {value.dtype} to {old.dtype}")
        m = self.active()
        shape = (self.n,) + old.shape[1:]
        value = np.broadcast_to(value[:, None] if (value.ndim == 1 and old.ndim == 2) else value, shape)
        oldb = np.broadcast_to(old, shape)
        container[key] = np.where(m[:, None] if old.ndim == 2 else m, value, oldb)


def int_div(a, b):
    """WGSL integer division: x / 0 = x; i32: MIN / -1 = MIN; truncation towards zero."""
    dt = a.dtype
    a64, b64 = a.astype(np.int64), b.astype(np.int64)
    safe = np.where(b64 == 0, 1, b64)
    q = np.abs(a64) // np.abs(safe)
    q = np.where((a64 < 0) != (safe < 0), -q, q)
    q = np.where(b64 == 0, a64, q)
    if dt == I32:
        q = np.where((a64 == -2 ** 31) & (b64 == -1), a64, q)
    return q.astype(dt)


def int_rem(a, b):
    dt = a.dtype
    a64, b64 = a.astype(np.int64), b.astype(np.int64)
    safe = np.where(b64 == 0, 1, b64)
    q = np.abs(a64) // np.abs(safe)
    q = np.where((a64 < 0) != (safe < 0), -q, q)
    r = a64 - safe * q
    r = np.where(b64 == 0, 0, r)
    if dt == I32:
        r = np.where((a64 == -2 ** 31) & (b64 == -1), 0, r)
    return r.astype(dt)
