"""Seeded random effect assets for differential testing (oracle vs product).

Random float expression trees over the operators of SURVEY.md section 8a row A23 with static type tracking
(scalar / vec2 / vec3 / vec4), assigned to random attributes in init and update, mixed with the stock
modifiers. Everything stays float so that every generated asset type-checks; integer / bool operators have
their own zoo assets in test_lowering_cpu.py.
"""
import numpy as np

import bevy_hanabi_amd as bh

h = bh
A = bh.Attribute
F = h.ValueType(h.ScalarType.Float)
VEC = {1: F, 2: h.VectorType.VEC2F, 3: h.VectorType.VEC3F, 4: h.VectorType.VEC4F}
ATTR_BY_WIDTH = {1: [A.F32_0, A.F32_1, A.F32_2, A.SIZE, A.ALPHA], 2: [A.F32X2_0, A.F32X2_1, A.SIZE2], 3: [A.F32X3_0, A.F32X3_1, A.AXIS_X],
                 4: [A.F32X4_0, A.F32X4_1, A.HDR_COLOR]}

UNARY_SAFE = ["abs", "ceil", "floor", "fract", "round", "saturate", "sign", "sin", "cos", "exp2", "atan"]
BINARY = ["add", "sub", "mul", "min", "max", "step"]


class Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.w = h.ExprWriter()
        self.readable = {}   # width -> attributes holding a value at this point
        self.has_age = False

    def lit(self, width):
        v = self.rng.uniform(-3.0, 3.0, width).round(3)
        return self.w.lit(float(v[0])) if width == 1 else self.w.lit(tuple(float(x) for x in v))

    def leaf(self, width, ctx):
        r = self.rng.random()
        if r < 0.30:
            return self.lit(width)
        if r < 0.55:
            return self.w.rand(VEC[width])
        if r < 0.70 and self.readable.get(width):
            return self.w.attr(self.readable[width][int(self.rng.integers(len(self.readable[width])))])
        if r < 0.78 and width == 1:
            return self.w.time() if self.rng.random() < 0.5 else self.w.delta_time()
        if r < 0.86 and width == 3 and ctx == "update":
            return self.w.attr(A.POSITION if self.rng.random() < 0.5 else A.VELOCITY)
        if r < 0.92 and width == 1 and ctx == "update" and self.has_age:
            return self.w.attr(A.AGE)
        return self.lit(width)

    def expr(self, width, depth, ctx):
        if depth == 0 or self.rng.random() < 0.18:
            return self.leaf(width, ctx)
        r = self.rng.random()
        if r < 0.30:
            x = self.expr(width, depth - 1, ctx)
            return getattr(x, UNARY_SAFE[int(self.rng.integers(len(UNARY_SAFE)))])()
        if r < 0.62:
            op = BINARY[int(self.rng.integers(len(BINARY)))]
            a = self.expr(width, depth - 1, ctx)
            b = self.expr(width if self.rng.random() < 0.6 or op in ("min", "max", "step") else 1, depth - 1, ctx)
            if op == "add":
                return a + b
            if op == "sub":
                return a - b
            if op == "mul":
                return a * b
            return getattr(a, op)(b)
        if r < 0.70:
            return self.expr(width, depth - 1, ctx).mix(self.expr(width, depth - 1, ctx), self.expr(width, depth - 1, ctx).saturate())
        if r < 0.76:
            lo = self.expr(width, depth - 1, ctx)
            return self.expr(width, depth - 1, ctx).clamp(lo, lo + self.w.lit(1.5))
        if r < 0.82 and width == 1:   # reductions of a vector
            v = self.expr(int(self.rng.integers(2, 5)), depth - 1, ctx)
            k = self.rng.random()
            return v.length() if k < 0.4 else (v.dot(v) if k < 0.7 else v.x())
        if r < 0.88 and width == 3:
            k = self.rng.random()
            a = self.expr(3, depth - 1, ctx)
            if k < 0.4:
                return a.cross(self.expr(3, depth - 1, ctx))
            if k < 0.7:
                return (a + self.w.lit((0.0, 0.5, 0.0))).normalized()
            return self.expr(1, depth - 1, ctx).vec3(self.expr(1, depth - 1, ctx), self.expr(1, depth - 1, ctx))
        if r < 0.92 and width == 2:
            return self.expr(1, depth - 1, ctx).vec2(self.expr(1, depth - 1, ctx))
        if r < 0.96 and width == 4:
            return self.expr(3, depth - 1, ctx).vec4_xyz_w(self.expr(1, depth - 1, ctx))
        if width == 1:  # rand_uniform needs statically typed operands (expr.rs:1162-1190): literals
            return self.lit(1).uniform(self.lit(1))
        return self.leaf(width, ctx)

    def asset(self, capacity):
        rng = self.rng
        w = self.w
        init, update = [], []
        # position / velocity / age / lifetime
        k = rng.random()
        if k < 0.4:
            init.append(h.SetPositionSphereModifier(self.lit(3).expr(), w.lit(float(rng.uniform(0.2, 2.0))).expr(),
                                                    h.ShapeDimension.Volume if rng.random() < 0.5 else h.ShapeDimension.Surface))
        elif k < 0.7:
            init.append(h.SetPositionCircleModifier(self.lit(3).expr(), w.lit((0.0, 1.0, 0.0)).expr(), w.lit(float(rng.uniform(0.2, 2.0))).expr(),
                                                    h.ShapeDimension.Volume if rng.random() < 0.5 else h.ShapeDimension.Surface))
        else:
            init.append(h.SetAttributeModifier(A.POSITION, self.expr(3, 2, "init").expr()))
        init.append(h.SetAttributeModifier(A.VELOCITY, self.expr(3, 3, "init").expr()))
        has_age = self.has_age = rng.random() < 0.85
        if has_age:
            init.append(h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()))
            init.append(h.SetAttributeModifier(A.LIFETIME, w.lit(0.15).uniform(w.lit(float(rng.uniform(0.3, 1.2)))).expr()))
        for _ in range(int(rng.integers(1, 4))):
            width = int(rng.integers(1, 5))
            attr = ATTR_BY_WIDTH[width][int(rng.integers(len(ATTR_BY_WIDTH[width])))]
            init.append(h.SetAttributeModifier(attr, self.expr(width, 3, "init").expr()))
            self.readable.setdefault(width, [])
            if attr not in self.readable[width]:
                self.readable[width].append(attr)
        # update stack
        pool = ["drag", "accel", "radial", "tangent", "conform", "kill_sphere", "kill_aabb", "set", "set"]
        for _ in range(int(rng.integers(0, 6))):
            m = pool[int(rng.integers(len(pool)))]
            uniform_only = rng.random() < 0.5   # keep some stacks on the streaming kernel
            ex = (lambda wd: self.lit(wd)) if uniform_only else (lambda wd: self.expr(wd, 2, "update"))
            if m == "drag":
                update.append(h.LinearDragModifier(ex(1).abs().expr()))
            elif m == "accel":
                update.append(h.AccelModifier(ex(3).expr()))
            elif m == "radial":
                update.append(h.RadialAccelModifier(ex(3).expr(), ex(1).expr()))
            elif m == "tangent":
                update.append(h.TangentAccelModifier(ex(3).expr(), w.lit((0.0, 1.0, 0.0)).expr(), ex(1).expr()))
            elif m == "conform":
                update.append(h.ConformToSphereModifier(ex(3).expr(), w.lit(1.0).expr(), w.lit(4.0).expr(), ex(1).abs().expr(), w.lit(3.0).expr()))
            elif m == "kill_sphere":
                update.append(h.KillSphereModifier(ex(3).expr(), w.lit(float(rng.uniform(4.0, 60.0))).expr(), rng.random() < 0.2))
            elif m == "kill_aabb":
                update.append(h.KillAabbModifier(ex(3).expr(), w.lit((6.0, 6.0, 6.0)).expr(), False))
            else:
                width = int(rng.integers(1, 5))
                attr = ATTR_BY_WIDTH[width][int(rng.integers(len(ATTR_BY_WIDTH[width])))]
                update.append(h.SetAttributeModifier(attr, self.expr(width, 2, "update").expr()))
                self.readable.setdefault(width, [])
                if attr not in self.readable[width]:
                    self.readable[width].append(attr)
        asset = h.EffectAsset(capacity, h.SpawnerSettings.once(float(capacity)), w.finish())
        for m in init:
            asset.init(m)
        for m in update:
            asset.update(m)
        asset.motion_integration = [h.MotionIntegration.PostUpdate, h.MotionIntegration.PreUpdate, h.MotionIntegration.None_][int(rng.integers(3))]
        asset.simulation_space = h.SimulationSpace.Global if rng.random() < 0.5 else h.SimulationSpace.Local
        return asset


def random_asset(seed, capacity=400):
    return Gen(seed).asset(capacity)


def random_frames(seed, capacity, n=36):
    from helpers import Frame, frame_seed
    rng = np.random.default_rng(seed + 977)
    xf = np.array([1, 0, 0, rng.uniform(-2, 2), 0, 1, 0, rng.uniform(-2, 2), 0, 0, 1, rng.uniform(-2, 2)], dtype=np.float32)
    out = []
    for f in range(n):
        spawn = capacity // 2 if f == 0 else (int(rng.integers(0, capacity // 3)) if rng.random() < 0.25 else 0)
        out.append(Frame(1 / 60 if rng.random() < 0.8 else 1 / 30, spawn, frame_seed(seed * 131 + f), xf, time=f / 60))
    return out
