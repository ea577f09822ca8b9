// Lowering: EffectAsset -> HnbProgram blob (include/hanabi_amd.h).
//
// Replaces EffectShaderSources::generate (reference src/lib.rs:805-1336): instead of WGSL
// text, every modifier's `apply()` and every expression's `eval()` is lowered to
// (a) a UNIFORM stream — all sub-expressions that depend only on literals, properties and
//     simulation clocks; the runtime evaluates it on the host per instance per frame into
//     the parameter block ("U registers");
// (b) an INIT stream and (c) an UPDATE stream of per-particle instructions whose operands
//     are V registers (per particle) or U registers.
//
// Semantics preserved from the reference:
//  * modifiers are applied in insertion order (lib.rs:1028-1038, 1078-1088);
//  * an expression is a *string* in the reference: non-side-effect expressions are
//    re-evaluated at every use (they read the particle as it is at that statement), while
//    side-effect expressions (rand / rand_uniform / rand_normal) are hoisted to one
//    `let varN` at first evaluation and memoised per ShaderWriter (expr.rs:1812-1824,
//    modifier/mod.rs:309-319); function-style modifiers evaluate in a fresh writer
//    (modifier/mod.rs:321-349);
//  * operand evaluation order is left, then right (expr.rs:1149-1152);
//  * float literals carry 6 decimals (lib.rs:264-269);
//  * age / reap / Euler placement (lib.rs:1106-1133, 1223-1258); PREV/NEXT = 0xffffffff at
//    init and never written back by update (vfx_init.wgsl:176-181, lib.rs:1266-1281).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>

#include "hanabi.hpp"

namespace hanabi {
namespace {

struct Loc {
    bool uniform = false;  // U register (parameter block) vs V register (per particle)
    uint8_t reg = 0;
    ValueType type;
    // WGSL abstract numeric (see "abstract numerics" below): no register yet, the value is a host-side constant
    uint8_t abs = 0;  // 0: concrete, 1: AbstractInt (ai), 2: AbstractFloat (ad)
    int64_t ai = 0;
    double ad = 0.0;
};

[[noreturn]] void type_error(const std::string& msg) { throw ExprError(ExprError::TypeError, msg); }

struct AttrSlot {
    Attribute attr;
    uint8_t reg = 0, ncomp = 0;
    uint8_t upd_flags = 0;
};

struct Stream {
    std::vector<uint64_t> code;
};

// a, b, c: a plain register index (V register, or U register of the uniform stream), or
// HNB_OPERAND_DECODED_U | index for a U register read by a varying stream (-> operand byte + bank bit).
uint64_t encode(uint32_t op, uint32_t d, uint32_t a, uint32_t b, uint32_t c, uint32_t width, bool ba, bool bb, bool bc, uint32_t aux) {
    auto byte = [](uint32_t v) { return (v & HNB_OPERAND_DECODED_U) ? (HNB_OPERAND_U | (v & 0x7fu)) : (v & 0xffu); };
    auto bank = [](uint32_t v) { return (v & HNB_OPERAND_DECODED_U) ? ((v >> 7) & 1u) : 0u; };
    const uint32_t w0 = (op & 0xffu) | ((d & 0xffu) << 8) | (byte(a) << 16) | (byte(b) << 24);
    const uint32_t w1 = byte(c) | (((width - 1u) & 3u) << 8) | (ba ? 1u << 10 : 0u) | (bb ? 1u << 11 : 0u) | (bc ? 1u << 12 : 0u) |
                        (bank(a) << 13) | (bank(b) << 14) | (bank(c) << 15) | ((aux & 0xffffu) << 16);
    return (uint64_t)w0 | ((uint64_t)w1 << 32);
}

enum class StreamId { Uniform = 0, Init = 1, Update = 2 };

// One ShaderWriter of the reference: the main writer of a stream or a function writer.
struct Writer {
    StreamId stream;
    std::map<uint32_t, Loc> memo;  // side-effect expressions already hoisted (expr cache)
};

// Internal: the stream does not fit the V file it was lowered for.
struct RegisterPressure : ShaderGenerateError { using ShaderGenerateError::ShaderGenerateError; };

class Lowerer {
   public:
    // `vlimit`: size of the V file the program may use (HNB_VM_MAX_REGS, or HNB_VM_MAX_REGS_WIDE for the retry
    // of a program that does not fit the fast file).
    explicit Lowerer(const EffectAsset& a, uint32_t vlimit = HNB_VM_MAX_REGS) : asset_(a), mod_(a.module()), vlimit_(vlimit) {}
    std::vector<uint8_t> run();

   private:
    const EffectAsset& asset_;
    const Module& mod_;
    Stream uni_, init_, upd_;
    std::vector<AttrSlot> attrs_;
    int attr_index_[HNB_ATTR_COUNT];
    uint32_t attr_end_ = HNB_REG_FIRST_FREE;  // first V register after the attributes
    // U allocation
    uint32_t utop_ = 0;
    std::map<uint32_t, Loc> umemo_;                                    // uniform expression -> U registers
    std::map<std::array<uint32_t, 6>, uint32_t> ulit_;                  // literal dedupe
    std::map<uint32_t, int> uniform_cache_;                             // is_uniform() memo
    // V allocation (per stream while lowering it)
    uint32_t vlimit_ = HNB_VM_MAX_REGS;
    uint32_t vtop_ = 0, ptop_ = HNB_VM_MAX_REGS, vmax_ = 0;
    [[noreturn]] void out_of_registers() const {
        throw RegisterPressure("expression too complex: more than " + std::to_string(vlimit_) + " per-particle registers needed");
    }
    std::vector<uint32_t> prop_offset_;
    uint32_t prop_words_ = 0;

    Stream& S(StreamId id) { return id == StreamId::Uniform ? uni_ : (id == StreamId::Init ? init_ : upd_); }
    bool has(Attribute a) const { return attr_index_[a.id] >= 0; }
    const AttrSlot& slot(Attribute a) const { return attrs_[attr_index_[a.id]]; }
    void touch(StreamId s, Attribute a, bool write) {
        if (s != StreamId::Update || !has(a)) return;
        AttrSlot& sl = attrs_[attr_index_[a.id]];
        sl.upd_flags |= HNB_ATTR_UPD_LOAD;
        if (write && a != Attribute::PREV && a != Attribute::NEXT) sl.upd_flags |= HNB_ATTR_UPD_STORE;
    }

    uint32_t alloc_u(uint32_t n) {
        if (utop_ + n > HNB_VM_MAX_UREGS)
            throw ShaderGenerateError("effect uses more than " + std::to_string(HNB_VM_MAX_UREGS) + " uniform parameter words");
        const uint32_t r = utop_;
        utop_ += n;
        return r;
    }
    uint32_t alloc_v(uint32_t n) {
        if (vtop_ + n > ptop_) out_of_registers();
        const uint32_t r = vtop_;
        vtop_ += n;
        vmax_ = std::max(vmax_, vtop_);
        return r;
    }
    uint32_t alloc_persistent(uint32_t n) {
        if (ptop_ < vtop_ + n) out_of_registers();
        ptop_ -= n;
        vmax_ = vlimit_;
        return ptop_;
    }
    static uint32_t opnd(const Loc& l) { return l.uniform ? (HNB_OPERAND_DECODED_U | l.reg) : l.reg; }

    // ---- emission --------------------------------------------------------------------------
    std::vector<uint32_t> parent_attrs_;  // HnbAttr ids the init stream reads from the parent particle
    uint32_t n_event_channels_ = 0;       // child event channels the update stream appends to
    void note_parent_attr(Attribute a) {
        for (uint32_t id : parent_attrs_) if (id == a.id) return;
        parent_attrs_.push_back((uint32_t)a.id);
    }
    void emit(StreamId s, uint32_t op, uint32_t d, uint32_t a, uint32_t b, uint32_t c, uint32_t width, bool ba, bool bb, bool bc,
              uint32_t aux = 0) {
        S(s).code.push_back(encode(op, d, a, b, c, width, ba, bb, bc, aux));
    }
    // element-wise op with 1..3 operands; unused operand fields alias operand a (broadcast)
    Loc emit_elementwise(StreamId s, uint32_t op, ValueType out_type, const Loc* x, const Loc* y, const Loc* z, int base = -1) {
        const bool uniform_out = s == StreamId::Uniform;
        const uint32_t w = out_type.count;
        Loc out = result_loc(s, out_type, base < 0 ? vtop_ : (uint32_t)base, {x, y, z}, true);
        auto field = [&](const Loc* l) -> uint32_t { return uniform_out ? l->reg : opnd(*l); };
        const uint32_t fa = field(x), fb = y ? field(y) : fa, fc = z ? field(z) : fa;
        const bool ba = x->type.count == 1 && w > 1, bb = y ? (y->type.count == 1 && w > 1) : true, bc = z ? (z->type.count == 1 && w > 1) : true;
        emit(s, op, out.reg, fa, fb, fc, w, ba || w == 1, bb || w == 1, bc || w == 1);
        return out;
    }
    // copy `src` (w components) into consecutive registers starting at dst (same stream space rules)
    void emit_mov(StreamId s, uint32_t dst, const Loc& src, uint32_t w, bool bcast = false) {
        const uint32_t f = s == StreamId::Uniform ? src.reg : opnd(src);
        emit(s, HNB_OP_MOV, dst, f, f, f, w, bcast || w == 1, true, true);
    }

    // ---- classification ------------------------------------------------------------------------
    bool is_uniform(ExprHandle h) {
        auto it = uniform_cache_.find(h.id);
        if (it != uniform_cache_.end()) return it->second != 0;
        const Expr& e = mod_.try_get(h);
        bool u = false;
        switch (e.kind) {
            case Expr::Kind::Literal: case Expr::Kind::Property: u = true; break;
            case Expr::Kind::BuiltIn:
                u = !(e.builtin == BuiltInOperator::Rand || e.builtin == BuiltInOperator::IsAlive || e.builtin == BuiltInOperator::AlphaCutoff);
                break;
            case Expr::Kind::Attribute: case Expr::Kind::ParentAttribute: case Expr::Kind::TextureSample: u = false; break;
            case Expr::Kind::Unary: case Expr::Kind::Cast: u = is_uniform(e.a); break;
            case Expr::Kind::Binary: u = !e.has_side_effect() && is_uniform(e.a) && is_uniform(e.b); break;
            case Expr::Kind::Ternary: u = is_uniform(e.a) && is_uniform(e.b) && is_uniform(e.c); break;
        }
        uniform_cache_[h.id] = u ? 1 : 0;
        return u;
    }

    // ---- literals -----------------------------------------------------------------------------------
    Loc load_literal(const Value& v, bool as_written = true) {
        Value r = v;
        if (as_written && v.type.elem == ScalarType::Float)
            for (int i = 0; i < v.type.count; ++i) r.set_f(i, round_literal_f32(v.get_f(i)));
        std::array<uint32_t, 6> key = {(uint32_t)r.type.elem, r.type.count, 0, 0, 0, 0};
        for (int i = 0; i < r.type.count; ++i) key[2 + i] = r.bits[i];
        Loc l;
        l.uniform = true;
        l.type = r.type;
        auto it = ulit_.find(key);
        if (it != ulit_.end()) { l.reg = (uint8_t)it->second; return l; }
        l.reg = (uint8_t)alloc_u(r.type.count);
        for (int i = 0; i < r.type.count; ++i) S(StreamId::Uniform).code.push_back((uint64_t)(HNB_OP_LOADK | ((l.reg + i) << 8)) | ((uint64_t)r.bits[i] << 32));
        ulit_[key] = l.reg;
        return l;
    }
    Loc lit_f32(float x) { return load_literal(Value(x)); }

    // ---- abstract numerics ---------------------------------------------------------------------------
    // `ToWgslString` writes a scalar f32 literal as `5.` / `0.1` and a scalar i32 literal as `-3` (src/lib.rs:264-269,
    // 354-358): an AbstractFloat and an AbstractInt of WGSL. The front end evaluates expressions of abstract operands
    // only in f64 / i64, converts an abstract operand to the type of the concrete operand it meets (AbstractInt -> i32,
    // u32 or f32; AbstractFloat -> f32), and a `let` without a type concretises to i32 / f32. That is what makes
    // examples/instancing.rs:274 (`RadialAccelModifier::new(origin, writer.lit(-3).expr())`) a valid effect. u32
    // literals (`3u`), booleans and vector literals (`vec3<f32>(...)`) are concrete.
    static Loc abstract_int(int64_t i) {
        Loc l;
        l.uniform = true;
        l.type = ValueType(ScalarType::Int);
        l.reg = 0xff;
        l.abs = 1;
        l.ai = i;
        return l;
    }
    static Loc abstract_float(double d) {
        if (!std::isfinite(d)) type_error("abstract float constant expression is not finite");
        Loc l;
        l.uniform = true;
        l.type = ValueType(ScalarType::Float);
        l.reg = 0xff;
        l.abs = 2;
        l.ad = d;
        return l;
    }
    static double abs_d(const Loc& l) { return l.abs == 1 ? (double)l.ai : l.ad; }
    // conversion of an abstract value to a concrete scalar type: materialised as a uniform constant
    Loc conv(const Loc& l, ScalarType elem) {
        if (!l.abs) return l;
        if (l.abs == 2) {
            if (elem != ScalarType::Float) type_error("an abstract float constant does not convert to " + ValueType(elem).to_string());
            const float f = (float)l.ad;
            if (!std::isfinite(f)) type_error("abstract float constant out of range for f32");
            return load_literal(Value(f), false);
        }
        switch (elem) {
            case ScalarType::Float: return load_literal(Value((float)l.ai), false);
            case ScalarType::Int:
                if (l.ai < INT32_MIN || l.ai > INT32_MAX) type_error("abstract int constant out of range for i32");
                return load_literal(Value((int32_t)l.ai));
            case ScalarType::Uint:
                if (l.ai < 0 || l.ai > (int64_t)UINT32_MAX) type_error("abstract int constant out of range for u32");
                return load_literal(Value((uint32_t)l.ai));
            default: type_error("an abstract int constant does not convert to bool");
        }
    }
    Loc conc(const Loc& l) { return l.abs ? conv(l, l.abs == 1 ? ScalarType::Int : ScalarType::Float) : l; }
    // operands of one operator or constructor: abstract ones take the concrete one's element type; all abstract: f32 if
    // any is a float, else i32
    void unify(std::initializer_list<Loc*> v) {
        bool have = false, any_float = false;
        ScalarType elem = ScalarType::Int;
        for (Loc* l : v) {
            if (!l->abs) { if (!have) { elem = l->type.elem; have = true; } }
            else if (l->abs == 2) any_float = true;
        }
        if (!have) elem = any_float ? ScalarType::Float : ScalarType::Int;
        for (Loc* l : v) *l = conv(*l, elem);
    }
    // `l op r` with both operands abstract: i64 when both are ints, f64 otherwise
    static Loc abstract_arith(BinaryOperator op, const Loc& l, const Loc& r) {
        if (l.abs == 1 && r.abs == 1) {
            const int64_t x = l.ai, y = r.ai;
            int64_t z = 0;
            bool bad = false;
            switch (op) {
                case BinaryOperator::Add: bad = __builtin_add_overflow(x, y, &z); break;
                case BinaryOperator::Sub: bad = __builtin_sub_overflow(x, y, &z); break;
                case BinaryOperator::Mul: bad = __builtin_mul_overflow(x, y, &z); break;
                case BinaryOperator::Div: if (y == 0 || (x == INT64_MIN && y == -1)) bad = true; else z = x / y; break;
                default: if (y == 0 || (x == INT64_MIN && y == -1)) bad = true; else z = x % y; break;
            }
            if (bad) type_error("abstract int constant expression overflows or divides by zero");
            return abstract_int(z);
        }
        const double x = abs_d(l), y = abs_d(r);
        switch (op) {
            case BinaryOperator::Add: return abstract_float(x + y);
            case BinaryOperator::Sub: return abstract_float(x - y);
            case BinaryOperator::Mul: return abstract_float(x * y);
            case BinaryOperator::Div: return abstract_float(x / y);
            default: return abstract_float(std::fmod(x, y));
        }
    }
    // a modifier parameter pasted into an expression of the template (`... * ({speed})`): an abstract value takes the
    // type the expression asks for. Parameters the template binds with `let` first go through eval(), which concretises.
    Loc eval_inline(Writer& w, ExprHandle h, ScalarType elem) {
        const Loc l = eval_abs(w, h);
        return l.abs ? conv(l, elem) : l;
    }
    Loc delta_time_loc() {
        Expr e;
        e.kind = Expr::Kind::BuiltIn;
        e.builtin = BuiltInOperator::DeltaTime;
        return eval_builtin_uniform(e);
    }
    std::map<int, Loc> builtin_memo_;
    Loc eval_builtin_uniform(const Expr& e) {
        const int field = e.builtin == BuiltInOperator::Time ? 0 : e.builtin == BuiltInOperator::DeltaTime ? 1
                          : e.builtin == BuiltInOperator::VirtualTime ? 2 : e.builtin == BuiltInOperator::VirtualDeltaTime ? 3
                          : e.builtin == BuiltInOperator::RealTime ? 4 : 5;
        auto it = builtin_memo_.find(field);
        if (it != builtin_memo_.end()) return it->second;
        Loc l;
        l.uniform = true;
        l.type = ValueType(ScalarType::Float);
        l.reg = (uint8_t)alloc_u(1);
        emit(StreamId::Uniform, HNB_OP_LDB, l.reg, (uint32_t)field, 0, 0, 1, false, false, false);
        builtin_memo_[field] = l;
        return l;
    }

    // ---- expression evaluation --------------------------------------------------------------------
    Loc eval(Writer& w, ExprHandle h) { return conc(eval_abs(w, h)); }
    Loc eval_abs(Writer& w, ExprHandle h) {
        const Expr& e = mod_.try_get(h);
        if (is_uniform(h)) {
            auto it = umemo_.find(h.id);
            if (it != umemo_.end()) return it->second;
            Writer uw{StreamId::Uniform, {}};
            Loc l = eval_node(uw, h, e, StreamId::Uniform);
            umemo_[h.id] = l;
            return l;
        }
        if (e.has_side_effect()) {
            auto it = w.memo.find(h.id);
            if (it != w.memo.end()) return it->second;
        }
        if (e.has_side_effect()) {
            // hoisted `let varN = ...;`: the value outlives the statement (and later statements of
            // this writer), so it is produced straight into a persistent register.
            const uint32_t base = vtop_;
            persistent_result_ = true;
            Loc l = eval_node(w, h, e, w.stream);
            vtop_ = base;  // operand temporaries of the draw are dead
            w.memo[h.id] = l;
            return l;
        }
        return eval_node(w, h, e, w.stream);
    }

    Loc eval_node(Writer& w, ExprHandle h, const Expr& e, StreamId s) {
        switch (e.kind) {
            case Expr::Kind::Literal:
                if (e.literal.type == ValueType(ScalarType::Int)) return abstract_int((int32_t)e.literal.bits[0]);
                if (e.literal.type == ValueType(ScalarType::Float) && std::isfinite(e.literal.get_f(0))) {
                    char buf[400];  // the f64 the front end reads back from the 6-decimal text
                    std::snprintf(buf, sizeof buf, "%.6f", (double)e.literal.get_f(0));
                    return abstract_float(std::strtod(buf, nullptr));
                }
                return load_literal(e.literal);
            case Expr::Kind::Property: {
                const Property* p = mod_.get_property(e.property);
                if (!p) throw ExprError(ExprError::PropertyError, "Unknown property handle in evaluation module.");
                Loc l;
                l.uniform = true;
                l.type = p->default_value.type;
                l.reg = (uint8_t)alloc_u(l.type.count);
                S(StreamId::Uniform).code.push_back((uint64_t)(HNB_OP_LDP | (l.reg << 8) | ((l.type.count - 1u) << 16)) |
                                                    ((uint64_t)prop_offset_[e.property.index()] << 32));
                return l;
            }
            case Expr::Kind::BuiltIn: return eval_builtin(w, e, s);
            case Expr::Kind::Attribute: return eval_attribute(e, s);
            case Expr::Kind::ParentAttribute: {
                // expr.rs: `parent_particle.<name>`; the variable only exists in the init shader of an
                // effect with a parent (READ_PARENT_PARTICLE, vfx_init.wgsl:166-171)
                if (s != StreamId::Init)
                    throw ExprError(ExprError::GraphEvalError, "parent attributes are only available in the init context (vfx_init.wgsl:166-171)");
                if (e.attribute == Attribute::ID || e.attribute == Attribute::PARTICLE_COUNTER)
                    throw ExprError(ExprError::GraphEvalError, "pseudo-attributes of the parent particle cannot be read");
                Loc l;
                l.type = e.attribute.value_type();
                l.reg = (uint8_t)alloc_v(l.type.count);
                emit(s, HNB_OP_LDPARENT, l.reg, 0, 0, 0, l.type.count, true, true, true, (uint32_t)e.attribute.id);
                note_parent_attr(e.attribute);
                return l;
            }
            case Expr::Kind::TextureSample:
                throw ExprError(ExprError::GraphEvalError, "texture sampling is only available in the render context");
            case Expr::Kind::Unary: return eval_unary(w, e, s);
            case Expr::Kind::Binary: return eval_binary(w, e, s);
            case Expr::Kind::Ternary: return eval_ternary(w, e, s);
            case Expr::Kind::Cast: return eval_cast(w, e, s);
        }
        (void)h;
        throw ShaderGenerateError("unreachable expression kind");
    }

    Loc eval_builtin(Writer&, const Expr& e, StreamId s) {
        switch (e.builtin) {
            case BuiltInOperator::Rand: {
                if (e.rand_type.elem != ScalarType::Float)
                    type_error("rand() of type " + e.rand_type.to_string() + " is not available: the simulation shaders only define frand/frand2/frand3/frand4");
                Loc l = rand_loc(e.rand_type);
                emit(s, HNB_OP_FRAND, l.reg, 0, 0, 0, l.type.count, true, true, true);
                return l;
            }
            case BuiltInOperator::IsAlive: {
                if (s != StreamId::Update) throw ExprError(ExprError::GraphEvalError, "is_alive is only defined in the update context");
                Loc l;
                l.type = ValueType(ScalarType::Bool);
                l.reg = (uint8_t)alloc_v(1);
                emit(s, HNB_OP_LDALIVE, l.reg, 0, 0, 0, 1, true, true, true);
                return l;
            }
            case BuiltInOperator::AlphaCutoff:
                throw ExprError(ExprError::GraphEvalError, "alpha_cutoff is only available in the render context");
            default: return eval_builtin_uniform(e);
        }
    }

    Loc eval_attribute(const Expr& e, StreamId s) {
        if (e.attribute == Attribute::ID) {  // `particle_index` (expr.rs:1353-1360)
            Loc l;
            l.type = ValueType(ScalarType::Uint);
            l.reg = (uint8_t)alloc_v(1);
            emit(s, HNB_OP_LDID, l.reg, 0, 0, 0, 1, true, true, true);
            return l;
        }
        if (e.attribute == Attribute::PARTICLE_COUNTER) {  // `particle_counter` only exists in vfx_init.wgsl
            if (s != StreamId::Init)
                throw ExprError(ExprError::GraphEvalError, "PARTICLE_COUNTER is only defined in the init context (vfx_init.wgsl:148)");
            Loc l;
            l.type = ValueType(ScalarType::Uint);
            l.reg = (uint8_t)alloc_v(1);
            emit(s, HNB_OP_LDPC, l.reg, 0, 0, 0, 1, true, true, true);
            return l;
        }
        const AttrSlot& sl = slot(e.attribute);
        touch(s, e.attribute, false);
        Loc l;
        l.type = e.attribute.value_type();
        if (sl.reg != HNB_REG_NONE) { l.reg = sl.reg; return l; }  // pinned: lives in registers
        // any other attribute is a memory operand: load it where the expression is evaluated
        l.reg = (uint8_t)alloc_v(sl.ncomp);
        emit(s, HNB_OP_LDA, l.reg, 0, 0, 0, sl.ncomp, true, true, true, (uint32_t)attr_index_[e.attribute.id]);
        return l;
    }

    static uint32_t pick(ScalarType t, uint32_t f, uint32_t i, uint32_t u, const char* what) {
        switch (t) {
            case ScalarType::Float: if (f) return f; break;
            case ScalarType::Int: if (i) return i; break;
            case ScalarType::Uint: if (u) return u; break;
            default: break;
        }
        type_error(std::string("operator ") + what + " is not defined for this operand type");
    }

    // `Expr::eval` prints a component access as `{expr}.x` and an infix operator as `({l}) op ({r})`, neither with parentheses around the
    // whole (expr.rs:1145-1149, 1164-1176): `(a * b).x()` reaches the shader as `(a) * (b).x`, which does not compile when `b` is a scalar
    // and otherwise selects a component of the RIGHT OPERAND only - never what the expression graph says. An independent execution of the
    // emitted text found this (tests/wgsl_eval); such a graph is rejected here rather than given either meaning. (A hoisted rand() operand
    // is a `varN` in the text: fine.)
    void reject_swizzle_of_infix(const Expr& e) {
        if (!(e.unary == UnaryOperator::X || e.unary == UnaryOperator::Y || e.unary == UnaryOperator::Z || e.unary == UnaryOperator::W)) return;
        const Expr& in = mod_.try_get(e.a);
        if (in.kind != Expr::Kind::Binary || in.has_side_effect()) return;
        switch (in.binary) {
            case BinaryOperator::Add: case BinaryOperator::Div: case BinaryOperator::GreaterThan: case BinaryOperator::GreaterThanOrEqual:
            case BinaryOperator::LessThan: case BinaryOperator::LessThanOrEqual: case BinaryOperator::Mul: case BinaryOperator::Remainder:
            case BinaryOperator::Sub:
                throw ExprError(ExprError::GraphEvalError,
                                "component access on an infix expression: the reference prints `(l) op (r).x` (no parentheses around the operation, "
                                "src/graph/expr.rs:1145-1149), a shader that does not compile or that takes the component of the right operand only");
            default: break;
        }
    }

    Loc eval_unary(Writer& w, const Expr& e, StreamId s) {
        reject_swizzle_of_infix(e);
        const int base = (int)vtop_;
        Loc x = eval_abs(w, e.a);
        if (x.abs) {
            // abs() / sign() keep an integer an integer; the other builtins only exist for floats (AbstractInt ->
            // AbstractFloat -> f32) or fail below on a scalar
            switch (e.unary) {
                case UnaryOperator::Abs: case UnaryOperator::Sign: case UnaryOperator::All: case UnaryOperator::Any:
                case UnaryOperator::X: case UnaryOperator::Y: case UnaryOperator::Z: case UnaryOperator::W:
                case UnaryOperator::Unpack4x8snorm: case UnaryOperator::Unpack4x8unorm: x = conc(x); break;
                default: x = conv(x, ScalarType::Float); break;
            }
        }
        const ValueType t = x.type;
        auto float_only = [&](const char* name) { if (!t.is_float()) type_error(std::string(name) + "() requires a floating-point operand, got " + t.to_string()); };
        auto simple = [&](uint32_t op) { return emit_elementwise(s, op, t, &x, nullptr, nullptr, base); };
        switch (e.unary) {
            case UnaryOperator::Abs:
                if (t.elem == ScalarType::Bool) type_error("abs() of a boolean");
                return simple(t.elem == ScalarType::Float ? HNB_OP_FABS : (t.elem == ScalarType::Int ? HNB_OP_IABS : HNB_OP_MOV));
            case UnaryOperator::Acos: float_only("acos"); return simple(HNB_OP_FACOS);
            case UnaryOperator::Asin: float_only("asin"); return simple(HNB_OP_FASIN);
            case UnaryOperator::Atan: float_only("atan"); return simple(HNB_OP_FATAN);
            case UnaryOperator::Ceil: float_only("ceil"); return simple(HNB_OP_FCEIL);
            case UnaryOperator::Cos: float_only("cos"); return simple(HNB_OP_FCOS);
            case UnaryOperator::Exp: float_only("exp"); return simple(HNB_OP_FEXP);
            case UnaryOperator::Exp2: float_only("exp2"); return simple(HNB_OP_FEXP2);
            case UnaryOperator::Floor: float_only("floor"); return simple(HNB_OP_FFLOOR);
            case UnaryOperator::Fract: float_only("fract"); return simple(HNB_OP_FFRACT);
            case UnaryOperator::InvSqrt: float_only("inverseSqrt"); return simple(HNB_OP_FRSQ);
            case UnaryOperator::Log: float_only("log"); return simple(HNB_OP_FLOG);
            case UnaryOperator::Log2: float_only("log2"); return simple(HNB_OP_FLOG2);
            case UnaryOperator::Round: float_only("round"); return simple(HNB_OP_FROUND);
            case UnaryOperator::Saturate: float_only("saturate"); return simple(HNB_OP_FSAT);
            case UnaryOperator::Sin: float_only("sin"); return simple(HNB_OP_FSIN);
            case UnaryOperator::Sqrt: float_only("sqrt"); return simple(HNB_OP_FSQRT);
            case UnaryOperator::Tan: float_only("tan"); return simple(HNB_OP_FTAN);
            case UnaryOperator::Sign:
                if (t.elem == ScalarType::Float) return simple(HNB_OP_FSIGN);
                if (t.elem == ScalarType::Int) return simple(HNB_OP_ISIGN);
                type_error("sign() requires a float or signed integer operand");
            case UnaryOperator::All:
            case UnaryOperator::Any: {
                if (t.elem != ScalarType::Bool) type_error("all()/any() require a boolean operand, got " + t.to_string());
                return emit_reduce(s, e.unary == UnaryOperator::All ? HNB_OP_ALL : HNB_OP_ANY, ValueType(ScalarType::Bool), x, nullptr, base);
            }
            case UnaryOperator::Length: float_only("length"); return emit_reduce(s, HNB_OP_LENGTH, ValueType(ScalarType::Float), x, nullptr, base);
            case UnaryOperator::Normalize: {
                if (!t.is_float() || !t.is_vector()) type_error("normalize() requires a floating-point vector, got " + t.to_string());
                Loc out = result_loc(s, t, (uint32_t)base, {&x}, false);
                emit(s, HNB_OP_NORMALIZE, out.reg, fld(s, x), fld(s, x), fld(s, x), t.count, false, true, true);
                return out;
            }
            case UnaryOperator::Pack4x8snorm:
            case UnaryOperator::Pack4x8unorm: {
                if (t != VectorType::VEC4F) type_error("pack4x8*norm() requires a vec4<f32>, got " + t.to_string());
                Loc out = result_loc(s, ValueType(ScalarType::Uint), (uint32_t)base, {&x}, false);
                emit(s, e.unary == UnaryOperator::Pack4x8snorm ? HNB_OP_PACK4SNORM : HNB_OP_PACK4UNORM, out.reg, fld(s, x), fld(s, x), fld(s, x), 1,
                     false, true, true);
                return out;
            }
            case UnaryOperator::Unpack4x8snorm:
            case UnaryOperator::Unpack4x8unorm: {
                if (t != ValueType(ScalarType::Uint)) type_error("unpack4x8*norm() requires a u32, got " + t.to_string());
                Loc out = result_loc(s, VectorType::VEC4F, (uint32_t)base, {&x}, false);
                emit(s, e.unary == UnaryOperator::Unpack4x8snorm ? HNB_OP_UNPACK4SNORM : HNB_OP_UNPACK4UNORM, out.reg, fld(s, x), fld(s, x),
                     fld(s, x), 4, true, true, true);
                return out;
            }
            case UnaryOperator::X: case UnaryOperator::Y: case UnaryOperator::Z: case UnaryOperator::W: {
                const uint32_t idx = e.unary == UnaryOperator::X ? 0 : e.unary == UnaryOperator::Y ? 1 : e.unary == UnaryOperator::Z ? 2 : 3;
                if (!t.is_vector() || idx >= t.count) type_error("invalid component access on " + t.to_string());
                Loc l = x;
                l.reg = (uint8_t)(x.reg + idx);
                l.type = ValueType(t.elem, 1);
                return l;
            }
        }
        throw ShaderGenerateError("unhandled unary operator");
    }

    uint32_t fld(StreamId s, const Loc& l) const { return s == StreamId::Uniform ? l.reg : opnd(l); }
    Loc new_loc(StreamId s, ValueType t) {
        Loc l;
        l.uniform = s == StreamId::Uniform;
        l.type = t;
        l.reg = (uint8_t)(l.uniform ? alloc_u(t.count) : alloc_v(t.count));
        return l;
    }
    // Result placement with stack discipline: `base` is the V stack top before the operands
    // were evaluated. The result is put at `base` (releasing the operands' temporaries) when
    // the interpreter's evaluation order makes that safe:
    //  * non-element-wise ops compute every output before storing any -> always safe;
    //  * element-wise ops read r[x+k] then write r[d+k] per component k -> safe unless a
    //    BROADCAST operand lives in [base, base+w-1) (it would be overwritten before its
    //    last read).
    Loc result_loc(StreamId s, ValueType t, uint32_t base, std::initializer_list<const Loc*> operands, bool elementwise) {
        if (s == StreamId::Uniform) return new_loc(s, t);
        const uint32_t w = t.count;
        bool safe = base + w <= ptop_;
        if (elementwise)
            for (const Loc* o : operands)
                if (o && !o->uniform && o->type.count == 1 && w > 1 && o->reg >= base && o->reg + 1u < base + w) safe = false;
        Loc l;
        l.type = t;
        if (safe) {
            vtop_ = base;
            l.reg = (uint8_t)alloc_v(w);
        } else {
            l.reg = (uint8_t)alloc_v(w);
        }
        return l;
    }
    // ops that read `x.type.count` components (and optionally y) and write one register
    Loc emit_reduce(StreamId s, uint32_t op, ValueType out_type, const Loc& x, const Loc* y, int base = -1) {
        Loc out = result_loc(s, out_type, base < 0 ? vtop_ : (uint32_t)base, {&x, y}, false);
        emit(s, op, out.reg, fld(s, x), y ? fld(s, *y) : fld(s, x), fld(s, x), x.type.count, false, false, true);
        return out;
    }

    // WGSL arithmetic typing: T op T, vecN<T> op T, T op vecN<T>
    static ValueType arith_type(const Loc& l, const Loc& r, const char* what) {
        if (l.type.elem != r.type.elem || l.type.elem == ScalarType::Bool)
            type_error(std::string("operands of '") + what + "' have incompatible types " + l.type.to_string() + " and " + r.type.to_string());
        if (l.type.count == r.type.count) return l.type;
        if (l.type.count == 1) return r.type;
        if (r.type.count == 1) return l.type;
        type_error(std::string("operands of '") + what + "' have mismatched vector sizes " + l.type.to_string() + " and " + r.type.to_string());
    }
    static void same_type(const Loc& l, const Loc& r, const char* what) {
        if (l.type != r.type) type_error(std::string(what) + " requires operands of the same type, got " + l.type.to_string() + " and " + r.type.to_string());
    }

    Loc eval_binary(Writer& w, const Expr& e, StreamId s) {
        if (e.binary == BinaryOperator::UniformRand || e.binary == BinaryOperator::NormalRand) return eval_rand_binary(w, e, s);
        const int base = (int)vtop_;
        Loc l = eval_abs(w, e.a);
        Loc r = eval_abs(w, e.b);
        if (l.abs || r.abs) {
            const bool both = l.abs && r.abs;
            switch (e.binary) {
                case BinaryOperator::Add: case BinaryOperator::Sub: case BinaryOperator::Mul: case BinaryOperator::Div: case BinaryOperator::Remainder:
                    if (both) return abstract_arith(e.binary, l, r);
                    if (l.abs) l = conv(l, r.type.elem); else r = conv(r, l.type.elem);
                    break;
                case BinaryOperator::LessThan: case BinaryOperator::LessThanOrEqual: case BinaryOperator::GreaterThan: case BinaryOperator::GreaterThanOrEqual:
                    if (both) {
                        bool t;
                        if (l.abs == 1 && r.abs == 1) {
                            const int64_t x = l.ai, y = r.ai;
                            t = e.binary == BinaryOperator::LessThan ? x < y : e.binary == BinaryOperator::LessThanOrEqual ? x <= y
                                : e.binary == BinaryOperator::GreaterThan ? x > y : x >= y;
                        } else {
                            const double x = abs_d(l), y = abs_d(r);
                            t = e.binary == BinaryOperator::LessThan ? x < y : e.binary == BinaryOperator::LessThanOrEqual ? x <= y
                                : e.binary == BinaryOperator::GreaterThan ? x > y : x >= y;
                        }
                        return load_literal(Value(t));
                    }
                    unify({&l, &r});
                    break;
                case BinaryOperator::Max: case BinaryOperator::Min:
                    if (both) {
                        const bool mx = e.binary == BinaryOperator::Max;
                        if (l.abs == 1 && r.abs == 1) return abstract_int(mx ? (l.ai < r.ai ? r.ai : l.ai) : (r.ai < l.ai ? r.ai : l.ai));
                        const double x = abs_d(l), y = abs_d(r);
                        return abstract_float(mx ? (x < y ? y : x) : (y < x ? y : x));
                    }
                    unify({&l, &r});
                    break;
                case BinaryOperator::Vec2: case BinaryOperator::Vec4XyzW: unify({&l, &r}); break;
                default: l = conv(l, ScalarType::Float); r = conv(r, ScalarType::Float); break;  // step, atan2, cross, dot, distance: float only
            }
        }
        switch (e.binary) {
            case BinaryOperator::Add: { const ValueType t = arith_type(l, r, "+"); return emit_elementwise(s, pick(t.elem, HNB_OP_FADD, HNB_OP_IADD, HNB_OP_IADD, "+"), t, &l, &r, nullptr, base); }
            case BinaryOperator::Sub: { const ValueType t = arith_type(l, r, "-"); return emit_elementwise(s, pick(t.elem, HNB_OP_FSUB, HNB_OP_ISUB, HNB_OP_ISUB, "-"), t, &l, &r, nullptr, base); }
            case BinaryOperator::Mul: { const ValueType t = arith_type(l, r, "*"); return emit_elementwise(s, pick(t.elem, HNB_OP_FMUL, HNB_OP_IMUL, HNB_OP_IMUL, "*"), t, &l, &r, nullptr, base); }
            case BinaryOperator::Div: { const ValueType t = arith_type(l, r, "/"); return emit_elementwise(s, pick(t.elem, HNB_OP_FDIV, HNB_OP_IDIV, HNB_OP_UDIV, "/"), t, &l, &r, nullptr, base); }
            case BinaryOperator::Remainder: { const ValueType t = arith_type(l, r, "%"); return emit_elementwise(s, pick(t.elem, HNB_OP_FREM, HNB_OP_IREM, HNB_OP_UREM, "%"), t, &l, &r, nullptr, base); }
            case BinaryOperator::LessThan: case BinaryOperator::LessThanOrEqual: case BinaryOperator::GreaterThan: case BinaryOperator::GreaterThanOrEqual: {
                same_type(l, r, "comparison");
                if (l.type.elem == ScalarType::Bool) type_error("ordering comparison of booleans");
                uint32_t op = 0;
                switch (e.binary) {
                    case BinaryOperator::LessThan: op = pick(l.type.elem, HNB_OP_FLT, HNB_OP_ILT, HNB_OP_ULT, "<"); break;
                    case BinaryOperator::LessThanOrEqual: op = pick(l.type.elem, HNB_OP_FLE, HNB_OP_ILE, HNB_OP_ULE, "<="); break;
                    case BinaryOperator::GreaterThan: op = pick(l.type.elem, HNB_OP_FGT, HNB_OP_IGT, HNB_OP_UGT, ">"); break;
                    default: op = pick(l.type.elem, HNB_OP_FGE, HNB_OP_IGE, HNB_OP_UGE, ">="); break;
                }
                return emit_elementwise(s, op, ValueType(ScalarType::Bool, l.type.count), &l, &r, nullptr, base);
            }
            case BinaryOperator::Max: same_type(l, r, "max()"); return emit_elementwise(s, pick(l.type.elem, HNB_OP_FMAX, HNB_OP_IMAX, HNB_OP_UMAX, "max"), l.type, &l, &r, nullptr, base);
            case BinaryOperator::Min: same_type(l, r, "min()"); return emit_elementwise(s, pick(l.type.elem, HNB_OP_FMIN, HNB_OP_IMIN, HNB_OP_UMIN, "min"), l.type, &l, &r, nullptr, base);
            case BinaryOperator::Step: same_type(l, r, "step()"); return emit_elementwise(s, pick(l.type.elem, HNB_OP_FSTEP, 0, 0, "step"), l.type, &l, &r, nullptr, base);
            case BinaryOperator::Atan2: same_type(l, r, "atan2()"); return emit_elementwise(s, pick(l.type.elem, HNB_OP_FATAN2, 0, 0, "atan2"), l.type, &l, &r, nullptr, base);
            case BinaryOperator::Cross: {
                if (l.type != VectorType::VEC3F || r.type != VectorType::VEC3F) type_error("cross() requires two vec3<f32> operands");
                Loc out = result_loc(s, VectorType::VEC3F, (uint32_t)base, {&l, &r}, false);
                emit(s, HNB_OP_CROSS, out.reg, fld(s, l), fld(s, r), fld(s, l), 3, false, false, true);
                return out;
            }
            case BinaryOperator::Dot:
                same_type(l, r, "dot()");
                if (!l.type.is_float() || !l.type.is_vector()) type_error("dot() requires floating-point vectors");
                return emit_reduce(s, HNB_OP_DOT, ValueType(ScalarType::Float), l, &r, base);
            case BinaryOperator::Distance:
                same_type(l, r, "distance()");
                if (!l.type.is_float()) type_error("distance() requires floating-point operands");
                return emit_reduce(s, HNB_OP_DISTANCE, ValueType(ScalarType::Float), l, &r, base);
            case BinaryOperator::Vec2: {
                if (!l.type.is_scalar() || !r.type.is_scalar() || l.type.elem != r.type.elem) type_error("vec2() requires two scalars of the same type");
                Loc out = new_loc(s, ValueType(l.type.elem, 2));
                emit_mov(s, out.reg, l, 1);
                emit_mov(s, out.reg + 1u, r, 1);
                return out;
            }
            case BinaryOperator::Vec4XyzW: {
                if (l.type.count != 3 || !r.type.is_scalar() || l.type.elem != r.type.elem) type_error("vec4(xyz, w) requires a vec3 and a scalar of the same type");
                Loc out = new_loc(s, ValueType(l.type.elem, 4));
                emit_mov(s, out.reg, l, 3);
                emit_mov(s, out.reg + 3u, r, 1);
                return out;
            }
            default: break;
        }
        throw ShaderGenerateError("unhandled binary operator");
    }

    // rand_uniform_T / rand_normal_T: both operands need a statically known, equal float
    // type in the reference (expr.rs:1162-1190).
    Loc eval_rand_binary(Writer& w, const Expr& e, StreamId s) {
        const bool persistent = persistent_result_;
        persistent_result_ = false;  // operands are ordinary evaluations
        const Loc l = eval(w, e.a);
        const Loc r = eval(w, e.b);
        ValueType lt, rt;
        if (!mod_.try_get(e.a).value_type(&lt) || !mod_.try_get(e.b).value_type(&rt)) type_error("Can't determine the type of the operand");
        if (lt != rt) type_error("Mismatched types");
        if (!lt.is_float()) type_error("Unsupported type");
        persistent_result_ = persistent;
        Loc out = rand_loc(lt);
        emit(s, e.binary == BinaryOperator::UniformRand ? HNB_OP_RANDU : HNB_OP_RANDN, out.reg, fld(s, l), fld(s, r), fld(s, l), lt.count, false, false,
             true);
        return out;
    }

    Loc eval_ternary(Writer& w, const Expr& e, StreamId s) {
        const int base = (int)vtop_;
        Loc x = eval_abs(w, e.a);
        Loc y = eval_abs(w, e.b);
        Loc z = eval_abs(w, e.c);
        if (x.abs || y.abs || z.abs) {
            if (e.ternary == TernaryOperator::Mix || e.ternary == TernaryOperator::SmoothStep) {
                x = conv(x, ScalarType::Float); y = conv(y, ScalarType::Float); z = conv(z, ScalarType::Float);
            } else {
                unify({&x, &y, &z});  // clamp, vec3
            }
        }
        switch (e.ternary) {
            case TernaryOperator::Mix:
                same_type(x, y, "mix()");
                if (!x.type.is_float()) type_error("mix() requires floating-point operands");
                if (!(z.type == x.type || z.type == ValueType(ScalarType::Float))) type_error("mix() fraction must be f32 or match the operands");
                return emit_elementwise(s, HNB_OP_FMIX, x.type, &x, &y, &z, base);
            case TernaryOperator::Clamp:
                same_type(x, y, "clamp()");
                same_type(x, z, "clamp()");
                return emit_elementwise(s, pick(x.type.elem, HNB_OP_FCLAMP, HNB_OP_ICLAMP, HNB_OP_UCLAMP, "clamp"), x.type, &x, &y, &z, base);
            case TernaryOperator::SmoothStep:
                same_type(x, y, "smoothstep()");
                same_type(x, z, "smoothstep()");
                if (!x.type.is_float()) type_error("smoothstep() requires floating-point operands");
                return emit_elementwise(s, HNB_OP_FSMOOTH, x.type, &x, &y, &z, base);
            case TernaryOperator::Vec3: {
                if (!x.type.is_scalar() || x.type != y.type || x.type != z.type) type_error("vec3() requires three scalars of the same type");
                Loc out = new_loc(s, ValueType(x.type.elem, 3));
                emit_mov(s, out.reg, x, 1);
                emit_mov(s, out.reg + 1u, y, 1);
                emit_mov(s, out.reg + 2u, z, 1);
                return out;
            }
        }
        throw ShaderGenerateError("unhandled ternary operator");
    }

    Loc eval_cast(Writer& w, const Expr& e, StreamId s) {
        const int base = (int)vtop_;
        const Loc x = eval(w, e.a);
        const ValueType to = e.rand_type;
        if (to.is_scalar() && !x.type.is_scalar()) type_error("cannot cast " + x.type.to_string() + " to " + to.to_string());
        if (to.is_vector() && x.type.is_vector() && to.count != x.type.count) type_error("cannot cast " + x.type.to_string() + " to " + to.to_string());
        uint32_t op = HNB_OP_MOV;
        const ScalarType f = x.type.elem, t = to.elem;
        if (f != t) {
            if (t == ScalarType::Float) op = f == ScalarType::Int ? HNB_OP_I2F : (f == ScalarType::Uint ? HNB_OP_U2F : HNB_OP_B2F);
            else if (t == ScalarType::Bool) op = f == ScalarType::Float ? HNB_OP_F2B : HNB_OP_I2B;
            else if (f == ScalarType::Float) op = t == ScalarType::Int ? HNB_OP_F2I : HNB_OP_F2U;
            else op = HNB_OP_MOV;  // i32 <-> u32 bit reinterpretation; bool -> int is 0/1
        }
        return emit_elementwise(s, op, to, &x, nullptr, nullptr, base);
    }

    // ---- modifiers -------------------------------------------------------------------------------------
    struct Mark { uint32_t vtop, ptop; };
    Mark mark() const { return Mark{vtop_, ptop_}; }
    void release_temps(const Mark& m) { vtop_ = m.vtop; }
    void release_all(const Mark& m) { vtop_ = m.vtop; ptop_ = m.ptop; }

    void need(StreamId s, Attribute a, const char* who) {
        if (!has(a)) throw ShaderGenerateError(std::string(who) + " requires the " + a.name() + " attribute");
        (void)s;
    }
    Loc want(const Loc& l, ValueType t, const char* what) {
        if (l.type != t) type_error(std::string(what) + " must be " + t.to_string() + ", got " + l.type.to_string());
        return l;
    }
    // gather operands into one block of consecutive registers of one space
    Loc make_block(StreamId s, const std::vector<Loc>& parts) {
        bool all_uniform = true;
        uint32_t n = 0;
        for (const Loc& p : parts) { all_uniform = all_uniform && p.uniform; n += p.type.count; }
        const StreamId bs = all_uniform ? StreamId::Uniform : s;
        Loc blk = new_loc(bs, ValueType(ScalarType::Float, 1));
        if (n > 1) { if (all_uniform) alloc_u(n - 1); else alloc_v(n - 1); }
        uint32_t off = 0;
        for (const Loc& p : parts) {
            emit_mov(bs, blk.reg + off, p, p.type.count);
            off += p.type.count;
        }
        return blk;
    }
    Loc mul_dt(StreamId s, const Loc& x) {  // (x) * sim_params.delta_time
        const Loc dt = delta_time_loc();
        const StreamId ms = x.uniform ? StreamId::Uniform : s;
        if (!x.type.is_float()) type_error("acceleration must be a floating-point value, got " + x.type.to_string());
        return emit_elementwise(ms, HNB_OP_FMUL, x.type, &x, &dt, nullptr);
    }
    // `position - <origin>` where the reference pastes the origin expression without
    // parentheses (accel.rs:176, 291): an infix +/- origin `(l) op (r)` parses as (position - l) op r.
    bool origin_is_infix_addsub(ExprHandle h) const {
        const Expr& e = mod_.try_get(h);
        return e.kind == Expr::Kind::Binary && (e.binary == BinaryOperator::Add || e.binary == BinaryOperator::Sub);
    }
    Loc position_minus_unparenthesized(Writer& w, StreamId s, ExprHandle origin) {
        Loc pos;
        pos.type = VectorType::VEC3F;
        pos.reg = HNB_REG_POSITION;
        const Expr& e = mod_.try_get(origin);
        const Loc l = eval_inline(w, e.a, ScalarType::Float);
        const Loc r = eval_inline(w, e.b, ScalarType::Float);
        const Loc t = emit_elementwise(s, HNB_OP_FSUB, arith_type(pos, l, "-"), &pos, &l, nullptr);
        return emit_elementwise(s, e.binary == BinaryOperator::Add ? HNB_OP_FADD : HNB_OP_FSUB, arith_type(t, r, "+"), &t, &r, nullptr);
    }

    void lower_modifier(Writer& main, StreamId s, const Modifier& m) {
        const Mark mk = mark();
        const bool init = s == StreamId::Init;
        switch (m.kind) {
            case Modifier::Kind::SetAttribute: {
                // attr.rs:92-115
                const Expr& ve = mod_.try_get(m.e[0]);
                ValueType known;
                if (ve.value_type(&known) && known != m.attribute.value_type())
                    type_error(std::string("Mismatching expression type in SetAttributeModifer: attribute '") + upper(m.attribute.name()) +
                               "' requires an expression producing a value of type " + m.attribute.value_type().to_string() + ", but a value of type " +
                               known.to_string() + " was produced instead");
                Loc v = eval_abs(main, m.e[0]);  // `particle.A = <expr>;`: an abstract constant takes the attribute's type
                if (v.abs) v = m.attribute.value_type().is_scalar() ? conv(v, m.attribute.value_type().elem) : conc(v);
                if (v.type != m.attribute.value_type())
                    type_error(std::string("cannot assign a value of type ") + v.type.to_string() + " to attribute '" + upper(m.attribute.name()) + "' of type " +
                               m.attribute.value_type().to_string());
                const AttrSlot& sl = slot(m.attribute);
                touch(s, m.attribute, true);
                if (sl.reg != HNB_REG_NONE) emit(s, HNB_OP_M_PIN_SET, sl.reg, opnd(v), opnd(v), opnd(v), sl.ncomp, false, true, true);
                else if (!(s == StreamId::Update && (m.attribute == Attribute::PREV || m.attribute == Attribute::NEXT)))  // never written back by update
                    emit(s, HNB_OP_STA, 0, opnd(v), opnd(v), opnd(v), sl.ncomp, sl.ncomp == 1, true, true, (uint32_t)attr_index_[m.attribute.id]);
            } break;
            case Modifier::Kind::InheritAttribute: {
                // attr.rs:173-186: particle.A = parent_particle.A;
                const AttrSlot& sl = slot(m.attribute);
                touch(s, m.attribute, true);
                Loc v;
                v.type = m.attribute.value_type();
                v.reg = (uint8_t)alloc_v(sl.ncomp);
                emit(s, HNB_OP_LDPARENT, v.reg, 0, 0, 0, sl.ncomp, true, true, true, (uint32_t)m.attribute.id);
                note_parent_attr(m.attribute);
                if (sl.reg != HNB_REG_NONE) emit(s, HNB_OP_M_PIN_SET, sl.reg, opnd(v), opnd(v), opnd(v), sl.ncomp, false, true, true);
                else emit(s, HNB_OP_STA, 0, opnd(v), opnd(v), opnd(v), sl.ncomp, sl.ncomp == 1, true, true, (uint32_t)attr_index_[m.attribute.id]);
            } break;
            case Modifier::Kind::SetPositionCircle: {
                need(s, Attribute::POSITION, "SetPositionCircleModifier");
                Writer fn{s, {}};
                const Loc c = want(eval(fn, m.e[0]), VectorType::VEC3F, "circle center");
                const Loc n = want(eval(fn, m.e[1]), VectorType::VEC3F, "circle axis");
                // Surface: `let r = {radius};`, Volume: `let r = sqrt(frand()) * ({radius});` (position.rs:68-78)
                const Loc r = want(m.dimension == ShapeDimension::Volume ? eval_inline(fn, m.e[2], ScalarType::Float) : eval(fn, m.e[2]),
                                   ValueType(ScalarType::Float), "circle radius");
                const Loc blk = make_block(s, {c, n, r});
                touch(s, Attribute::POSITION, true);
                emit(s, HNB_OP_M_POS_CIRCLE, 0, opnd(blk), opnd(blk), opnd(blk), 1, true, true, true, m.dimension == ShapeDimension::Volume ? 1u : 0u);
            } break;
            case Modifier::Kind::SetPositionSphere: {
                need(s, Attribute::POSITION, "SetPositionSphereModifier");
                Writer fn{s, {}};
                const Loc c = want(eval(fn, m.e[0]), VectorType::VEC3F, "sphere center");
                const Loc r = want(m.dimension == ShapeDimension::Volume ? eval_inline(fn, m.e[1], ScalarType::Float) : eval(fn, m.e[1]),
                                   ValueType(ScalarType::Float), "sphere radius");  // position.rs:167-181
                const Loc blk = make_block(s, {c, r});
                touch(s, Attribute::POSITION, true);
                emit(s, HNB_OP_M_POS_SPHERE, 0, opnd(blk), opnd(blk), opnd(blk), 1, true, true, true, m.dimension == ShapeDimension::Volume ? 1u : 0u);
            } break;
            case Modifier::Kind::SetPositionCone3d: {
                need(s, Attribute::POSITION, "SetPositionCone3dModifier");
                if (!init) throw ShaderGenerateError("SetPositionCone3dModifier uses the emitter transform, which is only defined in the init context");
                Writer fn{s, {}};
                const Loc h = want(eval(fn, m.e[0]), ValueType(ScalarType::Float), "cone height");          // height
                const Loc rt = want(eval(fn, m.e[2]), ValueType(ScalarType::Float), "cone top radius");     // top_radius
                const Loc rb = want(eval(fn, m.e[1]), ValueType(ScalarType::Float), "cone base radius");    // base_radius
                const Loc blk = make_block(s, {h, rt, rb});
                emit(s, HNB_OP_M_POS_CONE3D, 0, opnd(blk), opnd(blk), opnd(blk), 1, true, true, true);
            } break;
            case Modifier::Kind::SetVelocityCircle:
            case Modifier::Kind::SetVelocityTangent: {
                const char* who = m.kind == Modifier::Kind::SetVelocityCircle ? "SetVelocityCircleModifier" : "SetVelocityTangentModifier";
                need(s, Attribute::POSITION, who);
                need(s, Attribute::VELOCITY, who);
                if (!init) throw ShaderGenerateError(std::string(who) + " uses the emitter transform, which is only defined in the init context");
                Writer fn{s, {}};
                const Loc c = want(eval(fn, m.e[0]), VectorType::VEC3F, "center/origin");
                const Loc ax = want(eval(fn, m.e[1]), VectorType::VEC3F, "axis");
                const Loc sp = want(eval_inline(fn, m.e[2], ScalarType::Float), ValueType(ScalarType::Float), "speed");
                const Loc blk = make_block(s, {c, ax, sp});
                emit(s, m.kind == Modifier::Kind::SetVelocityCircle ? HNB_OP_M_VEL_CIRCLE : HNB_OP_M_VEL_TANGENT, 0, opnd(blk), opnd(blk), opnd(blk), 1,
                     true, true, true);
            } break;
            case Modifier::Kind::SetVelocitySphere: {
                need(s, Attribute::POSITION, "SetVelocitySphereModifier");
                need(s, Attribute::VELOCITY, "SetVelocitySphereModifier");
                const Loc c = want(eval(main, m.e[0]), VectorType::VEC3F, "center");
                const Loc sp = want(eval_inline(main, m.e[1], ScalarType::Float), ValueType(ScalarType::Float), "speed");
                touch(s, Attribute::POSITION, false);
                touch(s, Attribute::VELOCITY, true);
                emit(s, HNB_OP_M_VEL_SPHERE, 0, opnd(c), opnd(sp), opnd(c), 1, true, true, true);
            } break;
            case Modifier::Kind::Accel: {
                // accel.rs:79-86: `velocity += (<accel>) * delta_time;`
                need(s, Attribute::VELOCITY, "AccelModifier");
                const Loc a = eval_inline(main, m.e[0], ScalarType::Float);
                Loc t = mul_dt(s, a);
                if (t.type == ValueType(ScalarType::Float)) {  // vec3 += f32: scalar broadcast
                    Loc v = new_loc(t.uniform ? StreamId::Uniform : s, VectorType::VEC3F);
                    emit_mov(t.uniform ? StreamId::Uniform : s, v.reg, t, 3, true);
                    t = v;
                }
                want(t, VectorType::VEC3F, "acceleration");
                touch(s, Attribute::VELOCITY, true);
                emit(s, HNB_OP_M_VEL_ADD, 0, opnd(t), opnd(t), opnd(t), 1, true, true, true);
            } break;
            case Modifier::Kind::RadialAccel:
            case Modifier::Kind::TangentAccel: {
                const bool radial = m.kind == Modifier::Kind::RadialAccel;
                const char* who = radial ? "RadialAccelModifier" : "TangentAccelModifier";
                need(s, Attribute::POSITION, who);
                need(s, Attribute::VELOCITY, who);
                Writer fn{s, {}};
                Writer& wr = radial ? fn : main;  // RadialAccel evaluates inside make_fn, TangentAccel in the main writer
                touch(s, Attribute::POSITION, false);
                touch(s, Attribute::VELOCITY, true);
                const ExprHandle origin_h = m.e[0];
                if (origin_is_infix_addsub(origin_h)) {
                    // generic path reproducing the reference's operator precedence quirk
                    const Loc d = position_minus_unparenthesized(wr, s, origin_h);
                    want(d, VectorType::VEC3F, "position - origin");
                    Loc dir = new_loc(s, VectorType::VEC3F);
                    emit(s, HNB_OP_NORMALIZE, dir.reg, opnd(d), opnd(d), opnd(d), 3, false, true, true);
                    if (!radial) {
                        const Loc ax = want(eval(wr, m.e[1]), VectorType::VEC3F, "axis");
                        Loc cr = new_loc(s, VectorType::VEC3F);
                        emit(s, HNB_OP_CROSS, cr.reg, opnd(ax), opnd(dir), opnd(ax), 3, false, false, true);
                        Loc tn = new_loc(s, VectorType::VEC3F);
                        emit(s, HNB_OP_NORMALIZE, tn.reg, opnd(cr), opnd(cr), opnd(cr), 3, false, true, true);
                        dir = tn;
                    }
                    const Loc acc = want(eval_inline(wr, m.e[radial ? 1 : 2], ScalarType::Float), ValueType(ScalarType::Float), "acceleration");
                    const Loc sdt = mul_dt(s, acc);
                    const Loc dv = emit_elementwise(s, HNB_OP_FMUL, VectorType::VEC3F, &dir, &sdt, nullptr);
                    emit(s, HNB_OP_M_VEL_ADD, 0, opnd(dv), opnd(dv), opnd(dv), 1, true, true, true);
                } else if (radial) {
                    const Loc o = want(eval(wr, m.e[0]), VectorType::VEC3F, "origin");
                    const Loc acc = want(eval_inline(wr, m.e[1], ScalarType::Float), ValueType(ScalarType::Float), "acceleration");
                    const Loc sdt = mul_dt(s, acc);
                    emit(s, HNB_OP_M_RADIAL_ACCEL, 0, opnd(o), opnd(sdt), opnd(o), 1, true, true, true);
                } else {
                    const Loc o = want(eval(wr, m.e[0]), VectorType::VEC3F, "origin");
                    const Loc ax = want(eval(wr, m.e[1]), VectorType::VEC3F, "axis");
                    const Loc acc = want(eval_inline(wr, m.e[2], ScalarType::Float), ValueType(ScalarType::Float), "acceleration");
                    const Loc sdt = mul_dt(s, acc);
                    emit(s, HNB_OP_M_TANGENT_ACCEL, 0, opnd(o), opnd(ax), opnd(sdt), 1, true, true, true);
                }
            } break;
            case Modifier::Kind::LinearDrag: {
                // force.rs:284-297: velocity *= max(0., (1.) - ((drag) * (delta_time)))
                need(s, Attribute::VELOCITY, "LinearDragModifier");
                const Loc drag = eval_inline(main, m.e[0], ScalarType::Float);
                const Loc dt = delta_time_loc();
                const StreamId ms = drag.uniform ? StreamId::Uniform : s;
                const Loc drag_dt = emit_elementwise(ms, HNB_OP_FMUL, arith_type(drag, dt, "*"), &drag, &dt, nullptr);
                if (!drag_dt.type.is_float()) type_error("drag must be a floating-point value");
                const Loc one = lit_f32(1.0f), zero = lit_f32(0.0f);
                const Loc omd = emit_elementwise(ms, HNB_OP_FSUB, arith_type(one, drag_dt, "-"), &one, &drag_dt, nullptr);
                same_type(zero, omd, "max()");
                const Loc f = emit_elementwise(ms, HNB_OP_FMAX, omd.type, &zero, &omd, nullptr);
                touch(s, Attribute::VELOCITY, true);
                emit(s, HNB_OP_M_VEL_SCALE, 0, opnd(f), opnd(f), opnd(f), 1, true, true, true);
            } break;
            case Modifier::Kind::ConformToSphere: {
                need(s, Attribute::POSITION, "ConformToSphereModifier");
                need(s, Attribute::VELOCITY, "ConformToSphereModifier");
                Writer fn{s, {}};
                const ValueType F(ScalarType::Float);
                // evaluation order of force.rs:186-194
                const Loc origin = want(eval(fn, m.e[0]), VectorType::VEC3F, "origin");
                const Loc radius = want(eval(fn, m.e[1]), F, "radius");
                const Loc infl = want(eval(fn, m.e[2]), F, "influence_dist");
                const Loc shell = m.has_shell ? want(eval(fn, m.e[5]), F, "shell_half_thickness") : lit_f32(0.1f);
                const Loc maxs = want(eval(fn, m.e[4]), F, "max_attraction_speed");
                const Loc acc = want(eval(fn, m.e[3]), F, "attraction_accel");
                const Loc sticky = m.has_sticky ? want(eval_inline(fn, m.e[6], ScalarType::Float), F, "sticky_factor") : lit_f32(2.0f);
                const Loc blk = make_block(s, {origin, radius, infl, shell, maxs, acc, sticky});
                const Loc dt = delta_time_loc();
                touch(s, Attribute::POSITION, false);
                touch(s, Attribute::VELOCITY, true);
                emit(s, HNB_OP_M_CONFORM_SPHERE, 0, opnd(blk), opnd(dt), opnd(blk), 1, true, true, true);
            } break;
            case Modifier::Kind::KillSphere: {
                // kill.rs:76-96: dot(pos - center, pos - center) </> sqr_radius
                need(s, Attribute::POSITION, "KillSphereModifier");
                const Loc c = want(eval(main, m.e[0]), VectorType::VEC3F, "center");
                const Loc r2 = want(eval_inline(main, m.e[1], ScalarType::Float), ValueType(ScalarType::Float), "sqr_radius");
                touch(s, Attribute::POSITION, false);
                emit(s, HNB_OP_M_KILL_SPHERE, 0, opnd(c), opnd(r2), opnd(c), 1, true, true, true, m.kill_inside ? 1u : 0u);
            } break;
            case Modifier::Kind::KillAabb: {
                // kill.rs:156-181: any(abs(pos - center) > half_size) / all(... < half_size)
                need(s, Attribute::POSITION, "KillAabbModifier");
                const Loc c = want(eval(main, m.e[0]), VectorType::VEC3F, "center");
                const Loc hs = want(eval(main, m.e[1]), VectorType::VEC3F, "half_size");
                touch(s, Attribute::POSITION, false);
                emit(s, HNB_OP_M_KILL_AABB, 0, opnd(c), opnd(hs), opnd(c), 1, true, true, true, m.kill_inside ? 1u : 0u);
            } break;
            case Modifier::Kind::EmitSpawnEvent: {
                // modifier/mod.rs:669-695: `let count = <expr>;` is evaluated unconditionally, then
                // `if (is_alive)` / `if (was_alive && !is_alive)` append_spawn_events_<channel>(…, particle_index, count)
                const Loc cnt = eval(main, m.e[0]);
                if (cnt.type != ValueType(ScalarType::Uint))
                    type_error("EmitSpawnEventModifier::count must be an expression of type u32, got " + cnt.type.to_string());
                if (m.child_index >= HNB_MAX_EVENT_CHANNELS)
                    throw ShaderGenerateError("EmitSpawnEventModifier::child_index " + std::to_string(m.child_index) + " exceeds the supported event channels");
                emit(s, HNB_OP_M_EMIT_EVENTS, 0, opnd(cnt), opnd(cnt), opnd(cnt), 1, true, true, true,
                     m.child_index | (m.condition == EventEmitCondition::OnDie ? 0x100u : 0u));
                n_event_channels_ = std::max<uint32_t>(n_event_channels_, m.child_index + 1u);
            } break;
            case Modifier::Kind::Render: break;
        }
        // statement temporaries die here; hoisted rand values of the main writer stay alive
        release_temps(mk);
        // function writers' hoisted values die with the function: they were allocated below
        // ptop_ after `mk` and are not referenced by the main writer's memo.
        bool main_has_new = false;
        for (const auto& kv : main.memo) if (!kv.second.uniform && kv.second.reg < mk.ptop) main_has_new = true;
        if (!main_has_new) ptop_ = mk.ptop;
        (void)init;
    }

    static std::string upper(const char* s) {
        std::string r(s);
        for (char& c : r) c = (char)std::toupper((unsigned char)c);
        return r;
    }

    void lower_stream(StreamId s) {
        vtop_ = attr_end_;
        ptop_ = vlimit_;
        vmax_ = attr_end_;
        Writer main{s, {}};
        if (s == StreamId::Init) {
            for (const Modifier& m : asset_.init_modifiers()) lower_modifier(main, s, m);
            // PREV / NEXT = 0xffffffff (vfx_init.wgsl:176-181)
            for (Attribute a : {Attribute::PREV, Attribute::NEXT})
                if (has(a)) {
                    const Loc ones = load_literal(Value(0xffffffffu));
                    emit(s, HNB_OP_STA, 0, opnd(ones), opnd(ones), opnd(ones), 1, true, true, true, (uint32_t)attr_index_[a.id]);
                }
            // SimulationSpace::Global: particle.position += transform[3].xyz (lib.rs:518-531)
            if (asset_.simulation_space == SimulationSpace::Global) emit(s, HNB_OP_M_ADD_XLATE, 0, 0, 0, 0, 1, true, true, true);
            init_regs_ = vmax_;
        } else {
            const bool has_age = has(Attribute::AGE), has_life = has(Attribute::LIFETIME);
            const bool euler = asset_.motion_integration != MotionIntegration::None && has(Attribute::POSITION) && has(Attribute::VELOCITY);
            if (has_age) {
                if (!has_life)
                    throw ShaderGenerateError("the particle layout has AGE but no LIFETIME: the reference emits an update shader without `is_alive` "
                                              "(src/lib.rs:1223-1247), which fails to compile");
                const Loc dt = delta_time_loc();
                touch(s, Attribute::AGE, true);
                touch(s, Attribute::LIFETIME, false);
                emit(s, HNB_OP_M_AGE_TICK, 0, opnd(dt), opnd(dt), opnd(dt), 1, true, true, true, 1u);
            }
            auto emit_euler = [&] {
                const Loc dt = delta_time_loc();
                touch(s, Attribute::POSITION, true);
                touch(s, Attribute::VELOCITY, false);
                emit(s, HNB_OP_M_EULER, 0, opnd(dt), opnd(dt), opnd(dt), 1, true, true, true);
            };
            if (euler && asset_.motion_integration == MotionIntegration::PreUpdate) emit_euler();
            for (const Modifier& m : asset_.update_modifiers()) lower_modifier(main, s, m);
            if (euler && asset_.motion_integration == MotionIntegration::PostUpdate) emit_euler();
            update_regs_ = vmax_;
        }
    }

    uint32_t init_regs_ = 0, update_regs_ = 0;
    bool persistent_result_ = false;  // next rand result goes to a persistent register
    Loc rand_loc(ValueType t) {
        Loc l;
        l.type = t;
        l.reg = (uint8_t)(persistent_result_ ? alloc_persistent(t.count) : alloc_v(t.count));
        persistent_result_ = false;
        return l;
    }
};

std::vector<uint8_t> Lowerer::run() {
    // ---- validation (lib.rs:823-856) ---------------------------------------------------------
    const std::vector<Attribute> layout = asset_.particle_layout();
    std::vector<Attribute> stored;
    for (Attribute a : layout)
        if (!a.is_pseudo()) stored.push_back(a);
    if (stored.empty()) throw ShaderGenerateError("Asset " + asset_.name + " has invalid empty particle layout.");
    bool has_pos = false, has_ribbon = false, has_age = false;
    for (Attribute a : stored) {
        has_pos = has_pos || a == Attribute::POSITION;
        has_ribbon = has_ribbon || a == Attribute::RIBBON_ID;
        has_age = has_age || a == Attribute::AGE;
    }
    if (!has_pos)
        throw ShaderGenerateError("The particle layout of asset '" + asset_.name +
                                  "' is missing the 'POSITION' attribute. Add a modifier using that attribute, for example the SetAttributeModifier.");
    if (has_ribbon && !has_age)
        throw ShaderGenerateError("The particle layout of asset '" + asset_.name +
                                  "' uses ribbons (has the 'RIBBON_ID' attribute), but is missing the 'AGE' attribute, which is mandatory for ribbons. Add a "
                                  "modifier using that attribute, for example the SetAttributeModifier.");
    if (asset_.capacity() == 0) throw ShaderGenerateError("Asset " + asset_.name + " has zero capacity.");

    // ---- attribute table / V register assignment ------------------------------------------------
    for (int& i : attr_index_) i = -1;
    uint32_t next = HNB_REG_FIRST_FREE;
    for (Attribute a : stored) {
        AttrSlot sl;
        sl.attr = a;
        sl.ncomp = a.value_type().count;
        if (a == Attribute::POSITION) sl.reg = HNB_REG_POSITION;
        else if (a == Attribute::VELOCITY) sl.reg = HNB_REG_VELOCITY;
        else if (a == Attribute::AGE) sl.reg = HNB_REG_AGE;
        else if (a == Attribute::LIFETIME) sl.reg = HNB_REG_LIFETIME;
        else sl.reg = HNB_REG_NONE;  // memory operand (HNB_OP_LDA / HNB_OP_STA)
        if (attrs_.size() >= 40) throw ShaderGenerateError("the particle layout has too many attributes");
        attr_index_[a.id] = (int)attrs_.size();
        attrs_.push_back(sl);
    }
    attr_end_ = next;

    // ---- properties ----------------------------------------------------------------------------------
    for (const Property& p : mod_.properties()) {
        prop_offset_.push_back(prop_words_);
        prop_words_ += p.default_value.type.count;
    }

    lower_stream(StreamId::Init);
    lower_stream(StreamId::Update);

    // ---- serialise --------------------------------------------------------------------------------------
    HnbProgramHeader h;
    std::memset(&h, 0, sizeof h);
    h.magic = HNB_PROGRAM_MAGIC;
    h.version = HNB_PROGRAM_VERSION;
    h.capacity = asset_.capacity();
    h.flags = (asset_.simulation_space == SimulationSpace::Global ? HNB_PROG_GLOBAL_SPACE : 0u) | (has_ribbon ? HNB_PROG_HAS_RIBBONS : 0u) |
              (parent_attrs_.empty() ? 0u : HNB_PROG_READS_PARENT) | (n_event_channels_ ? HNB_PROG_EMITS_EVENTS : 0u);
    h.n_event_channels = n_event_channels_;
    {   // what the render modifiers read every frame (their declared attribute requirements: src/modifier/output.rs impl_mod_render!)
        uint64_t mask = 0;
        for (const Modifier& m : asset_.render_modifiers())
            for (const Attribute& a : m.attributes()) if (a.id < 64u) mask |= 1ull << a.id;
        h.render_reads_lo = (uint32_t)mask; h.render_reads_hi = (uint32_t)(mask >> 32);
    }
    h.parent_n_attrs = (uint32_t)parent_attrs_.size();
    h.n_attrs = (uint32_t)attrs_.size();
    h.n_props = (uint32_t)mod_.properties().size();
    h.prop_words = prop_words_;
    h.uniform_len = (uint32_t)uni_.code.size();
    h.init_len = (uint32_t)init_.code.size();
    h.update_len = (uint32_t)upd_.code.size();
    h.n_uregs = utop_;
    h.init_regs = std::max<uint32_t>(init_regs_, attr_end_);
    h.update_regs = std::max<uint32_t>(update_regs_, attr_end_);
    uint32_t off = sizeof(HnbProgramHeader);
    h.attrs_off = off; off += h.n_attrs * (uint32_t)sizeof(HnbAttrEntry);
    h.props_off = off; off += h.n_props * (uint32_t)sizeof(HnbPropEntry);
    h.parent_attrs_off = off; off += h.parent_n_attrs * 4u;
    off = (off + 7u) & ~7u;
    h.uniform_off = off; off += h.uniform_len * 8u;
    h.init_off = off; off += h.init_len * 8u;
    h.update_off = off; off += h.update_len * 8u;
    h.total_size = off;
    std::vector<uint8_t> blob(off, 0);
    std::memcpy(blob.data(), &h, sizeof h);
    for (size_t i = 0; i < attrs_.size(); ++i) {
        HnbAttrEntry e;
        std::memset(&e, 0, sizeof e);
        e.attr = (uint16_t)attrs_[i].attr.id;
        e.ncomp = attrs_[i].ncomp;
        e.reg = attrs_[i].reg;
        e.scalar_type = (uint8_t)attrs_[i].attr.value_type().elem;
        e.update_flags = attrs_[i].upd_flags;
        std::memcpy(blob.data() + h.attrs_off + i * sizeof e, &e, sizeof e);
    }
    for (size_t i = 0; i < mod_.properties().size(); ++i) {
        const Property& p = mod_.properties()[i];
        HnbPropEntry e;
        std::memset(&e, 0, sizeof e);
        if (p.name.size() >= sizeof e.name) throw ShaderGenerateError("property name '" + p.name + "' is too long (max 47 bytes)");
        std::memcpy(e.name, p.name.c_str(), p.name.size());
        e.scalar_type = (uint8_t)p.default_value.type.elem;
        e.ncomp = p.default_value.type.count;
        e.word_offset = (uint16_t)prop_offset_[i];
        for (int c = 0; c < 4; ++c) e.default_bits[c] = p.default_value.bits[c];
        std::memcpy(blob.data() + h.props_off + i * sizeof e, &e, sizeof e);
    }
    if (h.parent_n_attrs) std::memcpy(blob.data() + h.parent_attrs_off, parent_attrs_.data(), (size_t)h.parent_n_attrs * 4);
    if (h.uniform_len) std::memcpy(blob.data() + h.uniform_off, uni_.code.data(), (size_t)h.uniform_len * 8);
    if (h.init_len) std::memcpy(blob.data() + h.init_off, init_.code.data(), (size_t)h.init_len * 8);
    if (h.update_len) std::memcpy(blob.data() + h.update_off, upd_.code.data(), (size_t)h.update_len * 8);
    return blob;
}

const char* op_name(uint32_t op) {
    static const char* names[] = {
        "NOP", "LOADK", "LDB", "LDP", "LDID", "LDPC", "LDALIVE", "LDPARENT", "LDA", "STA", "MOV",
        "FABS", "FCEIL", "FFLOOR", "FROUND", "FFRACT", "FSQRT", "FRSQ", "FSIGN", "FSAT", "FSIN", "FCOS", "FTAN", "FASIN", "FACOS", "FATAN", "FEXP",
        "FEXP2", "FLOG", "FLOG2",
        "FADD", "FSUB", "FMUL", "FDIV", "FREM", "FMIN", "FMAX", "FSTEP", "FATAN2", "FPOW",
        "FMIX", "FCLAMP", "FSMOOTH", "FLT", "FLE", "FGT", "FGE",
        "IADD", "ISUB", "IMUL", "IDIV", "IREM", "IMIN", "IMAX", "IABS", "ISIGN", "ICLAMP", "ILT", "ILE", "IGT", "IGE",
        "UDIV", "UREM", "UMIN", "UMAX", "UCLAMP", "ULT", "ULE", "UGT", "UGE",
        "F2I", "F2U", "I2F", "U2F", "B2F", "F2B", "I2B",
        "ALL", "ANY", "DOT", "LENGTH", "DISTANCE", "NORMALIZE", "CROSS", "PACK4UNORM", "PACK4SNORM", "UNPACK4UNORM", "UNPACK4SNORM",
        "FRAND", "RANDU", "RANDN", "ALIVE_SET", "ALIVE_AND", "KILL_IF",
        "M_AGE_TICK", "M_EULER", "M_VEL_SCALE", "M_VEL_ADD", "M_PIN_SET", "M_RADIAL_ACCEL", "M_TANGENT_ACCEL", "M_CONFORM_SPHERE", "M_KILL_SPHERE",
        "M_KILL_AABB", "M_VEL_SPHERE", "M_POS_CIRCLE", "M_POS_SPHERE", "M_POS_CONE3D", "M_VEL_CIRCLE", "M_VEL_TANGENT", "M_ADD_XLATE", "M_EMIT_EVENTS"};
    static_assert(sizeof(names) / sizeof(names[0]) == HNB_OP_COUNT, "op name table out of sync with HnbOp");
    return op < HNB_OP_COUNT ? names[op] : "?";
}

}  // namespace

std::vector<uint8_t> lower(const EffectAsset& asset) {
    // Programs that fit the 32-register file keep it (the interpreter kernels hold it in VGPRs); deeper
    // expression trees are lowered again for the wide file.
    try {
        return Lowerer(asset).run();
    } catch (const RegisterPressure&) {
        return Lowerer(asset, HNB_VM_MAX_REGS_WIDE).run();
    }
}

std::string disassemble(const std::vector<uint8_t>& blob) {
    std::ostringstream os;
    if (blob.size() < sizeof(HnbProgramHeader)) return "<truncated>";
    HnbProgramHeader h;
    std::memcpy(&h, blob.data(), sizeof h);
    os << "capacity " << h.capacity << " flags " << h.flags << " uregs " << h.n_uregs << " init_regs " << h.init_regs << " update_regs " << h.update_regs << "\n";
    for (uint32_t i = 0; i < h.n_attrs; ++i) {
        HnbAttrEntry e;
        std::memcpy(&e, blob.data() + h.attrs_off + i * sizeof e, sizeof e);
        os << "attr " << Attribute((HnbAttr)e.attr).name() << " r" << (int)e.reg << " x" << (int)e.ncomp << " upd=" << (int)e.update_flags << "\n";
    }
    auto reg = [](uint32_t o, bool ustream) {
        char b[16];
        if (ustream || (o & HNB_OPERAND_DECODED_U)) std::snprintf(b, sizeof b, "u%u", o & 0xffu);
        else std::snprintf(b, sizeof b, "r%u", o);
        return std::string(b);
    };
    const char* names[3] = {"uniform", "init", "update"};
    const uint32_t offs[3] = {h.uniform_off, h.init_off, h.update_off}, lens[3] = {h.uniform_len, h.init_len, h.update_len};
    for (int s = 0; s < 3; ++s) {
        os << names[s] << ":\n";
        for (uint32_t i = 0; i < lens[s]; ++i) {
            uint32_t w[2];
            std::memcpy(w, blob.data() + offs[s] + (size_t)i * 8, 8);
            const uint32_t op = w[0] & 0xff, d = (w[0] >> 8) & 0xff;
            uint32_t a = (w[0] >> 16) & 0xff, b = w[0] >> 24, c = w[1] & 0xff;
            if (s != 0) { a = HNB_OPERAND_DECODE(a, w[1] >> 13); b = HNB_OPERAND_DECODE(b, w[1] >> 14); c = HNB_OPERAND_DECODE(c, w[1] >> 15); }
            os << "  " << op_name(op) << " ";
            if (op == HNB_OP_LOADK) { float f; std::memcpy(&f, &w[1], 4); os << reg(d, true) << " = 0x" << std::hex << w[1] << std::dec << " (" << f << ")"; }
            else if (op == HNB_OP_LDB) os << reg(d, true) << " = sim[" << a << "]";
            else if (op == HNB_OP_LDP) os << reg(d, true) << " = prop[" << w[1] << "] x" << ((a & 3) + 1);
            else os << reg(d, s == 0) << ", " << reg(a, s == 0) << ", " << reg(b, s == 0) << ", " << reg(c, s == 0) << " w" << (((w[1] >> 8) & 3) + 1)
                    << " bc" << ((w[1] >> 10) & 7) << " aux" << (w[1] >> 16);
            os << "\n";
        }
    }
    return os.str();
}

}  // namespace hanabi
