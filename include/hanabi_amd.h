/* hanabi_amd.h — C ABI of the MI355X-native particle simulation hot path.
 *
 * This is the drop-in boundary for bevy_hanabi's GPU simulation path. The reference has
 * no FFI for this path; the seam it sits behind is:
 *   (1) `EffectShaderSources::generate()`            src/lib.rs:805        -> hnb_program_create()
 *   (2) effect slab + metadata allocation            src/render/effect_cache.rs:232-356,
 *                                                    src/render/mod.rs:6048-6070 -> hnb_effect_create()
 *   (3) per-frame GpuSimParams / GpuSpawnerParams    src/render/mod.rs:218-243,381-449 -> hnb_frame_begin(),
 *                                                                                          hnb_effect_set_frame()
 *   (4) the `simulate` render-graph system           src/render/mod.rs:6942-7613 -> hnb_simulate()
 * A host crate (Rust in the reference; C++17 `hanabi::` in this repo) lowers an EffectAsset's
 * modifier/expression graph to an HnbProgram blob (bytecode + attribute table) and drives
 * these entry points. Plain pointers and sizes only; no C++/torch types.
 *
 * All functions return HNB_OK (0) or a negative HnbStatus; hnb_last_error() gives text.
 * Thread model: one context per GPU, externally synchronised; hnb_simulate() is
 * asynchronous on the context's stream, everything that reads back synchronises.
 */
#ifndef HANABI_AMD_H
#define HANABI_AMD_H

#ifndef __HIPCC_RTC__  /* runtime-compiled device code gets these types from hiprtc */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------- */
/* Status codes                                                                       */
/* ---------------------------------------------------------------------------------- */
typedef enum HnbStatus {
    HNB_OK = 0,
    HNB_ERR_INVALID_ARG = -1,
    HNB_ERR_BAD_PROGRAM = -2,   /* malformed / unsupported HnbProgram blob */
    HNB_ERR_HIP = -3,           /* a HIP runtime call failed */
    HNB_ERR_NO_DEVICE = -4,     /* no gfx950 device: the product path never falls back to CPU */
    HNB_ERR_OUT_OF_MEMORY = -5,
    HNB_ERR_NOT_FOUND = -6,     /* unknown attribute / property */
    HNB_ERR_DEVICE_FAULT = -7   /* a kernel reported a watchdog / consistency fault */
} HnbStatus;

/* ---------------------------------------------------------------------------------- */
/* Attributes (reference: src/attributes.rs:549-675, ALL list at :1338-1378)           */
/* ---------------------------------------------------------------------------------- */
typedef enum HnbAttr {
    HNB_ATTR_ID = 0, HNB_ATTR_PARTICLE_COUNTER = 1,  /* pseudo attributes, never stored */
    HNB_ATTR_POSITION = 2, HNB_ATTR_VELOCITY = 3, HNB_ATTR_AGE = 4, HNB_ATTR_LIFETIME = 5,
    HNB_ATTR_COLOR = 6, HNB_ATTR_HDR_COLOR = 7, HNB_ATTR_ALPHA = 8,
    HNB_ATTR_SIZE = 9, HNB_ATTR_SIZE2 = 10, HNB_ATTR_SIZE3 = 11,
    HNB_ATTR_PREV = 12, HNB_ATTR_NEXT = 13,
    HNB_ATTR_AXIS_X = 14, HNB_ATTR_AXIS_Y = 15, HNB_ATTR_AXIS_Z = 16,
    HNB_ATTR_SPRITE_INDEX = 17,
    HNB_ATTR_F32_0 = 18, HNB_ATTR_F32_1 = 19, HNB_ATTR_F32_2 = 20, HNB_ATTR_F32_3 = 21,
    HNB_ATTR_F32X2_0 = 22, HNB_ATTR_F32X2_1 = 23, HNB_ATTR_F32X2_2 = 24, HNB_ATTR_F32X2_3 = 25,
    HNB_ATTR_F32X3_0 = 26, HNB_ATTR_F32X3_1 = 27, HNB_ATTR_F32X3_2 = 28, HNB_ATTR_F32X3_3 = 29,
    HNB_ATTR_F32X4_0 = 30, HNB_ATTR_F32X4_1 = 31, HNB_ATTR_F32X4_2 = 32, HNB_ATTR_F32X4_3 = 33,
    HNB_ATTR_U32_0 = 34, HNB_ATTR_U32_1 = 35, HNB_ATTR_U32_2 = 36, HNB_ATTR_U32_3 = 37,
    HNB_ATTR_RIBBON_ID = 38,
    HNB_ATTR_COUNT = 39
} HnbAttr;

typedef enum HnbScalarType { HNB_BOOL = 0, HNB_F32 = 1, HNB_I32 = 2, HNB_U32 = 3 } HnbScalarType;

/* ---------------------------------------------------------------------------------- */
/* Lowered effect program                                                             */
/* What `Modifier::apply()` + `Expr::eval()` emit as WGSL text (src/modifier/mod.rs:154-181,*/
/* src/graph/expr.rs:1121-1258) is lowered by the host into three instruction streams: */
/*   uniform : everything that depends only on literals, properties and sim params.   */
/*             Evaluated ON THE HOST once per instance per frame into a parameter     */
/*             block of "U registers" that the kernels read with scalar loads.        */
/*   init    : per-particle spawn program (INIT_CODE of vfx_init.wgsl).               */
/*   update  : per-particle update program (AGE/REAP/UPDATE_CODE of vfx_update.wgsl). */
/* ---------------------------------------------------------------------------------- */
/* Instruction = two little-endian u32 words:
 *   w0: op[7:0] dst[15:8] a[23:16] b[31:24]
 *   w1: c[7:0] width-1[9:8] bcast_a[10] bcast_b[11] bcast_c[12] ubank_a[13] ubank_b[14] ubank_c[15] aux[31:16]
 *   (HNB_OP_LOADK: w1 is the 32-bit immediate; HNB_OP_LDP: w1 = property word offset,
 *    width-1 in a[1:0]).
 * Operand bytes: bit 7 set = U register (parameter block, index = bits 6:0 | ubank << 7), clear = V
 * register (per-particle, up to 128 x 32-bit). In the uniform stream every operand is a U
 * register and the byte is its index (0..255). A vector operand occupies `width` consecutive
 * registers; an operand with its bcast bit set is a scalar broadcast.                  */
#define HNB_VM_MAX_REGS 32u    /* V registers per particle: the file every kernel holds in VGPRs */
#define HNB_VM_MAX_REGS_WIDE 128u /* largest V file a program may ask for (init_regs / update_regs). Programs above
                                    HNB_VM_MAX_REGS run on the wide file: specialised kernels scalarise it, the
                                    interpreter kernels index it in scratch memory (correct, slower). */
#define HNB_VM_MAX_UREGS 256u  /* U registers per instance */
#define HNB_MAX_EVENT_CHANNELS 8u /* child event channels per parent effect (EmitSpawnEventModifier::child_index < 8; the reference loops over any number: src/lib.rs:964-1002) */
#define HNB_OPERAND_U 0x80u
/* Decoded operand of a varying stream (what the VM works with): bit 8 = U register, bits 7:0 = index. */
#define HNB_OPERAND_DECODED_U 0x100u
#define HNB_OPERAND_DECODE(byte, bank) (((byte) & HNB_OPERAND_U) ? (HNB_OPERAND_DECODED_U | ((byte) & 0x7fu) | (((bank) & 1u) << 7)) : (byte))

typedef enum HnbOp {
    HNB_OP_NOP = 0,
    /* uniform stream only */
    HNB_OP_LOADK,
    HNB_OP_LDB,      /* dst = sim_params field a: 0 time,1 delta_time,2 virtual_time,3 virtual_delta_time,4 real_time,5 real_delta_time */
    HNB_OP_LDP,      /* dst[..w] = properties[w1 ..] */
    /* varying streams only */
    HNB_OP_LDID,     /* dst = particle_index (slot + slot_base) — Attribute::ID, expr.rs:1353-1360 */
    HNB_OP_LDPC,     /* dst = particle_counter — Attribute::PARTICLE_COUNTER, expr.rs:1361-1363 */
    HNB_OP_LDALIVE,  /* dst = is_alive (bool) — BuiltInOperator::IsAlive */
    HNB_OP_LDPARENT, /* dst[..w] = parent_particle.<HnbAttr aux>: init stream of an effect with a parent (vfx_init.wgsl:166-171) */
    HNB_OP_LDA,      /* dst[..w] = particle.<attribute table entry aux> (non-pinned attributes live in memory) */
    HNB_OP_STA,      /* particle.<attribute table entry aux> = r[a..a+w] */
    HNB_OP_MOV,
    /* f32 unary */
    HNB_OP_FABS, HNB_OP_FCEIL, HNB_OP_FFLOOR, HNB_OP_FROUND, HNB_OP_FFRACT, HNB_OP_FSQRT, HNB_OP_FRSQ,
    HNB_OP_FSIGN, HNB_OP_FSAT, HNB_OP_FSIN, HNB_OP_FCOS, HNB_OP_FTAN, HNB_OP_FASIN, HNB_OP_FACOS,
    HNB_OP_FATAN, HNB_OP_FEXP, HNB_OP_FEXP2, HNB_OP_FLOG, HNB_OP_FLOG2,
    /* f32 binary */
    HNB_OP_FADD, HNB_OP_FSUB, HNB_OP_FMUL, HNB_OP_FDIV, HNB_OP_FREM, HNB_OP_FMIN, HNB_OP_FMAX,
    HNB_OP_FSTEP,    /* dst = step(edge=a, x=b) */
    HNB_OP_FATAN2, HNB_OP_FPOW,
    /* f32 ternary */
    HNB_OP_FMIX, HNB_OP_FCLAMP, HNB_OP_FSMOOTH,   /* smoothstep(lo=a, hi=b, x=c) */
    /* f32 compare -> bool */
    HNB_OP_FLT, HNB_OP_FLE, HNB_OP_FGT, HNB_OP_FGE,
    /* i32 */
    HNB_OP_IADD, HNB_OP_ISUB, HNB_OP_IMUL, HNB_OP_IDIV, HNB_OP_IREM, HNB_OP_IMIN, HNB_OP_IMAX,
    HNB_OP_IABS, HNB_OP_ISIGN, HNB_OP_ICLAMP, HNB_OP_ILT, HNB_OP_ILE, HNB_OP_IGT, HNB_OP_IGE,
    /* u32 (add/sub/mul share the i32 opcodes) */
    HNB_OP_UDIV, HNB_OP_UREM, HNB_OP_UMIN, HNB_OP_UMAX, HNB_OP_UCLAMP,
    HNB_OP_ULT, HNB_OP_ULE, HNB_OP_UGT, HNB_OP_UGE,
    /* conversions */
    HNB_OP_F2I, HNB_OP_F2U, HNB_OP_I2F, HNB_OP_U2F, HNB_OP_B2F, HNB_OP_F2B, HNB_OP_I2B,
    /* reductions / vector ops (input width = `width`) */
    HNB_OP_ALL, HNB_OP_ANY,
    HNB_OP_DOT, HNB_OP_LENGTH, HNB_OP_DISTANCE, HNB_OP_NORMALIZE, HNB_OP_CROSS,
    HNB_OP_PACK4UNORM, HNB_OP_PACK4SNORM, HNB_OP_UNPACK4UNORM, HNB_OP_UNPACK4SNORM,
    /* PRNG (reference src/render/vfx_common.wgsl:278-343), varying only */
    HNB_OP_FRAND,    /* width 1: frand(); 2/3: frand2/3(); 4: frand4() */
    HNB_OP_RANDU,    /* rand_uniform_{f,vecN}(a, b) */
    HNB_OP_RANDN,    /* rand_normal_{f,vecN}(mean=a, std_dev=b) */
    /* alive flag, varying only */
    HNB_OP_ALIVE_SET, HNB_OP_ALIVE_AND, HNB_OP_KILL_IF,
    /* Macro ops: closed-form statement/modifier bodies acting on the pinned POSITION /
     * VELOCITY / AGE / LIFETIME registers. Operands a/b/c name the first register of a
     * value (U or V space). An update stream made only of the HNB_OP_M_AGE_TICK ..
     * HNB_OP_M_KILL_AABB ops with U operands runs on the streaming kernel.            */
    HNB_OP_M_AGE_TICK,       /* age = age + r[a]; if aux&1: is_alive = age < lifetime      (src/lib.rs:1223-1258) */
    HNB_OP_M_EULER,          /* position += velocity * r[a]                                (src/lib.rs:1106-1120) */
    HNB_OP_M_VEL_SCALE,      /* velocity *= r[a]                  (LinearDragModifier, modifier/force.rs:284-297) */
    HNB_OP_M_VEL_ADD,        /* velocity += r[a..a+2]             (AccelModifier, modifier/accel.rs:79-86) */
    HNB_OP_M_PIN_SET,        /* pinned register dst[..width] = r[a..]        (SetAttributeModifier on a pinned attribute) */
    HNB_OP_M_RADIAL_ACCEL,   /* velocity += normalize(position - r[a..]) * r[b]            (modifier/accel.rs:162-189) */
    HNB_OP_M_TANGENT_ACCEL,  /* velocity += normalize(cross(r[b..], normalize(position - r[a..]))) * r[c] (accel.rs:281-307) */
    HNB_OP_M_CONFORM_SPHERE, /* a: origin[3] radius influence_dist shell_half_thickness max_attraction_speed attraction_accel sticky_factor; b = delta_time (modifier/force.rs:175-238) */
    HNB_OP_M_KILL_SPHERE,    /* a: center[3], b: sqr_radius, aux&1 = kill_inside           (modifier/kill.rs:76-96) */
    HNB_OP_M_KILL_AABB,      /* a: center[3], b: half_size[3], aux&1 = kill_inside         (modifier/kill.rs:156-181) */
    /* generic-kernel-only macro ops */
    HNB_OP_M_VEL_SPHERE,     /* velocity = normalize(position - (r[a..])) * (r[b])         (modifier/velocity.rs:124-139) */
    HNB_OP_M_POS_CIRCLE,     /* a: center[3] axis[3] radius[1]; aux&1 = Volume             (modifier/position.rs:52-108) */
    HNB_OP_M_POS_SPHERE,     /* a: center[3] radius[1];         aux&1 = Volume             (modifier/position.rs:152-210) */
    HNB_OP_M_POS_CONE3D,     /* a: height[1] top_radius[1] base_radius[1]                  (modifier/position.rs:267-324) */
    HNB_OP_M_VEL_CIRCLE,     /* a: center[3] axis[3] speed[1]                              (modifier/velocity.rs:45-80) */
    HNB_OP_M_VEL_TANGENT,    /* a: origin[3] axis[3] speed[1]                              (modifier/velocity.rs:188-223) */
    HNB_OP_M_ADD_XLATE,      /* position += transform[3].xyz (SimulationSpace::Global, src/lib.rs:518-531) */
    HNB_OP_M_EMIT_EVENTS,    /* EmitSpawnEventModifier (modifier/mod.rs:653-717): if (aux&0x100 ? was_alive && !is_alive : is_alive)
                              * append r[a] (u32) spawn events to child channel aux&0xff (append_spawn_events_N, src/lib.rs:976-993) */
    HNB_OP_COUNT
} HnbOp;

/* Pinned V registers (component registers). Every other attribute is a memory operand
 * (HNB_OP_LDA / HNB_OP_STA); registers from r8 up are temporaries. */
#define HNB_REG_NONE 0xffu
#define HNB_REG_POSITION 0u
#define HNB_REG_VELOCITY 3u
#define HNB_REG_AGE 6u
#define HNB_REG_LIFETIME 7u
#define HNB_REG_FIRST_FREE 8u

/* Program flags */
#define HNB_PROG_GLOBAL_SPACE 0x1u      /* informational: init stream ends with M_ADD_XLATE */
#define HNB_PROG_HAS_RIBBONS 0x2u       /* layout contains RIBBON_ID (post-update sort stage) */
#define HNB_PROG_READS_PARENT 0x4u      /* init stream uses HNB_OP_LDPARENT: instances need hnb_effect_set_parent() */
#define HNB_PROG_EMITS_EVENTS 0x8u

/* Per-attribute update flags */
#define HNB_ATTR_UPD_LOAD 0x1u   /* the update program references the attribute: load it */
#define HNB_ATTR_UPD_STORE 0x2u  /* the update program writes it: store it back */

#define HNB_PROGRAM_MAGIC 0x32424e48u /* "HNB2" */
#define HNB_PROGRAM_VERSION 2u

typedef struct HnbAttrEntry {
    uint16_t attr;        /* HnbAttr */
    uint8_t ncomp;        /* 1..4 components of 4 bytes (vec3 is packed 12 B, no padding) */
    uint8_t reg;          /* first V register of a pinned attribute, HNB_REG_NONE otherwise */
    uint8_t scalar_type;  /* HnbScalarType */
    uint8_t update_flags; /* HNB_ATTR_UPD_* */
    uint16_t reserved;
} HnbAttrEntry;

typedef struct HnbPropEntry {
    char name[48];        /* NUL-terminated property name (Module::add_property) */
    uint8_t scalar_type;  /* HnbScalarType */
    uint8_t ncomp;        /* 1..4 */
    uint16_t word_offset; /* offset into the per-instance property block, in 32-bit words */
    uint32_t default_bits[4];
} HnbPropEntry;

/* Flat program blob: header, then attrs[], props[], uniform code, init code, update code. */
typedef struct HnbProgramHeader {
    uint32_t magic, version, total_size;
    uint32_t capacity;          /* EffectAsset::capacity (src/asset.rs:391) */
    uint32_t flags;             /* HNB_PROG_* */
    uint32_t n_attrs, n_props, prop_words;
    uint32_t uniform_len, init_len, update_len;  /* instruction counts */
    uint32_t n_uregs;                            /* U registers (parameter block words) */
    uint32_t init_regs, update_regs;             /* highest V register used + 1 */
    uint32_t attrs_off, props_off, uniform_off, init_off, update_off; /* byte offsets from blob start */
    uint32_t n_event_channels;  /* number of child event channels this program appends to */
    uint32_t parent_n_attrs;    /* HNB_PROG_READS_PARENT: u32 HnbAttr ids read from the parent particle, at parent_attrs_off */
    uint32_t parent_attrs_off;
    /* Attributes the asset's RENDER modifiers read every frame, as a bit mask over HnbAttr ids (bit a of lo | hi << 32): what the reference
     * declares with impl_mod_render!(ColorOverLifetimeModifier, &[Attribute::AGE, Attribute::LIFETIME]) (src/modifier/output.rs:310-312,
     * 423-425, 602-611). The simulation does not execute render modifiers; the mask tells it which planes a renderer behind
     * hnb_effect_device_view reads after every frame (HNB_AGE_COHORT_AUTO: keep AGE current). 0 in blobs written before the field existed. */
    uint32_t render_reads_lo, render_reads_hi;
} HnbProgramHeader;

/* ---------------------------------------------------------------------------------- */
/* Per-frame inputs                                                                   */
/* ---------------------------------------------------------------------------------- */
/* Mirrors GpuSimParams (src/render/mod.rs:218-243, vfx_common.wgsl:3-20). */
typedef struct HnbSimParams {
    float delta_time, time;
    float virtual_delta_time, virtual_time;
    float real_delta_time, real_time;
} HnbSimParams;

/* Mirrors the readable part of GpuEffectMetadata + DispatchIndirectArgs
 * (vfx_common.wgsl:186-255, src/render/mod.rs:566-622). */
typedef struct HnbEffectMetadata {
    uint32_t capacity;
    uint32_t alive_count;          /* after the last simulated frame */
    uint32_t max_update;           /* particles the last update pass processed */
    uint32_t max_spawn;            /* capacity - alive_count: spawn cap of the next init pass */
    uint32_t indirect_write_index; /* ping-pong column the last update wrote (EffectMetadata::indirect_write_index) */
    uint32_t particle_counter;
    uint32_t instance_count;       /* render instance count == survivors of the last update */
    uint32_t dispatch_x;           /* ceil(alive_count / 64): indirect args of the next update */
    uint32_t dead_count;           /* particles killed by the last update */
    uint32_t spawned;              /* particles spawned by the last init */
    uint32_t fault;                /* 0; 1 = a particle died in a frame whose list kernels the host had proven unnecessary and skipped (a bug) */
    uint32_t reserved;
} HnbEffectMetadata;

typedef struct HnbContext HnbContext;
typedef struct HnbProgram HnbProgram;
typedef struct HnbEffect HnbEffect;

/* ---------------------------------------------------------------------------------- */
/* Entry points                                                                       */
/* ---------------------------------------------------------------------------------- */
const char* hnb_last_error(void);
const char* hnb_version(void);

/* One context per GPU. Replaces the render-world resources (EffectCache, EffectsMeta,
 * src/render/mod.rs:2553-2880). Fails with HNB_ERR_NO_DEVICE when no HIP device exists. */
int hnb_ctx_create(int device_id, HnbContext** out_ctx);
int hnb_ctx_destroy(HnbContext* ctx);
/* Stream used by hnb_simulate (a hipStream_t; NULL selects the context's own stream). */
int hnb_ctx_set_stream(HnbContext* ctx, void* hip_stream);
int hnb_ctx_synchronize(HnbContext* ctx);
/* Context options, applied to the programs created afterwards.
 * HNB_OPT_LIST_ORDER: order of an effect's alive list after each frame. The reference's order is whatever its
 * atomics produce (vfx_update.wgsl:161-165); both values are deterministic, legal outcomes:
 *   HNB_LIST_ORDER_SPAWN (default)  serial-thread order: survivors keep their relative order, spawns are appended;
 *   HNB_LIST_ORDER_SLOT             survivors in increasing slot order: under steady spawn/kill churn the update
 *                                   keeps streaming through memory instead of gathering 12 bytes at random. */
#define HNB_OPT_LIST_ORDER 1u
#define HNB_LIST_ORDER_SPAWN 0u
#define HNB_LIST_ORDER_SLOT 1u
/* HNB_OPT_ALTERNATE (default 1): the update walks an effect's chunks in alternating directions from
 *   frame to frame, so that each frame starts on what the previous one wrote last and finds it in the 256 MiB Infinity Cache.
 * HNB_OPT_SKIP_LISTS (default 1): frames in which provably no particle can die or spawn (the update
 *   kernel publishes a lower bound of every particle's remaining life) do not launch the list kernels.
 * Both are pure scheduling choices: results are identical with either value. They apply from the next hnb_simulate on. */
#define HNB_OPT_ALTERNATE 2u
#define HNB_OPT_SKIP_LISTS 3u
/* HNB_OPT_AGE_COHORT (fixed in a program when it is created; default HNB_AGE_COHORT_AUTO): a 4096-slot chunk whose alive particles share one
 *   AGE, bit for bit - every burst effect - keeps it in one word instead of reading and writing 8 bytes per particle and frame. THE ONE OPTION
 *   THAT CHANGES DEVICE-VISIBLE STATE: the AGE plane of such a chunk is stale until hnb_effect_materialise (or a host read, which does the
 *   same) - see "Device-side output" below. OFF: the plane is current after every frame. LEAN: bandwidth-bound update stacks only
 *   (drag / gravity / Euler); ALL: every eligible stack (measured slower for issue-bound stacks such as force_field.rs).
 * HNB_OPT_CULL_LIFETIME, HNB_OPT_HORIZON (fixed at program creation; default 1), HNB_OPT_TRANSPOSE, HNB_OPT_SCENE_MERGE, HNB_OPT_SUFFIX_PROOF
 *   (from the next hnb_simulate on; default 1): lifetime culling, row-chunk death horizons, the LDS transpose of vec3 planes, shared launches
 *   for small programs, the ribbon "casualties are the last rows" proof. Results are identical with either value: they exist for
 *   A/B measurements. The library reads NO environment variable for any of these. */
#define HNB_OPT_AGE_COHORT 4u
#define HNB_AGE_COHORT_OFF 0u
#define HNB_AGE_COHORT_LEAN 1u
#define HNB_AGE_COHORT_ALL 2u
#define HNB_AGE_COHORT_AUTO 3u   /* THE DEFAULT: chosen from the asset. Where the render modifiers read AGE after every frame (HnbProgramHeader::
                                  * render_reads_*: ColorOverLifetime / SizeOverLifetime) nothing is ever stale for a renderer behind
                                  * hnb_effect_device_view: an effect of 65,536 slots or more keeps the cohorts and its update kernel writes a cohort
                                  * chunk's common age into the plane as it goes (write-only: 4 of the 8 bytes per particle the cohort saves; no
                                  * second pass, no extra launch), a smaller one keeps per-particle ages in the plane (as OFF). Every other program
                                  * gets LEAN. A headless host that never looks at AGE between frames asks for LEAN. */
#define HNB_OPT_CULL_LIFETIME 5u
#define HNB_OPT_HORIZON 6u
#define HNB_OPT_TRANSPOSE 7u
#define HNB_OPT_SCENE_MERGE 8u
#define HNB_OPT_SUFFIX_PROOF 9u
/* HNB_OPT_RING_LISTS (default 1; from the next hnb_simulate on): a ribbon effect for which the host can prove both that this frame's spawns sort in
 *   front of every older particle and that the frame's casualties are the list's last rows (one ribbon id, ages from +0, one lifetime: examples/
 *   ribbon.rs) keeps its alive list as a RING: the init pass writes the spawns in front of the list's head, the head moves, the count drops - no row is
 *   rewritten (HnbDeviceMeta::list_column carries the head). 0: the list is rotated by rewriting it, as before round 5. Same list either way. */
#define HNB_OPT_RING_LISTS 16u
/* HNB_OPT_SLOT_INIT (default 1; from the next hnb_simulate on): a frame that spawns an eighth or more of a program's slots (a burst, the re-burst of
 *   SpawnerSettings::burst(count, period) into slots a die-off left in killing order) runs its init pass over the SLOTS - contiguous plane stores
 *   whatever the dead list looks like - instead of over the spawn ranks; programs whose init reads PARTICLE_COUNTER or a parent particle, and
 *   ribbon effects, keep the rank-major pass. Same state bit for bit (the serial pop order only decides the LIST, which is copied). 0: never;
 *   2: every eligible program in every frame that spawns anything, whatever its size (tests, A/B runs). */
#define HNB_OPT_SLOT_INIT 17u
/* HNB_OPT_DIRECT_UPLOAD (default 1; from the next hnb_simulate on): the frame's parameter block (instance rows, uniform blocks, job tables: a few hundred
 *   bytes to a few KiB) is WRITTEN BY THE HOST into fine-grained device memory through the PCIe BAR - posted writes in front of the first launch's
 *   doorbell, the way the HIP runtime places kernel arguments - where the device exposes a large BAR (hipDeviceProp_t::isLargeBar; checked against a
 *   device-side read-back when the slots are created). 0, or a device without one: hipMemcpyAsync on an internal stream and a host wait, 10 us of every
 *   hnb_simulate. Same results; hnb_program_kernel_info says which way the context's frames went. */
#define HNB_OPT_DIRECT_UPLOAD 18u
/* HNB_OPT_OVERLAP_UPDATES (default 1; from the next hnb_simulate on): when one program of the context holds at least four times the slots
 *   of all the others together (and >= 1M), the update phase of the others - update, spawn-event ordering, lists, sort: independent chains -
 *   runs on an internal second stream next to the heavy program's update and joins the context's stream before hnb_simulate returns.
 *   Same results; frames on which kernel timing is sampled use one stream. */
#define HNB_OPT_OVERLAP_UPDATES 10u
/* HNB_OPT_STREAM_HINTS (default 1; from the next hnb_simulate on): programs whose frame touches more than the 256 MiB Infinity Cache holds
 *   read and write their lists, and read the update's read-only planes, with the nontemporal hint (a cache-policy choice: same results). */
#define HNB_OPT_STREAM_HINTS 11u
/* HNB_OPT_JIT_ASYNC (default 0; applies to programs created afterwards): hnb_program_create does not wait for hiprtc. A program whose specialised
 *   kernels are not in the cache starts on the ahead-of-time / interpreter kernels (same results bit for bit) and a thread of the library's own compiles
 *   them, one program after the other; the first hnb_simulate that finds them ready uses them. (The reference's pipelines are compiled asynchronously too,
 *   and an effect whose pipelines are not ready is SKIPPED, src/render/mod.rs:3852-3900; here it is simulated from its first frame.) hnb_ctx_destroy waits
 *   for a compilation in flight. */
#define HNB_OPT_JIT_ASYNC 13u
/* HNB_OPT_TEST_BREAK_PROOF (default 0): a TEST HOOK, never for production. 1 = every frame that spawns nothing is treated as proven to have no
 *   casualty (HNB_OPT_SKIP_LISTS's proof, claimed without evidence). A particle that dies in such a frame raises HnbEffectMetadata::fault and
 *   leaves the lists stale: what hnb_effect_check, hnb_effect_compare and bench.py's parity gate exist to notice, and are tested with.
 *   REFUSED (HNB_ERR_INVALID_ARG) unless the process environment holds HNB_ENABLE_TEST_HOOKS=1: no host switches it on by accident. */
#define HNB_OPT_TEST_BREAK_PROOF 15u
/* HNB_OPT_SET_MODULE (default HNB_SET_MODULE_CACHED; from the next hnb_simulate on): the launches the small programs of a context share
 *   (HNB_OPT_SCENE_MERGE) run the byte-code INTERPRETERS - the only code that fits every program - unless the context has a SET MODULE: one
 *   hiprtc module whose two kernels switch, per job, into the SPECIALISED code of each program (the counterpart of the reference compiling one
 *   WGSL module per effect, src/lib.rs:805-1336, for effects that share a dispatch here). The set is every program of the context that can take
 *   part in merged launches (independent of other effects, <= 65,536 slots over its instances, narrow register file); the module is keyed by the
 *   set (not by creation order or multiplicity) and lives in the same on-disk cache as the per-program kernels.
 *     OFF      the interpreters serve the merged launches (round 3's behaviour)
 *     CACHED   a module is used when the cache holds it (hnb_jit_precompile_set, or an earlier run with COMPILE); nothing is compiled on the frame path
 *     COMPILE  a missing module is compiled inside hnb_simulate (seconds to a minute, once per set: loading screens, tests, benchmarks)
 *     BACKGROUND  a missing module is compiled on a thread of the library's own while the frames go on (on the interpreters); hnb_simulate loads it
 *              when it is ready. One compilation at a time; hnb_ctx_destroy waits for one in flight (hiprtc cannot be interrupted)
 *   Same results bit for bit in every mode; programs created after the module was built keep their launches on the interpreters until the set has
 *   stood for two merged frames and a module for it is found (or compiled). */
#define HNB_OPT_SET_MODULE 12u
#define HNB_SET_MODULE_OFF 0u
#define HNB_SET_MODULE_CACHED 1u
#define HNB_SET_MODULE_COMPILE 2u
#define HNB_SET_MODULE_BACKGROUND 3u
int hnb_ctx_set_option(HnbContext* ctx, uint32_t option, uint32_t value);

/* Replaces EffectShaderSources::generate + pipeline specialisation (src/lib.rs:805-1336). */
int hnb_program_create(HnbContext* ctx, const void* blob, size_t blob_size, HnbProgram** out_prog);
int hnb_program_destroy(HnbProgram* prog);
/* Validate a blob without a device (structure, register bounds, opcode range). */
int hnb_program_validate(const void* blob, size_t blob_size);

/* Replaces slab allocation + initial metadata (effect_cache.rs:298-323, mod.rs:6048-6070):
 * dead_index[i] = i, alive_count = 0, max_spawn = capacity, indirect_write_index = 0.
 * `slot_base` offsets the particle index used for PRNG seeding / Attribute::ID so that a
 * capacity slab of a larger logical effect reproduces the single-GPU result. */
int hnb_effect_create(HnbProgram* prog, uint32_t slot_base, HnbEffect** out_fx);
int hnb_effect_destroy(HnbEffect* fx);
/* EffectParent (src/render/event.rs, vfx_init.wgsl:123-129,166-171): `child`'s init pass consumes the spawn
 * events `parent`'s update appends on `channel` (EmitSpawnEventModifier::child_index; the N-th child reads
 * channel N), one frame later, and may read the emitting parent particle (InheritAttributeModifier,
 * parent_attr). `event_capacity` = arrayLength(&event_buffer.spawn_events): events past it in a frame are
 * dropped; the reference hard-codes 256 (src/render/event.rs:267). Both effects must share a context. */
int hnb_effect_set_parent(HnbEffect* child, HnbEffect* parent, uint32_t channel, uint32_t event_capacity);

/* Per-frame inputs (ExtractSchedule data, src/render/mod.rs:2670-2691,4437-4445):
 * sim clocks for every effect of the context ... */
int hnb_frame_begin(HnbContext* ctx, const HnbSimParams* params);
/* ... and per effect: CPU spawn count (EffectSpawner::tick), PRNG seed, row-major 3x4
 * emitter transform (NULL = identity). */
int hnb_effect_set_frame(HnbEffect* fx, uint32_t spawn_count, uint32_t seed, const float* transform3x4);
/* The same for a range of instances of one program in one call (instances are numbered in creation order, see
 * hnb_effect_index; destroying an instance moves the last one into its place): the reference re-uploads one
 * GpuSpawnerParams row per instance per frame (src/render/mod.rs:4316,4679-4687). transforms3x4 may be NULL. */
int hnb_program_set_frames(HnbProgram* prog, uint32_t first, uint32_t count, const uint32_t* spawn_counts, const uint32_t* seeds,
                           const float* transforms3x4);
int hnb_effect_index(HnbEffect* fx, uint32_t* out_index);
/* SimulationCondition (src/asset.rs, src/spawn.rs:983-991, src/render/mod.rs:4347-4356): an effect whose asset
 * says WhenVisible is neither ticked nor simulated while it is not visible. The host decides visibility and
 * passes 0 here: the instance is skipped by the following hnb_simulate calls (its state is frozen, its
 * pending spawn request is dropped) until 1 is passed again. Default: simulated. */
int hnb_effect_set_simulated(HnbEffect* fx, int simulated);
/* Properties (EffectProperties::set, src/properties.rs:216-395). `n_words` 32-bit words. */
int hnb_effect_set_property(HnbEffect* fx, const char* name, const void* value, uint32_t n_words);

/* Enqueue one simulation frame for every effect of the context:
 * init -> (indirect/prefix-sum folded) -> update+kill+compaction. Replaces `simulate`
 * (src/render/mod.rs:6942-7613). Asynchronous. */
int hnb_simulate(HnbContext* ctx);

/* ---- Device-side output -------------------------------------------------------------------------------------------------------
 * Where the reference's hot path ENDS: GPU-resident buffers the next stage binds directly - the particle buffer, the indirect index
 * buffer and the effect metadata row that vfx_indirect.wgsl:57-85 turns into draw-indirect arguments (bind groups
 * src/render/mod.rs:5152-5820; vfx_render.wgsl reads particle_buffer[indirect_buffer[instance_index * 3 + column]]). A consumer on the
 * same GPU (a renderer through external-memory interop, a baking kernel, a physics query) gets the same three things as device pointers,
 * without a copy and without a synchronisation:
 *     HnbDeviceView v; hnb_effect_device_view(fx, &v);
 *     consumer<<<.., (hipStream_t)v.stream>>>(v);    // row r < v.meta->alive_count: slot = v.alive_list[v.meta->list_column & 1][((v.meta->list_column >> 1) + r) % v.capacity];
 *                                                    //   position = ((const float*)v.attrs[i].plane) + 3 * slot   (attrs[i].attr == HNB_ATTR_POSITION)
 * Stream-ordering contract: the view describes the effect AFTER every hnb_simulate enqueued so far. Kernels enqueued on v.stream (or on
 * another stream behind an event recorded on v.stream) after the call and before the next hnb_simulate see exactly that state; they must
 * have completed (or the next hnb_simulate be ordered behind them, which it is on v.stream) before the planes are written again.
 * Lifetime of the pointers: planes and lists live as long as the effect; `meta` / `meta_next` ALTERNATE from frame to frame and move when an
 * effect of the program is created or destroyed - fetch the view again after each hnb_simulate (a few stores, no HIP call).
 * Attributes in stale_attr_mask are current only after hnb_effect_materialise(fx, mask), which is enqueued on the simulation stream like a
 * frame. With the default HNB_AGE_COHORT_AUTO the mask is empty for every asset whose render modifiers read AGE (ColorOverLifetime /
 * SizeOverLifetime, src/modifier/output.rs:310-312: HnbProgramHeader::render_reads_*) - its ages are in the plane or put there by hnb_simulate - and for
 * programs created with HNB_AGE_COHORT_OFF; AGE is stale under LEAN / ALL, and under AUTO for assets whose renderer does not read it.
 * Free slots hold the values their last particle died with. */
typedef struct HnbDeviceMeta {          /* one 32-byte row per effect instance, device-resident; written by the frame's last kernel */
    uint32_t alive_count;               /* EffectMetadata::alive_count after the frame */
    uint32_t particle_counter;
    uint32_t list_column;               /* bit 0: which of HnbDeviceView::alive_list[2] holds the alive list (changes only in frames with casualties);
                                         * bits 1..31: list_head - row r of the list is alive_list[list_column & 1][(list_head + r) % capacity]. The head is 0
                                         * for every effect except single-ribbon trails whose list is kept as a RING (HNB_OPT_RING_LISTS): their spawns go in
                                         * front of the head and their casualties drop off the end, so nothing is rewritten. A reader that always applies the
                                         * formula is right for every effect. */
    uint32_t max_update;                /* rows the frame's update pass processed */
    uint32_t dead_count, spawned;
    uint32_t indirect_write_index;      /* EffectMetadata::indirect_write_index as the reference counts it (flips every frame) */
    uint32_t instance_count;            /* render instance count == alive_count: a draw-indirect source (vfx_indirect.wgsl:66-75) */
} HnbDeviceMeta;
typedef struct HnbDeviceAttr {
    uint16_t attr;                      /* HnbAttr */
    uint8_t ncomp, scalar_type;         /* 1..4 components of 4 bytes; HnbScalarType */
    uint16_t stride_bytes;              /* ncomp * 4: planes are packed (vec3 = 12 bytes per slot) */
    uint16_t reserved;
    void* plane;                        /* device pointer, indexed by SLOT: capacity * stride_bytes bytes */
} HnbDeviceAttr;
#define HNB_VIEW_MAX_ATTRS 40u
typedef struct HnbDeviceView {
    uint32_t struct_size;               /* sizeof(HnbDeviceView) of the library that filled it */
    int32_t device;                     /* HIP device ordinal the pointers belong to */
    void* stream;                       /* hipStream_t the simulation is enqueued on */
    uint32_t capacity, slot_base, n_attrs, reserved;
    uint64_t stale_attr_mask;           /* bit a: plane of HnbAttr a needs hnb_effect_materialise before a device-side read */
    const uint32_t* alive_list[2];      /* both list columns: capacity slot indices each; rows [0, meta->alive_count) of column meta->list_column are the list */
    const uint32_t* dead_list;          /* rows [meta->alive_count, capacity): the free slots, next spawn first */
    const HnbDeviceMeta* meta;          /* this effect's row after the frames enqueued so far */
    const HnbDeviceMeta* meta_next;     /* the row the NEXT hnb_simulate will write (the two alternate) */
    HnbDeviceAttr attrs[HNB_VIEW_MAX_ATTRS];   /* [n_attrs], layout order */
} HnbDeviceView;
int hnb_effect_device_view(HnbEffect* fx, HnbDeviceView* out_view);
/* The same for ALL instances of a program in one call: the shape the reference binds per BATCH - one particle buffer, one indirect buffer and
 * the per-instance slices / metadata rows (src/render/batch.rs:348-386, vfx_prefix_sum.wgsl:13-43). Instance k (hnb_effect_index) lives in the
 * slab at slabs[k]; its lists and planes are at the byte offsets below from that base; its metadata row is meta[k]. Same stream-ordering
 * contract and pointer lifetimes as HnbDeviceView: `slabs`, `meta`, `meta_next` are device arrays that move when an instance is created or
 * destroyed and meta / meta_next alternate every frame - fetch the view after each hnb_simulate. A consumer kernel over the whole batch:
 *     k = blockIdx.y; base = (const char*)v.slabs[k]; m = v.meta[k]; list = (const uint32_t*)(base + v.alive_list_off[m.list_column & 1]);
 *     row r < m.alive_count: slot = list[((m.list_column >> 1) + r) % v.capacity]   (bits 1..31 of list_column: the list's head, non-zero only for
 *     ribbon effects kept as a ring - on by default); position = (const float*)(base + v.attrs[i].plane_off) + 3 * slot */
typedef struct HnbProgramAttr {
    uint16_t attr;                      /* HnbAttr */
    uint8_t ncomp, scalar_type;
    uint16_t stride_bytes, reserved;
    uint64_t plane_off;                 /* bytes from an instance's slab base */
} HnbProgramAttr;
typedef struct HnbProgramView {
    uint32_t struct_size;
    int32_t device;
    void* stream;
    uint32_t capacity, n_instances, n_attrs, reserved;
    uint64_t stale_attr_mask;
    const uint64_t* slabs;              /* device: [n_instances] slab base addresses */
    const HnbDeviceMeta* meta;          /* device: [n_instances] rows after the frames enqueued so far */
    const HnbDeviceMeta* meta_next;     /* ... the rows the next hnb_simulate will write */
    uint64_t alive_list_off[2], dead_list_off;
    HnbProgramAttr attrs[HNB_VIEW_MAX_ATTRS];
} HnbProgramView;
int hnb_program_device_view(HnbProgram* prog, HnbProgramView* out_view);
/* Makes the planes of the attributes in `attr_mask` (bit a = HnbAttr a) current for device-side readers; enqueued on the simulation stream,
 * returns at once. A no-op for attributes that are never stale. */
int hnb_effect_materialise(HnbEffect* fx, uint64_t attr_mask);

/* Readback (synchronising; reporting / parity only, never on the frame path). */
int hnb_effect_metadata(HnbEffect* fx, HnbEffectMetadata* out);
int hnb_effect_alive_count(HnbEffect* fx, uint32_t* out);
/* Copy one attribute plane, indexed by slot: `dst` receives capacity * ncomp * 4 bytes. */
int hnb_effect_read_attr(HnbEffect* fx, uint32_t attr, void* dst, size_t dst_size);
/* Copy the alive list (alive_count slot indices, in list order) and the dead list
 * (capacity - alive_count free slots, stack order from the top). */
int hnb_effect_read_alive_list(HnbEffect* fx, uint32_t* dst, size_t dst_count);
int hnb_effect_read_dead_list(HnbEffect* fx, uint32_t* dst, size_t dst_count);
/* Overwrite an attribute plane / force counters (tests and state restore). */
int hnb_effect_write_attr(HnbEffect* fx, uint32_t attr, const void* src, size_t src_size);

/* ---- Verification on the device ----------------------------------------------------------------------------------------------------
 * State too large to read back and compare on the host in reasonable time (16.7M particles: 0.7 GB per effect) is checked where it lives.
 * Both calls materialise stale planes (hnb_effect_materialise), synchronise the effect's context and return small host structs.
 *
 * hnb_effect_check: the invariants the reference's passes maintain between frames (src/render/vfx_update.wgsl:148-166, vfx_init.wgsl:141-143):
 *   rows [0, alive_count) of the alive list and rows [alive_count, capacity) of the dead list together name every slot exactly once, the alive
 *   byte of a slot says which of the two lists it is in, every alive particle of a program that reaps by lifetime has age < lifetime
 *   (src/lib.rs:1223-1258), and the device raised no fault (HnbEffectMetadata::fault: a host-side proof that did not hold).
 * hnb_effect_compare: two effects of the same layout on the same device, bit for bit: the eight counter words, the alive rows, the dead rows,
 *   every attribute plane over every slot. What bench.py holds the state its timed frames produced against: a second context that replayed the
 *   same frames with every proof and hint of hnb_ctx_set_option switched off. */
typedef struct HnbEffectCheck {
    uint32_t ok;                     /* 1: every count below is zero */
    uint32_t capacity, alive_count;
    uint32_t bad_slots;              /* list rows that name a slot >= capacity */
    uint32_t duplicate_slots;        /* slots named by more than one row */
    uint32_t alive_byte_mismatches;  /* slots whose alive byte disagrees with the list that names them */
    uint32_t alive_past_lifetime;    /* alive particles with !(age < lifetime) (0 for programs that do not reap by lifetime) */
    uint32_t fault;                  /* HnbEffectMetadata::fault */
} HnbEffectCheck;
typedef struct HnbEffectDiff {
    uint32_t equal;                  /* 1: no difference anywhere */
    uint32_t counter_diffs;          /* of the counter words alive_count, particle_counter, max_update, dead_count, spawned, indirect_write_index, instance_count
                                      * (not the list column: where the rows live is not what they are) */
    uint64_t alive_list_diffs, dead_list_diffs, attr_diffs;   /* differing 32-bit words */
    int32_t first_section;           /* -1: none; 0 alive list, 1 dead list, 2 + HNB_ATTR_*: that attribute's plane */
    uint32_t reserved;
    uint64_t first_index;            /* word index of the first difference inside that section */
} HnbEffectDiff;
int hnb_effect_check(HnbEffect* fx, HnbEffectCheck* out);
int hnb_effect_compare(HnbEffect* a, HnbEffect* b, HnbEffectDiff* out);

/* Ribbon sort by (RIBBON_ID, AGE bits) ascending, stable (vfx_sort*.wgsl, src/render/mod.rs:7372-7612).
 * hnb_simulate runs it after the update of every effect whose layout has RIBBON_ID, like the reference;
 * this entry point only reports whether the effect is such an effect (OK) or not (INVALID_ARG). */
int hnb_effect_sort_ribbons(HnbEffect* fx);

/* Which kernels run this program, as text: "init=jit|interp|none update=aot-stream:<name>|jit-stream|
 * jit-generic|interp-stream|interp-generic", followed by the specialisation log if it failed.
 * At creation the library specialises the kernels a program would otherwise interpret (the
 * counterpart of the reference compiling generated WGSL per effect, src/lib.rs:805-1336) with
 * hiprtc; HNB_JIT=0 in the environment keeps the interpreter kernels, HNB_JIT_CACHE=<dir> moves the
 * on-disk cache of compiled code objects (default: jit_cache/ next to the library). */
int hnb_program_kernel_info(HnbProgram* prog, char* buf, size_t buf_size);
/* Compile and cache the specialised kernels of a program blob. Needs no device (build boxes). */
int hnb_jit_precompile(const void* blob, size_t blob_size);
/* Compile and cache the SET MODULE (HNB_OPT_SET_MODULE) of the given program blobs: what a context that holds exactly these programs as its
 * small effects looks up. Order and duplicates do not matter; blobs that can never join a merged launch are skipped as hnb_simulate skips them
 * (wide register file, capacity > 65,536, spawn events in or out). The runtime set also leaves out programs whose INSTANCES rule them out (more
 * than 65,536 slots over all instances, an instance with a parent): pass the blobs of the effects that stay small. Needs no device. */
int hnb_jit_precompile_set(const void* const* blobs, const size_t* blob_sizes, uint32_t n_blobs);

/* Timing helper: average device time in ms of the update kernel, of the compaction kernel that
 * follows it (event after update -> event after compact, i.e. including the launch gap) and of the
 * init kernel, over the frames simulated since the last reset (HIP events on the simulation stream). */
int hnb_ctx_enable_kernel_timing(HnbContext* ctx, int every_n_frames); /* 0 = off; n: time every n-th hnb_simulate (events cost ~20 us of stream bubbles per timed frame).
 * A timed frame runs one launch per program and stage: the shared launches of small programs (HNB_OPT_SCENE_MERGE) are not used in it, so the
 * averages describe each program's own specialised kernels, not the merged interpreter launches the untimed frames of a small-effect scene run. */
int hnb_ctx_kernel_timing(HnbContext* ctx, double* update_ms_avg, double* compact_ms_avg, double* init_ms_avg, uint32_t* frames);
/* the same averages over the kernels of ONE program of the context (a context with several programs: parent and child effects) */
int hnb_program_kernel_timing(HnbProgram* prog, double* update_ms_avg, double* compact_ms_avg, double* init_ms_avg, uint32_t* frames);
/* Profiling aid: enqueues an empty kernel of `tag` workgroups x 1 thread (1 <= tag <= 65535) on the simulation stream. It shows up in
 * a kernel trace / counter collection (rocprofv3) as `k_marker` with Grid_Size == tag, so that a tool can cut the dispatch list of
 * a run into the sections it cares about (bench.py brackets the timed frames of every configuration with it). */
int hnb_ctx_profile_marker(HnbContext* ctx, uint32_t tag);

/* ---- multi-GPU reporting ------------------------------------------------------------------------------------------------
 * Effects shard by instance and by capacity slab with no inter-GPU dependency per frame: one HnbContext per GPU,
 * each driven by its own submit thread of ONE process (the reference's seam is one process: src/plugin.rs:202-256,
 * src/render/mod.rs:126-131; examples/multi_gpu.c is such a host), or one rank per process. The only collective is the all-reduce of
 * the alive-particle counters, over RCCL (xGMI). librccl is resolved at run time: a single-GPU host does not need it. */
typedef struct HnbComm HnbComm;
#define HNB_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* Which collective library hnb_comm_* binds (dlopen). Default (never called, or path NULL): librccl by name. Must precede the first
 * hnb_comm_* call of the process that needs the library; afterwards it fails. HNB_COMM_LIB_DUPLICATE_DEVICES: the library accepts a
 * communicator that names one device twice (real RCCL does not) - for stand-ins such as tests/fake_rccl, which let the collective
 * branch run with two contexts on a one-GPU box.
 * HNB_COMM_LIB_SINGLE_RANK: a communicator of ONE context / ONE rank is built with the library too (ncclCommInitAll over one device,
 * ncclCommInitRank with n_ranks = 1) and its all-reduce runs on the context's stream, instead of the host path such a communicator takes by
 * default (a one-GPU host need not load librccl at all): the way to execute the collective branch for real on a one-GPU machine. */
#define HNB_COMM_LIB_DUPLICATE_DEVICES 1u
#define HNB_COMM_LIB_SINGLE_RANK 2u
int hnb_comm_set_library(const char* path, uint32_t flags);
/* What the communicator reduces through, as text: "rccl <path of the library the symbols were resolved from> ranks=<n> local=<contexts>"
 * or "host-sum ranks=<n> local=<contexts>". */
int hnb_comm_describe(HnbComm* comm, char* buf, size_t buf_size);
/* one process, n contexts (ncclCommInitAll over their devices; contexts that share a device are reduced through the host).
 * A communicator holds its contexts: destroy it before them (hnb_ctx_destroy refuses otherwise). */
int hnb_comm_create_local(HnbContext* const* ctxs, uint32_t n_ctx, HnbComm** out_comm);
/* one rank per process: rank 0 calls hnb_comm_unique_id and hands the 128 bytes to the others (ncclCommInitRank) */
int hnb_comm_unique_id(void* out_id);
int hnb_comm_create_rank(HnbContext* ctx, const void* id, uint32_t rank, uint32_t n_ranks, HnbComm** out_comm);
/* out_totals[e] = sum over every context (of every rank) of the alive count of its e-th effect. `effects` is context-major,
 * [local contexts][n_effects]; NULL entries count 0. Enqueued on the contexts' simulation streams behind their frames; returns when
 * the totals are in out_totals (the one call of this API that synchronises every context). */
int hnb_comm_allreduce_alive(HnbComm* comm, HnbEffect* const* effects, uint32_t n_effects, uint64_t* out_totals);
int hnb_comm_destroy(HnbComm* comm);

#ifdef __cplusplus
}
#endif
#endif /* HANABI_AMD_H */
