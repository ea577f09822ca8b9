/* TEST INFRASTRUCTURE / CPU BASELINE — a tuned CPU port of the *lowered* streaming update (SURVEY.md §8d "CPU baseline
 * timing": SoA, one parallel-for over particles, -O3 -march=native -ffp-contract=off -fopenmp).
 *
 * The reference has no CPU simulation path, so there is nothing of bevy_hanabi to time on the host.
 * hanabi_oracle.c restates the WGSL semantics by walking the expression tree per particle — the right shape for a
 * checker, a straw man as a baseline. This file is the honest counterpart of the GPU kernel: the same packed SoA planes
 * (vec3 = 12 B), the same closed-form statements the update of vfx_update.wgsl:105-167 reduces to for the example
 * stacks, evaluated blockwise so that every statement is a flat, auto-vectorisable loop over a cache-resident block:
 *
 *   HCS_AGE_TICK   age += dt; is_alive = age < lifetime                    src/lib.rs:1223-1258
 *   HCS_VEL_SCALE  velocity *= s          (LinearDragModifier, s = max(0, 1 - drag*dt))   src/modifier/force.rs:284-297
 *   HCS_VEL_ADD    velocity += a          (AccelModifier, a = accel*dt)    src/modifier/accel.rs:79-86
 *   HCS_EULER      position += velocity * dt                               src/lib.rs:1106-1120
 *
 * Uniform operands (s, a, dt) are computed by the caller once per frame, as the GPU path does on the host.
 * bench.py checks this port bit-for-bit against hanabi_oracle.c on the same particles before timing it.
 * Only tests/ and bench.py's cpu_baseline leg may load this code.
 */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { HCS_AGE_TICK = 1, HCS_VEL_SCALE = 2, HCS_VEL_ADD = 3, HCS_EULER = 4 };
typedef struct { uint32_t op; float v[3]; } HcsOp;

#define HCS_BLOCK 4096u /* particles per block: 56 B x 4096 = 224 KiB, L2-resident while the statements run over it */

int hcs_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void hcs_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* Parallel copy with the schedule of hcs_update: first touch places every block's pages on the NUMA node of the thread
 * that will update it. */
void hcs_copy(void* dst, const void* src, uint64_t n_particles, uint32_t bytes_per_particle) {
    const uint64_t n_blocks = (n_particles + HCS_BLOCK - 1) / HCS_BLOCK;
#pragma omp parallel for schedule(static)
    for (uint64_t b = 0; b < n_blocks; ++b) {
        const uint64_t first = b * HCS_BLOCK, n = (n_particles - first) < HCS_BLOCK ? (n_particles - first) : HCS_BLOCK;
        memcpy((char*)dst + first * bytes_per_particle, (const char*)src + first * bytes_per_particle, n * bytes_per_particle);
    }
}

/* One frame of the update over slots [0, n). alive: one byte per slot (1 = alive), cleared for particles that die.
 * Returns the number of particles that died. Statements also run on particles that expire in this frame and their
 * results are stored, as in the reference (the kill only takes the particle off the lists). */
uint64_t hcs_update(float* restrict pos, float* restrict vel, float* restrict age, const float* restrict life, uint8_t* restrict alive,
                    uint64_t n_particles, const HcsOp* ops, uint32_t n_ops) {
    const uint64_t n_blocks = (n_particles + HCS_BLOCK - 1) / HCS_BLOCK;
    uint64_t died_total = 0;
#pragma omp parallel for schedule(static) reduction(+ : died_total)
    for (uint64_t b = 0; b < n_blocks; ++b) {
        const uint64_t first = b * HCS_BLOCK;
        const uint32_t n = (uint32_t)((n_particles - first) < HCS_BLOCK ? (n_particles - first) : HCS_BLOCK);
        float* restrict p = pos + first * 3;
        float* restrict v = vel + first * 3;
        float* restrict a = age + first;
        const float* restrict l = life + first;
        uint8_t* restrict al = alive + first;
        uint32_t n_alive = 0;
        for (uint32_t i = 0; i < n; ++i) n_alive += al[i];
        if (n_alive == 0) continue;
        const int dense = n_alive == n; /* every slot of the block holds a live particle: flat loops over the planes */
        uint8_t still[HCS_BLOCK];
        memcpy(still, al, n);
        for (uint32_t k = 0; k < n_ops; ++k) {
            const HcsOp o = ops[k];
            switch (o.op) {
                case HCS_AGE_TICK: {
                    const float dt = o.v[0];
                    if (dense) {
                        for (uint32_t i = 0; i < n; ++i) { const float x = a[i] + dt; a[i] = x; still[i] = (uint8_t)(x < l[i]); }
                    } else {
                        for (uint32_t i = 0; i < n; ++i)
                            if (al[i]) { const float x = a[i] + dt; a[i] = x; still[i] = (uint8_t)(x < l[i]); }
                    }
                } break;
                case HCS_VEL_SCALE: {
                    const float s = o.v[0];
                    if (dense) { for (uint32_t i = 0; i < 3 * n; ++i) v[i] *= s; }
                    else { for (uint32_t i = 0; i < n; ++i) if (al[i]) { v[3 * i] *= s; v[3 * i + 1] *= s; v[3 * i + 2] *= s; } }
                } break;
                case HCS_VEL_ADD: {
                    if (dense) {
                        const float pat[12] = {o.v[0], o.v[1], o.v[2], o.v[0], o.v[1], o.v[2], o.v[0], o.v[1], o.v[2], o.v[0], o.v[1], o.v[2]};
                        uint32_t q = 0;
                        for (; q + 4 <= n; q += 4)
                            for (uint32_t j = 0; j < 12; ++j) v[3 * q + j] += pat[j];
                        for (; q < n; ++q) { v[3 * q] += o.v[0]; v[3 * q + 1] += o.v[1]; v[3 * q + 2] += o.v[2]; }
                    } else {
                        for (uint32_t i = 0; i < n; ++i) if (al[i]) { v[3 * i] += o.v[0]; v[3 * i + 1] += o.v[1]; v[3 * i + 2] += o.v[2]; }
                    }
                } break;
                case HCS_EULER: {
                    const float dt = o.v[0];
                    if (dense) { for (uint32_t i = 0; i < 3 * n; ++i) p[i] += v[i] * dt; }
                    else { for (uint32_t i = 0; i < n; ++i) if (al[i]) { p[3 * i] += v[3 * i] * dt; p[3 * i + 1] += v[3 * i + 1] * dt; p[3 * i + 2] += v[3 * i + 2] * dt; } }
                } break;
                default: break;
            }
        }
        uint32_t died = 0;
        for (uint32_t i = 0; i < n; ++i) died += (uint32_t)(al[i] && !still[i]);
        if (died) {
            for (uint32_t i = 0; i < n; ++i) if (al[i] && !still[i]) al[i] = 0;
            died_total += died;
        }
    }
    return died_total;
}
