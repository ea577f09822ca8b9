/* One process, one HnbContext per GPU, one submit thread per context: the multi-GPU host the reference's single-process seam implies
 * (src/plugin.rs:202-256: one app, src/render/mod.rs:126-131: one `simulate` call site). Plain C99 + pthreads over include/hanabi_amd.h.
 *
 * A single giant effect is sharded by CAPACITY SLAB: context g owns global slots [g*C, (g+1)*C) and passes
 * slot_base = g*C, so the PRNG - seeded by the global slot index - draws what one GPU would and the union of the slabs equals the
 * one-GPU effect. No context ever waits for another during a frame; the only collective is the all-reduce of the alive counters over
 * RCCL at the end (hnb_comm_allreduce_alive). Each context has its own thread because hnb_simulate costs 20-30 us of host time per frame:
 * eight contexts submitted from one thread (>= 160 us) would outlast the 139 us frame of the 16.7M-particle firework.
 *
 *   gcc -std=c99 -O2 -pthread -Iinclude examples/multi_gpu.c -Lbevy_hanabi_amd -lhanabi_amd -Wl,-rpath,$PWD/bevy_hanabi_amd -o examples/multi_gpu
 *
 * Usage: multi_gpu <program.blob> <devices> <warmup> <steps> [windows [dt [dump_prefix]]]
 *   program.blob  a lowered program (examples/firework_c99 lower <capacity> out.blob, or hnb_lower through any binding); its capacity is
 *                 the capacity of ONE slab;
 *   devices       comma-separated HIP device ids, one context each ("0,1,2,3,4,5,6,7"; "0,0" puts two contexts on one GPU: a dry run);
 *   frame 0 bursts `capacity` particles into every slab, then `warmup` untimed frames, then `windows` (default 5) timed windows of
 *   `steps` frames, each between two thread barriers with every context synchronised; dt defaults to 1/60 s;
 *   dump_prefix   writes <prefix>.<g>.pos (capacity x 3 floats) of every slab after the last frame (parity checks).
 * Prints one JSON line. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hanabi_amd.h"

#define MAX_CTX 16
#define MAX_WINDOWS 32

static uint32_t pcg_hash(uint32_t x) {
    const uint32_t state = x * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}
static uint32_t frame_seed(uint32_t f) { return pcg_hash(0xC0FFEEu + f); } /* the harness-defined seed list of bench.py */
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

typedef struct {
    int index, device, n_ctx;
    const void* blob; size_t blob_size;
    uint32_t capacity, warmup, steps, windows;
    float dt;
    pthread_barrier_t* barrier;
    HnbContext* ctx; HnbProgram* prog; HnbEffect* fx;
    double window_s[MAX_WINDOWS];
    int rc; char err[256];
} Worker;

#define TRY(call)                                                                                                    \
    do {                                                                                                             \
        const int rc_ = (call);                                                                                      \
        if (rc_ != HNB_OK) { w->rc = rc_; snprintf(w->err, sizeof w->err, "%s: %s", #call, hnb_last_error()); goto done; } \
    } while (0)

static int step(Worker* w, uint32_t f) {
    HnbSimParams sim;
    int rc;
    sim.delta_time = sim.virtual_delta_time = sim.real_delta_time = w->dt;
    sim.time = sim.virtual_time = sim.real_time = (float)f * w->dt;
    if ((rc = hnb_frame_begin(w->ctx, &sim)) != HNB_OK) return rc;
    if ((rc = hnb_effect_set_frame(w->fx, f == 0 ? w->capacity : 0u, frame_seed(f), NULL)) != HNB_OK) return rc;
    return hnb_simulate(w->ctx);
}

static void* worker_main(void* arg) {
    Worker* w = (Worker*)arg;
    uint32_t f = 0, k, win;
    TRY(hnb_ctx_create(w->device, &w->ctx));
    TRY(hnb_program_create(w->ctx, w->blob, w->blob_size, &w->prog));
    TRY(hnb_effect_create(w->prog, (uint32_t)w->index * w->capacity, &w->fx));   /* slot_base: this slab's first global slot */
    for (k = 0; k < 1u + w->warmup; ++k) TRY(step(w, f++));
    TRY(hnb_ctx_synchronize(w->ctx));
done:
    /* every thread reaches every barrier, failed or not: nobody hangs */
    for (win = 0; win < w->windows; ++win) {
        double t0;
        pthread_barrier_wait(w->barrier);
        t0 = now_s();
        if (w->rc == HNB_OK) {
            for (k = 0; k < w->steps && w->rc == HNB_OK; ++k) w->rc = step(w, f++);
            if (w->rc == HNB_OK) w->rc = hnb_ctx_synchronize(w->ctx);
            if (w->rc != HNB_OK && !w->err[0]) snprintf(w->err, sizeof w->err, "frame %u: %s", f, hnb_last_error());
        }
        w->window_s[win] = now_s() - t0;
        pthread_barrier_wait(w->barrier);
    }
    return NULL;
}

static int cmp_double(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

int main(int argc, char** argv) {
    Worker workers[MAX_CTX];
    pthread_t threads[MAX_CTX];
    pthread_barrier_t barrier;
    HnbContext* ctxs[MAX_CTX];
    HnbEffect* effects[MAX_CTX];
    HnbComm* comm = NULL;
    HnbProgramHeader hdr;
    uint64_t alive_total = 0;
    double win_max[MAX_WINDOWS], sorted[MAX_WINDOWS], median;
    int n = 0, i, rc = 0;
    uint32_t win, windows, warmup, steps;
    float dt;
    void* blob;
    size_t blob_size;
    char* tok;
    FILE* f;
    if (argc < 5) { fprintf(stderr, "usage: %s <program.blob> <devices> <warmup> <steps> [windows [dt [dump_prefix]]]\n", argv[0]); return 2; }
    f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    fseek(f, 0, SEEK_END); blob_size = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    blob = malloc(blob_size);
    if (!blob || fread(blob, 1, blob_size, f) != blob_size || blob_size < sizeof hdr) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);
    if (hnb_program_validate(blob, blob_size) != HNB_OK) { fprintf(stderr, "bad program: %s\n", hnb_last_error()); return 1; }
    memcpy(&hdr, blob, sizeof hdr);
    warmup = (uint32_t)strtoul(argv[3], NULL, 10);
    steps = (uint32_t)strtoul(argv[4], NULL, 10);
    windows = argc > 5 ? (uint32_t)strtoul(argv[5], NULL, 10) : 5u;
    if (windows < 1u) windows = 1u;
    if (windows > MAX_WINDOWS) windows = MAX_WINDOWS;
    dt = argc > 6 ? (float)atof(argv[6]) : 1.0f / 60.0f;
    for (tok = strtok(argv[2], ","); tok && n < MAX_CTX; tok = strtok(NULL, ",")) {
        Worker* w = &workers[n];
        memset(w, 0, sizeof *w);
        w->index = n; w->device = atoi(tok);
        w->blob = blob; w->blob_size = blob_size; w->capacity = hdr.capacity;
        w->warmup = warmup; w->steps = steps; w->windows = windows; w->dt = dt; w->barrier = &barrier;
        ++n;
    }
    if (n == 0) { fprintf(stderr, "no devices\n"); return 2; }
    pthread_barrier_init(&barrier, NULL, (unsigned)n);
    for (i = 0; i < n; ++i) { workers[i].n_ctx = n; pthread_create(&threads[i], NULL, worker_main, &workers[i]); }
    for (i = 0; i < n; ++i) pthread_join(threads[i], NULL);
    for (i = 0; i < n; ++i)
        if (workers[i].rc != HNB_OK) { fprintf(stderr, "context %d (device %d) failed (%d): %s\n", i, workers[i].device, workers[i].rc, workers[i].err); rc = 1; }
    if (rc) return rc;
    /* the one collective: alive counters, summed over the contexts (RCCL when every context has its own GPU) */
    for (i = 0; i < n; ++i) { ctxs[i] = workers[i].ctx; effects[i] = workers[i].fx; }
    if (hnb_comm_create_local(ctxs, (uint32_t)n, &comm) != HNB_OK || hnb_comm_allreduce_alive(comm, effects, 1u, &alive_total) != HNB_OK) {
        fprintf(stderr, "alive all-reduce failed: %s\n", hnb_last_error());
        return 1;
    }
    for (win = 0; win < windows; ++win) {   /* a window lasts as long as its slowest context */
        win_max[win] = 0.0;
        for (i = 0; i < n; ++i) if (workers[i].window_s[win] > win_max[win]) win_max[win] = workers[i].window_s[win];
        sorted[win] = win_max[win];
    }
    qsort(sorted, windows, sizeof sorted[0], cmp_double);
    median = (windows & 1u) ? sorted[windows / 2] : 0.5 * (sorted[windows / 2 - 1] + sorted[windows / 2]);
    if (argc > 7) {
        const size_t bytes = (size_t)hdr.capacity * 12;
        float* pos = (float*)malloc(bytes);
        for (i = 0; i < n && pos; ++i) {
            char path[512];
            FILE* out;
            snprintf(path, sizeof path, "%s.%d.pos", argv[7], i);
            out = fopen(path, "wb");
            if (!out || hnb_effect_read_attr(workers[i].fx, HNB_ATTR_POSITION, pos, bytes) != HNB_OK) { fprintf(stderr, "cannot dump %s\n", path); return 1; }
            fwrite(pos, 1, bytes, out);
            fclose(out);
        }
        free(pos);
    }
    printf("{\"launcher\": \"threads\", \"n_ctx\": %d, \"devices\": [", n);
    for (i = 0; i < n; ++i) printf("%s%d", i ? ", " : "", workers[i].device);
    printf("], \"capacity_per_ctx\": %u, \"warmup\": %u, \"steps\": %u, \"windows\": %u, \"dt\": %.9g, \"window_ms_per_step\": [", hdr.capacity, warmup, steps, windows, (double)dt);
    for (win = 0; win < windows; ++win) printf("%s%.6f", win ? ", " : "", win_max[win] / steps * 1e3);
    printf("], \"ms_per_step\": %.6f, \"min_ms_per_step\": %.6f, \"alive_total\": %llu, \"updates_per_s\": %.6e}\n", median / steps * 1e3,
           sorted[0] / steps * 1e3, (unsigned long long)alive_total, (double)alive_total * steps / median);
    hnb_comm_destroy(comm);
    for (i = 0; i < n; ++i) { hnb_program_destroy(workers[i].prog); hnb_ctx_destroy(workers[i].ctx); }
    pthread_barrier_destroy(&barrier);
    free(blob);
    return 0;
}
