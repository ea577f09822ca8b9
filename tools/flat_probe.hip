// Access-pattern probe (development tool, not part of the product): the firework update of the position and velocity planes
// (vel = vel * drag + a; pos += vel * dt; ages kept per chunk as in the product's cohort mode) over 16,777,216 particles, in place,
// walking the chunks in alternating directions, with two ways of mapping lanes to the 48 KiB a chunk has in each plane:
//   Q  the product's: a lane owns 4 consecutive particles = 48 contiguous bytes = three 16-byte accesses at a 48-byte lane stride
//      (every load instruction of a wave touches 24 cache lines for 1 KiB of payload; the other two instructions reuse them)
//   F  flat: every instruction of a wave covers 1 KiB contiguous (lane l takes 16-byte word l of it); legal for programs that are
//      component-wise (VEL_SCALE, VEL_ADD, EULER): word k of the plane is component k mod 3, no particle has to be reassembled
// Question: torch's elementwise kernels move 5.95 TB/s in place on these boxes (tools/stream_ceilings.py); the product's update 5.0.
// Usage: flat_probe [capacity] [allocations] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct P { float dt, drag, a[3]; };

__global__ void __launch_bounds__(256, 6) k_quad(char* base, uint32_t cap, const P k, uint32_t reverse) {
    const uint32_t chunk = reverse ? gridDim.x - 1u - blockIdx.x : blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    char* pp = base + (size_t)chunk * 49152;
    char* vp = base + (size_t)cap * 12 + (size_t)chunk * 49152;
#pragma unroll
    for (uint32_t step = 0; step < 4; ++step) {
        const uint32_t q = wave * 256 + step * 64 + lane;  // quad within the chunk
        f4* p = reinterpret_cast<f4*>(pp) + (size_t)q * 3;
        f4* v = reinterpret_cast<f4*>(vp) + (size_t)q * 3;
        f4 pv[3] = {p[0], p[1], p[2]}, vv[3] = {v[0], v[1], v[2]};
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int c = (4 * j + e) % 3; vv[j][e] = vv[j][e] * k.drag; vv[j][e] = vv[j][e] + k.a[c]; pv[j][e] = pv[j][e] + vv[j][e] * k.dt; }
        p[0] = pv[0]; p[1] = pv[1]; p[2] = pv[2]; v[0] = vv[0]; v[1] = vv[1]; v[2] = vv[2];
    }
}

template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_flat(char* base, uint32_t cap, const P k, uint32_t reverse) {
    const uint32_t chunk = reverse ? gridDim.x - 1u - blockIdx.x : blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    f4* pp = reinterpret_cast<f4*>(base + (size_t)chunk * 49152) + wave * 768;
    f4* vp = reinterpret_cast<f4*>(base + (size_t)cap * 12 + (size_t)chunk * 49152) + wave * 768;
    // word index within the chunk = wave * 768 + j * 64 + lane; its first float is component (4 * index) mod 3 = (wave * 768 + j * 64 + lane) mod 3
    // = (j + lane) mod 3 (768 = 0, 64 = 1 mod 3): rotate the acceleration once per lane, then the component is static per (j, e)
    const uint32_t r = lane % 3u;
    const float a0 = r == 0 ? k.a[0] : r == 1 ? k.a[1] : k.a[2], a1 = r == 0 ? k.a[1] : r == 1 ? k.a[2] : k.a[0], a2 = r == 0 ? k.a[2] : r == 1 ? k.a[0] : k.a[1];
    const float ar[3] = {a0, a1, a2};
#pragma unroll
    for (uint32_t step = 0; step < 4; ++step) {
        f4 pv[3], vv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { pv[j] = pp[(step * 3 + j) * 64 + lane]; vv[j] = vp[(step * 3 + j) * 64 + lane]; }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int c = ((step * 3 + j) + e) % 3; vv[j][e] = vv[j][e] * k.drag; vv[j][e] = vv[j][e] + ar[c]; pv[j][e] = pv[j][e] + vv[j][e] * k.dt; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { pp[(step * 3 + j) * 64 + lane] = pv[j]; vp[(step * 3 + j) * 64 + lane] = vv[j]; }
    }
}

template <class F> float time_ms(int iters, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(0); f(1); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) f(i);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms / iters);
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main(int argc, char** argv) {
    const uint32_t cap = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 24);
    const int n_alloc = argc > 2 ? atoi(argv[2]) : 6;
    const int iters = argc > 3 ? atoi(argv[3]) : 30;
    const uint32_t chunks = cap / 4096;
    const P k{1.0f / 60, 0.93f, {0.0f, -0.26f, 0.0f}};
    const size_t bytes = (size_t)cap * 24;
    printf("capacity %u, %d allocations of %zu MB, %d iterations (alternating direction unless noted), best of 3; MB per launch %.1f\n", cap, n_alloc, bytes >> 20, iters, bytes * 2 / 1e6);
    std::vector<char*> slabs;
    for (int i = 0; i < n_alloc; ++i) { char* s; CK(hipMalloc(&s, bytes + (1 << 20))); CK(hipMemset(s, 0, bytes)); slabs.push_back(s); }
    for (int v = 0; v < 5; ++v) {
        const char* names[5] = {"Q quads, 48-byte lane stride (product)", "F flat, 6 waves/SIMD budget", "F flat, 8 waves/SIMD budget", "Q quads, one direction", "F flat (8), one direction"};
        printf("%-42s", names[v]);
        for (char* s : slabs) {
            float ms = time_ms(iters, [&](int i) {
                const uint32_t rev = (v >= 3) ? 0u : (uint32_t)(i & 1);
                if (v == 0 || v == 3) k_quad<<<chunks, 256>>>(s, cap, k, rev);
                else if (v == 1) k_flat<6><<<chunks, 256>>>(s, cap, k, rev);
                else k_flat<8><<<chunks, 256>>>(s, cap, k, rev);
            });
            printf(" %.4f", ms);
        }
        printf("  ms\n");
    }
    return 0;
}
