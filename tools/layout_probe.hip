// Layout probe (development tool, not part of the product): the firework update (age += dt; vel = vel*drag + a;
// pos += vel*dt) over 16,777,216 particles on three storage layouts, each over several fresh allocations, to answer
// two questions of VERDICT r01: (4) what would a 32-byte particle record cost the steady update, and (5) does a
// chunk-interleaved layout (one contiguous region per 4096-slot chunk instead of planes hundreds of MB apart) remove the
// allocation-placement classes seen with plane-major SoA.
//   A  plane-major SoA (the product's layout): pos[cap] vel[cap] age[cap] life[cap], 16-byte accesses, 56 B / particle
//   B  chunk-interleaved SoA: per 4096 slots [pos 48K][vel 48K][age 16K][life 16K], same accesses, 56 B / particle
//   C  32-byte records {pos.xyz, vel.x | vel.yz, age, life}: 2 x 16 B per particle, strided; 64 B / particle
//   C' the same records through a wave-local transpose (every load instruction covers 1 KiB contiguous)
// Usage: layout_probe [capacity] [allocations] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

struct P { float dt, drag, ax, ay, az; };

__device__ __forceinline__ void upd12(f4& p0, f4& p1, f4& p2, f4& v0, f4& v1, f4& v2, const P k) {
    float Pp[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
    float V[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
    const float a[3] = {k.ax, k.ay, k.az};
#pragma unroll
    for (int i = 0; i < 12; ++i) { V[i] = V[i] * k.drag; V[i] = V[i] + a[i % 3]; Pp[i] = Pp[i] + V[i] * k.dt; }
    p0 = f4{Pp[0], Pp[1], Pp[2], Pp[3]}; p1 = f4{Pp[4], Pp[5], Pp[6], Pp[7]}; p2 = f4{Pp[8], Pp[9], Pp[10], Pp[11]};
    v0 = f4{V[0], V[1], V[2], V[3]}; v1 = f4{V[4], V[5], V[6], V[7]}; v2 = f4{V[8], V[9], V[10], V[11]};
}

// one workgroup per 4096-slot chunk, a wave owns 1024 slots in 4 steps, a lane 4 consecutive slots (as the product)
template <bool INTERLEAVED>
__global__ void __launch_bounds__(256, 6) k_soa(char* base, uint32_t cap, const P k, uint32_t* sink, size_t pos_off = 0, size_t vel_off = 0, size_t age_off = 0, uint32_t reverse = 0) {
    const uint32_t chunk = reverse ? gridDim.x - 1u - blockIdx.x : blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    char *pp, *vp, *ap;
    if (INTERLEAVED) { char* c = base + (size_t)chunk * 131072; pp = c; vp = c + 49152; ap = c + 98304; }
    else if (vel_off) { pp = base + pos_off + (size_t)chunk * 49152; vp = base + vel_off + (size_t)chunk * 49152; ap = base + age_off + (size_t)chunk * 16384; }
    else { pp = base + (size_t)chunk * 49152; vp = base + (size_t)cap * 12 + (size_t)chunk * 49152; ap = base + (size_t)cap * 24 + (size_t)chunk * 16384; }
    uint32_t died = 0;
#pragma unroll
    for (uint32_t step = 0; step < 4; ++step) {
        const uint32_t q = wave * 256 + step * 64 + lane;  // quad within the chunk
        f4* p = reinterpret_cast<f4*>(pp) + (size_t)q * 3;
        f4* v = reinterpret_cast<f4*>(vp) + (size_t)q * 3;
        f4* a = reinterpret_cast<f4*>(ap) + q;
        f4 p0 = p[0], p1 = p[1], p2 = p[2], v0 = v[0], v1 = v[1], v2 = v[2], ag = a[0];
        upd12(p0, p1, p2, v0, v1, v2, k);
        ag += k.dt;
        died += (ag.x > 1e30f) + (ag.y > 1e30f) + (ag.z > 1e30f) + (ag.w > 1e30f);
        p[0] = p0; p[1] = p1; p[2] = p2; v[0] = v0; v[1] = v1; v[2] = v2; a[0] = ag;
    }
    if (died) atomicAdd(sink, died);
}

// 32-byte records, one particle per lane per step: two 16-byte accesses at a 32-byte lane stride
__global__ void __launch_bounds__(256, 8) k_rec(char* base, uint32_t cap, const P k, uint32_t* sink) {
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    f4* rec = reinterpret_cast<f4*>(base + (size_t)chunk * 131072);
    uint32_t died = 0;
#pragma unroll 4
    for (uint32_t step = 0; step < 16; ++step) {
        const uint32_t s = step * 256 + tid;
        f4 r0 = rec[2 * s], r1 = rec[2 * s + 1];   // {px py pz vx} {vy vz age life}
        float vx = r0.w * k.drag + k.ax, vy = r1.x * k.drag + k.ay, vz = r1.y * k.drag + k.az;
        r0.x += vx * k.dt; r0.y += vy * k.dt; r0.z += vz * k.dt; r0.w = vx; r1.x = vy; r1.y = vz;
        r1.z += k.dt;
        died += !(r1.z < r1.w);
        rec[2 * s] = r0; rec[2 * s + 1] = r1;
    }
    if (died) atomicAdd(sink, died);
}

// 32-byte records, wave-local transpose: per step a wave covers 128 records = 4 KiB with four load instructions of
// 1 KiB contiguous each; lane pair (2j, 2j+1) exchanges halves so that each lane ends up with two whole records
__global__ void __launch_bounds__(256, 8) k_rec_t(char* base, uint32_t cap, const P k, uint32_t* sink) {
    const uint32_t chunk = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    f4* rec = reinterpret_cast<f4*>(base + (size_t)chunk * 131072) + (size_t)wave * 2048;  // 1024 records = 2048 f4 per wave
    uint32_t died = 0;
    const bool odd = lane & 1u;
#pragma unroll 2
    for (uint32_t step = 0; step < 16; ++step) {
        f4* blk = rec + step * 128;            // 64 records
        f4 x = blk[lane], y = blk[64 + lane];  // x: record lane/2 half lane%2; y: record 32 + lane/2, half lane%2
        // even lane keeps record lane/2 (needs neighbour's x), odd lane keeps record 32 + lane/2 (needs neighbour's y)
        f4 give = odd ? x : y, got;
        got.x = __shfl_xor(give.x, 1, 64); got.y = __shfl_xor(give.y, 1, 64); got.z = __shfl_xor(give.z, 1, 64); got.w = __shfl_xor(give.w, 1, 64);
        f4 r0 = odd ? got : x, r1 = odd ? y : got;
        float vx = r0.w * k.drag + k.ax, vy = r1.x * k.drag + k.ay, vz = r1.y * k.drag + k.az;
        r0.x += vx * k.dt; r0.y += vy * k.dt; r0.z += vz * k.dt; r0.w = vx; r1.x = vy; r1.y = vz;
        r1.z += k.dt;
        died += !(r1.z < r1.w);
        give = odd ? r0 : r1;
        got.x = __shfl_xor(give.x, 1, 64); got.y = __shfl_xor(give.y, 1, 64); got.z = __shfl_xor(give.z, 1, 64); got.w = __shfl_xor(give.w, 1, 64);
        blk[lane] = odd ? got : r0;
        blk[64 + lane] = odd ? r1 : got;
    }
    if (died) atomicAdd(sink, died);
}

__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) dst[i] = src[i];
}

template <class F> float time_ms(int iters, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms / iters);
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main(int argc, char** argv) {
    const uint32_t cap = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 24);
    const int n_alloc = argc > 2 ? atoi(argv[2]) : 8;
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    const uint32_t chunks = cap / 4096;
    const P k{1.0f / 60, 0.93f, 0.0f, -0.26f, 0.0f};
    uint32_t* sink; CK(hipMalloc(&sink, 4)); CK(hipMemset(sink, 0, 4));
    const size_t bytes = (size_t)cap * 32;  // every layout holds pos + vel + age + life = 32 B per particle
    // the product's slab also carries three list columns in front and other planes behind: the planes of layout A sit at the
    // same distances from each other as in the product (12 B * cap, 12 B * cap, 4 B * cap)
    printf("capacity %u, %d allocations of %zu MB each, %d iterations, best of 3\n", cap, n_alloc, bytes >> 20, iters);
    std::vector<char*> slabs;
    for (int i = 0; i < n_alloc; ++i) { char* s; CK(hipMalloc(&s, bytes + (1 << 20))); CK(hipMemset(s, 0, bytes)); slabs.push_back(s); }
    const char* names[4] = {"A plane-major SoA (56 B)", "B chunk-interleaved SoA (56 B)", "C 32-byte records, strided (64 B)", "C' 32-byte records, transposed (64 B)"};
    for (int v = 0; v < 4; ++v) {
        printf("%-40s", names[v]);
        std::vector<float> t;
        for (char* s : slabs) {
            // life plane / field: huge, so nothing dies
            std::vector<float> init((size_t)cap * 8, 0.0f);
            if (v == 0) for (size_t i = 0; i < cap; ++i) init[(size_t)cap * 7 + i] = 1e30f;
            else if (v == 1) for (size_t c = 0; c < chunks; ++c) for (size_t i = 0; i < 4096; ++i) init[c * 32768 + 28672 + i] = 1e30f;
            else for (size_t i = 0; i < cap; ++i) init[i * 8 + 7] = 1e30f;
            CK(hipMemcpy(s, init.data(), bytes, hipMemcpyHostToDevice));
            float ms = 0;
            if (v == 0) ms = time_ms(iters, [&] { k_soa<false><<<chunks, 256>>>(s, cap, k, sink); });
            if (v == 1) ms = time_ms(iters, [&] { k_soa<true><<<chunks, 256>>>(s, cap, k, sink); });
            if (v == 2) ms = time_ms(iters, [&] { k_rec<<<chunks, 256>>>(s, cap, k, sink); });
            if (v == 3) ms = time_ms(iters, [&] { k_rec_t<<<chunks, 256>>>(s, cap, k, sink); });
            t.push_back(ms);
            printf(" %.4f", ms);
        }
        const float mn = *std::min_element(t.begin(), t.end()), mx = *std::max_element(t.begin(), t.end());
        const double bpp = v < 2 ? 56.0 : 64.0;
        printf("  | min %.4f max %.4f ms -> %.0f .. %.0f GB/s physical\n", mn, mx, cap * bpp / mx / 1e6, cap * bpp / mn / 1e6);
    }
    {
        printf("%-40s", "float4 copy, half of each slab to the other half");
        for (char* s : slabs) {
            const size_t n16 = bytes / 32;
            float ms = time_ms(iters, [&] { k_copy16<<<2048, 256>>>((const uint4*)s, (uint4*)(s + bytes / 2), n16); });
            printf(" %.0f", (double)n16 * 32 / ms / 1e6);
        }
        printf(" GB/s (r+w)\n");
    }
    for (char* s : slabs) CK(hipFree(s));
    slabs.clear();
    // The product's slab: [list 0][list 1][dead list][pos][vel][age][life][color][alive bytes]... = 49 B per slot and more.
    // Same kernel, the product's plane offsets, allocations of the product's size — and of that size rounded up to 1 GiB / 2 GiB.
    const size_t al = 256;
    auto up = [&](size_t v) { return (v + al - 1) / al * al; };
    const size_t lists = 3 * up((size_t)cap * 4);
    const size_t pos_off = lists, vel_off = pos_off + up((size_t)cap * 12), age_off = vel_off + up((size_t)cap * 12);
    const size_t product_bytes = age_off + 3 * up((size_t)cap * 4) + up(cap) + (1 << 20);
    const size_t sizes[3] = {product_bytes, ((product_bytes + (1ull << 30) - 1) >> 30) << 30, 2ull << 30};
    const char* snames[3] = {"product-shaped slab", "same, size rounded up to 1 GiB", "same, 2 GiB allocations"};
    for (int si = 0; si < 3; ++si) {
        std::vector<char*> sl;
        const int n = std::max(n_alloc, 12);
        for (int i = 0; i < n; ++i) { char* s; CK(hipMalloc(&s, sizes[si])); CK(hipMemset(s, 0, product_bytes)); sl.push_back(s); }
        printf("%-34s (%4zu MB) ", snames[si], sizes[si] >> 20);
        std::vector<float> t;
        for (char* s : sl) {
            float ms = time_ms(iters, [&] { k_soa<false><<<chunks, 256>>>(s, cap, k, sink, pos_off, vel_off, age_off); });
            t.push_back(ms);
            printf(" %.4f", ms);
        }
        printf("  | min %.4f max %.4f\n", *std::min_element(t.begin(), t.end()), *std::max_element(t.begin(), t.end()));
        printf("%-34s  VA (GiB):   ", "");
        for (char* s : sl) printf(" %.3f", (double)(uintptr_t)s / (double)(1ull << 30));
        printf("\n");
        for (char* s : sl) CK(hipFree(s));
    }
    {   // Infinity Cache (256 MiB, memory side) across frames: a frame that walks the chunks in the opposite direction of the
        // previous one starts on the data that was written last. Same kernel, product-shaped slabs.
        std::vector<char*> sl;
        for (int i = 0; i < 4; ++i) { char* s; CK(hipMalloc(&s, product_bytes)); CK(hipMemset(s, 0, product_bytes)); sl.push_back(s); }
        const char* mn[3] = {"ascending every frame", "descending every frame", "alternating direction"};
        for (int mode = 0; mode < 3; ++mode) {
            printf("%-34s            ", mn[mode]);
            for (char* s : sl) {
                uint32_t frame = 0;
                float ms = time_ms(iters, [&] { const uint32_t rev = mode == 0 ? 0u : mode == 1 ? 1u : (frame++ & 1u); k_soa<false><<<chunks, 256>>>(s, cap, k, sink, pos_off, vel_off, age_off, rev); });
                printf(" %.4f", ms);
            }
            printf("\n");
        }
        for (char* s : sl) CK(hipFree(s));
    }
    uint32_t d; CK(hipMemcpy(&d, sink, 4, hipMemcpyDeviceToHost));
    printf("sink %u\n", d);
    return 0;
}
