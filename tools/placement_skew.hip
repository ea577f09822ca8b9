// Development probe: does the update's streaming time depend on the physical placement of a slab, and can an
// inter-plane skew repair a slow placement?  hipcc --offload-arch=gfx950 -O3 tools/placement_skew.hip -o tools/placement_skew
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct ProbeArgs { uint32_t n_planes; uint32_t stride16[8]; uint64_t off[8]; uint32_t write_mask; uint32_t salt; uint64_t n_quads; };
__global__ void __launch_bounds__(256) k_probe(char* __restrict__ base, const ProbeArgs a) {
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= a.n_quads) return;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t pl = 0; pl < 8; ++pl) {
        if (pl >= a.n_planes) break;
        uint4* ptr = reinterpret_cast<uint4*>(base + a.off[pl]) + q * a.stride16[pl];
        for (uint32_t i = 0; i < a.stride16[pl]; ++i) {
            uint4 v = ptr[i];
            acc += v.x + v.y + v.z + v.w;
            v.x ^= a.salt;
            if (a.write_mask >> pl & 1u) ptr[i] = v;
        }
    }
    if (acc == 0x9e3779b9u && a.salt) *reinterpret_cast<uint32_t*>(base) = acc;
}
int main(int argc, char** argv) {
    const int n_cand = argc > 1 ? atoi(argv[1]) : 12;
    const bool use_vmm = argc > 2 && atoi(argv[2]) == 1;
    const uint64_t cap = 1ull << 24;
    const uint64_t MiB = 1ull << 20;
    const size_t bytes = (argc > 3 ? (size_t)atoi(argv[3]) : 960) * MiB;
    const uint64_t skews[] = {0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<char*> bufs;
    printf("candidate:   "); for (uint64_t s : skews) printf(" skew=%-8llu", (unsigned long long)s); printf("\n");
    for (int c = 0; c < n_cand; ++c) {
        char* b = nullptr;
        if (use_vmm) {  // one physical allocation mapped into a reserved range
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0; hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
            const size_t sz = (bytes + gran - 1) / gran * gran;
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); break; }
            void* va = nullptr;
            if (hipMemAddressReserve(&va, sz, (size_t)1 << 30, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); break; }
            if (hipMemMap(va, sz, 0, h, 0) != hipSuccess) { printf("map failed\n"); break; }
            hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            if (hipMemSetAccess(va, sz, &ad, 1) != hipSuccess) { printf("setaccess failed\n"); break; }
            b = (char*)va;
            if (c == 0) printf("(vmm granularity %zu)\n", gran);
        } else {
            if (hipMalloc(&b, bytes) != hipSuccess) break;
            bufs.push_back(b);
        }
        hipMemset(b, 0, bytes);
        printf("%2d %p:", c, (void*)b);
        for (uint64_t s : skews) {
            ProbeArgs a{};
            a.n_planes = 4; a.n_quads = cap / 4; a.write_mask = 0x7;  // pos, vel, age written; lifetime read
            const uint64_t base_off = 192 * MiB;                       // after the three list columns
            a.off[0] = base_off;                 a.stride16[0] = 3;
            a.off[1] = base_off + 192 * MiB + s; a.stride16[1] = 3;
            a.off[2] = base_off + 384 * MiB + 2 * s; a.stride16[2] = 1;
            a.off[3] = base_off + 448 * MiB + 3 * s; a.stride16[3] = 1;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0, 0);
                k_probe<<<(uint32_t)((a.n_quads + 255) / 256), 256>>>(b, a);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (rep && t < best) best = t;
            }
            printf("  %.4f      ", best);
        }
        printf("\n"); fflush(stdout);
    }
    for (char* b : bufs) hipFree(b);
    return 0;
}
