// Round 6 probe: does kernel-argument PRELOAD (-mllvm -amdgpu-kernarg-preload-count=N: the first dwords of the kernarg segment arrive in SGPRs with the
// wave, no s_load round trip in front of the first dependent load) shorten a chain of small dependent launches on this GPU / firmware?
// Three dependent launches per "frame", each workgroup: pointer args -> a table entry -> a base pointer -> data (the shape of the product's small kernels).
//   hipcc --offload-arch=gfx950 -O3 kernarg_preload_probe.hip -o probe_off ; ... -mllvm -amdgpu-kernarg-preload-count=8 -o probe_on
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Pad { unsigned w[48]; };
__global__ void __launch_bounds__(256) k(const unsigned long long* __restrict__ table, const unsigned* __restrict__ fi, unsigned n, const Pad pad) {
    const unsigned k = blockIdx.x & 3u;
    if (fi[k * 8] == 12345u) return;
    unsigned* base = reinterpret_cast<unsigned*>(table[k]);
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) base[i] = base[i] + pad.w[k];
}
int main() {
    const unsigned n = 1u << 20;
    unsigned* data; unsigned long long* table; unsigned* fi;
    hipMalloc(&data, n * 4); hipMemset(data, 0, n * 4);
    hipMalloc(&table, 64); hipMalloc(&fi, 256); hipMemset(fi, 0, 256);
    unsigned long long h[4] = {(unsigned long long)data, (unsigned long long)data, (unsigned long long)data, (unsigned long long)data};
    hipMemcpy(table, h, 32, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    Pad pad{};
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 300; ++i) k<<<n / 256, 256, 0, st>>>(table, fi, n, pad);
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 3000; ++i) k<<<n / 256, 256, 0, st>>>(table, fi, n, pad);
        hipStreamSynchronize(st);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 3000.0;
        printf("%.2f us per dependent launch (4096 workgroups, 4 MB)\n", us);
    }
    return 0;
}
