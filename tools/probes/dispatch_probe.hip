// Round 6 probe: how long does a chain of dependent launches of 1024 workgroups x 256 threads take as a function of what a workgroup RESERVES
// (static LDS, VGPRs) - i.e. is the dispatch of C5's update (24.6 KB of LDS it never uses, 99 VGPRs) part of its 19 us?
//   hipcc --offload-arch=gfx950 -O3 dispatch_probe.hip -o dispatch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int LDS_WORDS, int REGS>
__global__ void __launch_bounds__(256) k(const unsigned* __restrict__ in, unsigned* __restrict__ out, unsigned n, unsigned salt) {
    __shared__ unsigned lds[LDS_WORDS > 0 ? LDS_WORDS : 1];
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    unsigned r[REGS];
#pragma unroll
    for (int q = 0; q < REGS; ++q) r[q] = in[(i + q * 4096u) % n] ^ salt;   // REGS independent loads in flight: REGS live VGPRs
    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < REGS; ++q) acc += r[q] * (q + 1);
    if (LDS_WORDS > 0 && salt == 0x12345u) { lds[threadIdx.x % LDS_WORDS] = acc; __syncthreads(); acc += lds[(threadIdx.x + 1) % LDS_WORDS]; }
    if (i < n) out[i] = acc;
}
template <int L, int R> void run(const char* what, unsigned* in, unsigned* out, unsigned n, hipStream_t st) {
    for (int i = 0; i < 200; ++i) k<L, R><<<1024, 256, 0, st>>>(in, out, n, 0);
    (void)hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) k<L, R><<<1024, 256, 0, st>>>(in, out, n, 0);
    (void)hipStreamSynchronize(st);
    printf("%-40s %.2f us per dependent launch of 1024 x 256\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000.0);
}
int main() {
    const unsigned n = 1u << 22;
    unsigned *in, *out; (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, n * 4); (void)hipMemset(in, 1, n * 4);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 4>("no LDS, 4 loads per thread", in, out, n, st);
        run<6168, 4>("24.6 KB LDS, 4 loads per thread", in, out, n, st);
        run<0, 32>("no LDS, 32 loads per thread", in, out, n, st);
        run<6168, 32>("24.6 KB LDS, 32 loads per thread", in, out, n, st);
        run<0, 80>("no LDS, 80 loads per thread", in, out, n, st);
        run<6168, 80>("24.6 KB LDS, 80 loads per thread", in, out, n, st);
    }
    return 0;
}
