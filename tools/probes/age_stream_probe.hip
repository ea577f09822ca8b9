// Round 6 probe: what does the access pattern of C5's update (4,194,304 slots: alive bytes + ages in, ages out, 1024 workgroups x 256 threads, a chain of
// dependent launches) cost on its own, and which of the product kernel's extras (instance row behind a pointer, wave reductions + barrier + epilogue,
// plain instead of nontemporal stores, died-bit words) adds what?
//   hipcc --offload-arch=gfx950 -O3 age_stream_probe.hip -o age_stream_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
struct Args { char* base; const uint64_t* inst_base; uint32_t age_off, flag_off, died_off, out_off; float dt, lm; };
// MODE bits: 1 = base behind inst_base[0] (a dependent scalar load), 2 = plain stores, 4 = wave reductions + barrier + epilogue, 8 = died-bit words, 16 = no stores at all
template <int MODE>
__global__ void __launch_bounds__(256) k(const Args a) {
    __shared__ uint32_t s_a[4]; __shared__ float s_m[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    char* base = (MODE & 1) ? reinterpret_cast<char*>(a.inst_base[0]) : a.base;
    const uint32_t wave0 = blockIdx.x * 4096u + wave * 1024u;
    const uint32_t* flags4 = reinterpret_cast<const uint32_t*>(base + a.flag_off);
    u4v* age = reinterpret_cast<u4v*>(base + a.age_off);
    uint32_t f[4]; u4v g[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { const uint32_t q = (wave0 + s * 256u + lane * 4u) >> 2; f[s] = __builtin_nontemporal_load(flags4 + q); g[s] = age[q]; }
    uint32_t alive = 0; float mn = 1e30f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint32_t q = (wave0 + s * 256u + lane * 4u) >> 2;
        u4v o;
        float x;
        x = __uint_as_float(g[s].x) + a.dt; o.x = (f[s] & 0xffu) ? __float_as_uint(x) : g[s].x; mn = fminf(mn, a.lm - x);
        x = __uint_as_float(g[s].y) + a.dt; o.y = (f[s] & 0xff00u) ? __float_as_uint(x) : g[s].y; mn = fminf(mn, a.lm - x);
        x = __uint_as_float(g[s].z) + a.dt; o.z = (f[s] & 0xff0000u) ? __float_as_uint(x) : g[s].z; mn = fminf(mn, a.lm - x);
        x = __uint_as_float(g[s].w) + a.dt; o.w = (f[s] & 0xff000000u) ? __float_as_uint(x) : g[s].w; mn = fminf(mn, a.lm - x);
        alive += __popc(f[s] & 0x01010101u);
        if (!(MODE & 16)) { if (MODE & 2) age[q] = o; else __builtin_nontemporal_store(o, age + q); }
        if ((MODE & 8) && (lane & 7u) == 0u) reinterpret_cast<uint32_t*>(base + a.died_off)[((wave0 + s * 256u) >> 5) + (lane >> 3)] = 0u;
    }
    if (MODE & 4) {
#pragma unroll
        for (uint32_t off = 32; off > 0; off >>= 1) { alive += __shfl_xor(alive, off, 64); mn = fminf(mn, __shfl_xor(mn, off, 64)); }
        if (lane == 0) { s_a[wave] = alive; s_m[wave] = mn; }
        __syncthreads();
        if (tid == 0) {
            uint32_t* out = reinterpret_cast<uint32_t*>(base + a.out_off);
            out[blockIdx.x] = s_a[0] + s_a[1] + s_a[2] + s_a[3];
            out[4096 + blockIdx.x] = __float_as_uint(fminf(fminf(s_m[0], s_m[1]), fminf(s_m[2], s_m[3])));
        }
    } else if (alive == 0xffffffffu || mn == 123.0f) reinterpret_cast<uint32_t*>(base + a.out_off)[0] = 1u;
}
__global__ void k_small(uint32_t* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1u; }   // a stand-in for the frame's other small launches
template <int MODE> void run(const char* what, const Args& a, hipStream_t st, int between) {
    uint32_t* scratch = reinterpret_cast<uint32_t*>(a.base + a.out_off + 65536);
    for (int i = 0; i < 200; ++i) k<MODE><<<1024, 256, 0, st>>>(a);
    (void)hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) { k<MODE><<<1024, 256, 0, st>>>(a); for (int b = 0; b < between; ++b) k_small<<<64, 64, 0, st>>>(scratch); }
    (void)hipStreamSynchronize(st);
    printf("%-70s %.2f us per frame (%d small launches between)\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000.0, between);
}
int main() {
    const uint32_t n = 1u << 22;
    char* base; (void)hipMalloc(&base, (size_t)n * 4 + n + n / 8 + (1 << 20));
    (void)hipMemset(base, 0, (size_t)n * 4 + n + n / 8 + (1 << 20));
    Args a; a.base = base; a.age_off = 0; a.flag_off = n * 4; a.died_off = n * 4 + n; a.out_off = n * 4 + n + n / 8; a.dt = 1.0f / 60.0f; a.lm = 1e9f;
    (void)hipMemset(base + a.flag_off, 1, n);
    uint64_t* ib; (void)hipMalloc(&ib, 8); uint64_t v = (uint64_t)base; (void)hipMemcpy(ib, &v, 8, hipMemcpyHostToDevice); a.inst_base = ib;
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) for (int between = 0; between <= 2; between += 2) {
        run<16>("loads only", a, st, between);
        run<0>("stream: alive bytes + ages in, ages out (nontemporal)", a, st, between);
        run<2>("... plain stores", a, st, between);
        run<1>("... base behind a pointer", a, st, between);
        run<4>("... wave reductions + barrier + epilogue", a, st, between);
        run<8>("... died-bit words", a, st, between);
        run<1 | 4 | 8>("... pointer + reductions + died bits (nontemporal)", a, st, between);
        run<1 | 2 | 4 | 8>("... pointer + reductions + died bits (plain stores)", a, st, between);
    }
    return 0;
}
