// Round 6 probe: how should a frame's small parameter block (a few hundred bytes the host knows only when the frame is submitted) reach the kernels?
//   A  pinned block -> hipMemcpyAsync on an upload stream -> hipStreamSynchronize (host waits) -> the frame's kernels        (the product until now)
//   B  the host WRITES the block into fine-grained device memory through the PCIe BAR (the way the runtime places kernel arguments), then launches
//   C  kernels read the pinned host block themselves (zero-copy)
//   D  like A without the host wait: the simulation stream waits for the copy's event
// Each frame = 3 dependent launches of 1024 x 256 that read the block and stream 16 MB each (about 5 us apiece on the device).
//   hipcc --offload-arch=gfx950 -O3 upload_probe.hip -o upload_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <immintrin.h>
struct Block { uint32_t w[96]; };   // 384 bytes
__global__ void __launch_bounds__(256) k(const Block* __restrict__ blk, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, uint32_t* err) {
    const uint32_t frame = blk->w[0], chk = blk->w[95];
    if (chk != (frame ^ 0xabcdu) && threadIdx.x == 0) atomicAdd(err, 1u);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = frame;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += in[(i + q * 262144u) % n];
    out[i] = acc;
}
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
int main() {
    const uint32_t n = 1u << 22;
    uint32_t *in, *out, *err; (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, n * 4); (void)hipMalloc(&err, 4); (void)hipMemset(in, 0, n * 4); (void)hipMemset(err, 0, 4);
    hipStream_t st, up; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
    Block* h_pin[4]; Block* d_blk[4];
    for (int i = 0; i < 4; ++i) { (void)hipHostMalloc(&h_pin[i], sizeof(Block), 0); (void)hipMalloc(&d_blk[i], sizeof(Block)); }
    Block* fine = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&fine), 4 * 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained): %s, ptr %p\n", hipGetErrorString(e), (void*)fine);
    bool bar_ok = false;
    if (e == hipSuccess && fine) {
        hipPointerAttribute_t at; if (hipPointerGetAttributes(&at, fine) == hipSuccess) printf("  type %d, device %d, hostPointer %p, devicePointer %p, isManaged %d\n", (int)at.type, at.device, at.hostPointer, at.devicePointer, at.isManaged);
        struct sigaction sa, old_segv, old_bus; memset(&sa, 0, sizeof sa); sa.sa_handler = on_segv; sigaction(SIGSEGV, &sa, &old_segv); sigaction(SIGBUS, &sa, &old_bus);
        if (sigsetjmp(jb, 1) == 0) { volatile uint32_t* p = reinterpret_cast<volatile uint32_t*>(fine); p[0] = 0x1234u; p[1] = p[0] + 1u; bar_ok = true; }
        sigaction(SIGSEGV, &old_segv, nullptr); sigaction(SIGBUS, &old_bus, nullptr);
        printf("  host write to it: %s\n", bar_ok ? "works" : "faults");
    }
    hipEvent_t ev[4]; for (int i = 0; i < 4; ++i) (void)hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    hipEvent_t done[4]; for (int i = 0; i < 4; ++i) (void)hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    auto frame_fill = [](Block* b, uint32_t f) { for (int i = 1; i < 95; ++i) b->w[i] = f + i; b->w[0] = f; b->w[95] = f ^ 0xabcdu; };
    for (int rep = 0; rep < 2; ++rep) for (int mode = 0; mode < 4; ++mode) {
        if (mode == 1 && !bar_ok) continue;
        const char* names[] = {"A  memcpyAsync + host wait", "B  host writes device memory (BAR)", "C  kernels read the pinned block", "D  memcpyAsync + stream waits for the event"};
        const int frames = 3000;
        std::chrono::steady_clock::time_point t0;
        for (int f = -300; f < frames; ++f) {
            if (f == 0) { (void)hipStreamSynchronize(st); t0 = std::chrono::steady_clock::now(); }
            const int slot = (f + 300) & 3;
            if (f + 300 >= 4) (void)hipEventSynchronize(done[slot]);   // the ring: the frame that last used the slot
            const Block* src = nullptr;
            if (mode == 0 || mode == 3) {
                frame_fill(h_pin[slot], (uint32_t)f);
                (void)hipMemcpyAsync(d_blk[slot], h_pin[slot], sizeof(Block), hipMemcpyHostToDevice, up);
                if (mode == 0) (void)hipStreamSynchronize(up);
                else { (void)hipEventRecord(ev[slot], up); (void)hipStreamWaitEvent(st, ev[slot], 0); }
                src = d_blk[slot];
            } else if (mode == 1) {
                Block tmp; frame_fill(&tmp, (uint32_t)f);
                Block* dst = reinterpret_cast<Block*>(reinterpret_cast<char*>(fine) + slot * 4096);
                memcpy(dst, &tmp, sizeof tmp);
                _mm_sfence();
                src = dst;
            } else { frame_fill(h_pin[slot], (uint32_t)f); src = h_pin[slot]; }
            for (int q = 0; q < 3; ++q) k<<<1024, 256, 0, st>>>(src, in, out, n, err);
            (void)hipEventRecord(done[slot], st);
        }
        (void)hipStreamSynchronize(st);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / frames;
        uint32_t herr = 0; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); (void)hipMemset(err, 0, 4);
        printf("%-50s %.2f us per frame of 3 dependent launches; stale / torn blocks seen: %u\n", names[mode], us, herr);
    }
    return 0;
}
