// Ribbon sort: after the update pass of an effect whose layout has RIBBON_ID, the alive list is
// reordered by (RIBBON_ID, AGE bits) ascending, stable with respect to the list order.
//
// Reference: vfx_sort_fill.wgsl (key = RIBBON_ID, key2 = AGE, value = particle index, in alive-list
// order), vfx_sort.wgsl (ONE thread running an insertion sort that only moves past strictly greater
// elements = stable), vfx_sort_copy.wgsl (values back into the same list column); scheduled after the
// update pass (src/render/mod.rs:4599-4618, 7372-7612).
//
// Here: a least-significant-digit radix sort on the 64-bit key (RIBBON_ID << 32 | AGE bits), 8 bits per
// pass, every instance of a program in the same launches:
//   k_sort_fill      keys / values from the compacted list + OR / AND of all keys of the instance;
//   per pass p       k_sort_hist (LDS histogram per 4096-item tile, plus per-group sums by atomics) ->
//                    k_sort_scatter (each tile derives its digit offsets itself from the group sums and the
//                    histograms of the earlier tiles of its group: no serial spine scan; stable ranks by
//                    wave match with ballots, per-round digit bases in LDS);
//   k_sort_copy      values of the final buffer back into the list column.
// A pass whose digit is identical in every key of the instance (known from OR ^ AND on the device,
// e.g. the whole RIBBON_ID half when there is one ribbon) returns immediately in all three kernels, and
// the ping-pong index is derived from the number of passes that did run, so nothing is read back.
#pragma once
#include "hnb_kernels.hip.h"

namespace hnb {

// Items per workgroup in the sort kernels (measured: 16384-item tiles leave the chip under-filled at 4M keys
// and are 30 % slower overall), and tiles per group of the two-level digit offsets.
constexpr uint32_t kSortTile = 4096;
constexpr uint32_t kSortGroup = 32;

struct SortArgs {
    uint32_t capacity, chunks_per_inst;   // chunks_per_inst = ceil(capacity / kSortTile) here
    uint32_t alive_off[2];
    uint32_t key_off[2];   // u64[capacity] ping-pong
    uint32_t val_off[2];   // u32[capacity] ping-pong
    uint32_t hist_off;     // u32[chunks_per_inst][256]: per-tile digit counts of the current pass
    uint32_t gsum_off;     // u32[8 passes][groups][256]: digit counts per group of kSortGroup tiles (zeroed by k_sort_fill)
    uint32_t bits_off;     // u64[2][2]: per frame parity {OR, AND} of the keys
    uint32_t rid_plane, age_plane;  // plane offsets (kNoPlane: key half is 0)
    uint32_t parity;       // frame parity of the bits double buffer
};

struct SortPass { bool active; uint32_t src; };
// Which ping-pong buffer pass p (0..7) reads, and whether it runs at all; p = 8 gives the buffer holding the result.
__device__ __forceinline__ SortPass sort_pass_info(const char* base, const SortArgs& a, uint32_t pass) {
    const uint64_t* bits = reinterpret_cast<const uint64_t*>(base + a.bits_off) + a.parity * 2u;
    const uint64_t varying = bits[0] ^ bits[1];
    uint32_t ran = 0;
    for (uint32_t q = 0; q < pass; ++q) ran += ((varying >> (8u * q)) & 0xffull) ? 1u : 0u;
    SortPass r;
    r.active = pass < 8u && ((varying >> (8u * (pass & 7u))) & 0xffull) != 0ull;
    r.src = ran & 1u;
    return r;
}

// n and list column after this frame's update + compaction
__device__ __forceinline__ void sort_setup(uint32_t chunk, const SortArgs& a, const uint64_t* inst_base, const DevMeta* meta, uint32_t& k, uint32_t& j,
                                           uint32_t& n, char*& base, uint32_t& column) {
    k = chunk / a.chunks_per_inst;
    j = chunk - k * a.chunks_per_inst;
    n = meta[k].alive_count;
    column = meta[k].write_index & 1u;
    base = reinterpret_cast<char*>(inst_base[k]);
}

__global__ void __launch_bounds__(kBlock)
k_sort_fill(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta) {
    __shared__ uint64_t s_or[kBlock / 64], s_and[kBlock / 64];
    uint32_t k, j, n, column; char* base;
    sort_setup(blockIdx.x, a, inst_base, meta, k, j, n, base, column);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint64_t* bits = reinterpret_cast<uint64_t*>(base + a.bits_off);
    if (j == 0 && tid == 0) { bits[(a.parity ^ 1u) * 2u] = 0ull; bits[(a.parity ^ 1u) * 2u + 1u] = ~0ull; }  // re-arm next frame's pair
    {   // clear the per-group digit sums of all 8 passes (each tile clears a slice)
        const uint32_t groups = (a.chunks_per_inst + kSortGroup - 1u) / kSortGroup;
        const uint32_t total = 8u * groups * 256u;
        uint32_t* gs = reinterpret_cast<uint32_t*>(base + a.gsum_off);
        for (uint32_t i = j * kBlock + tid; i < total; i += a.chunks_per_inst * kBlock) gs[i] = 0u;
    }
    const uint32_t* list = reinterpret_cast<const uint32_t*>(base + a.alive_off[column]);
    uint64_t* keys = reinterpret_cast<uint64_t*>(base + a.key_off[0]);
    uint32_t* vals = reinterpret_cast<uint32_t*>(base + a.val_off[0]);
    const uint32_t* rid = a.rid_plane == kNoPlane ? nullptr : reinterpret_cast<const uint32_t*>(base + a.rid_plane);
    const uint32_t* age = a.age_plane == kNoPlane ? nullptr : reinterpret_cast<const uint32_t*>(base + a.age_plane);
    uint64_t vor = 0ull, vand = ~0ull;
    for (uint32_t r = 0; r < kSortTile / kBlock; ++r) {
        const uint32_t i = j * kSortTile + r * kBlock + tid;
        if (i >= n) break;
        const uint32_t slot = list[i];
        const uint64_t key = ((uint64_t)(rid ? rid[slot] : 0u) << 32) | (uint64_t)(age ? age[slot] : 0u);
        keys[i] = key;
        vals[i] = slot;
        vor |= key;
        vand &= key;
    }
#pragma unroll
    for (uint32_t off = 32; off > 0; off >>= 1) {
        vor |= (uint64_t)__shfl_xor((unsigned long long)vor, off, 64);
        vand &= (uint64_t)__shfl_xor((unsigned long long)vand, off, 64);
    }
    if (lane == 0) { s_or[wave] = vor; s_and[wave] = vand; }
    __syncthreads();
    if (tid == 0 && j * kSortTile < n) {
        for (uint32_t w = 1; w < kBlock / 64; ++w) { vor |= s_or[w]; vand &= s_and[w]; }
        atomicOr(reinterpret_cast<unsigned long long*>(bits + a.parity * 2u), (unsigned long long)vor);
        atomicAnd(reinterpret_cast<unsigned long long*>(bits + a.parity * 2u + 1u), (unsigned long long)vand);
    }
}

__global__ void __launch_bounds__(kBlock)
k_sort_hist(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta, uint32_t pass) {
    __shared__ uint32_t s_hist[256];
    uint32_t k, j, n, column; char* base;
    sort_setup(blockIdx.x, a, inst_base, meta, k, j, n, base, column);
    const SortPass sp = sort_pass_info(base, a, pass);
    if (!sp.active) return;
    const uint32_t tid = threadIdx.x;
    s_hist[tid] = 0u;
    __syncthreads();
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(base + a.key_off[sp.src]);
    for (uint32_t r = 0; r < kSortTile / kBlock; ++r) {
        const uint32_t i = j * kSortTile + r * kBlock + tid;
        if (i >= n) break;
        atomicAdd(&s_hist[(uint32_t)(keys[i] >> (8u * pass)) & 0xffu], 1u);
    }
    __syncthreads();
    reinterpret_cast<uint32_t*>(base + a.hist_off)[(size_t)j * 256u + tid] = s_hist[tid];
    if (s_hist[tid]) {
        const uint32_t groups = (a.chunks_per_inst + kSortGroup - 1u) / kSortGroup;
        atomicAdd(reinterpret_cast<uint32_t*>(base + a.gsum_off) + ((size_t)pass * groups + j / kSortGroup) * 256u + tid, s_hist[tid]);
    }
}

__global__ void __launch_bounds__(kBlock)
k_sort_scatter(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta, uint32_t pass) {
    __shared__ uint32_t s_base[256];
    __shared__ uint32_t s_cnt[kBlock / 64][256];
    uint32_t k, j, n, column; char* base;
    sort_setup(blockIdx.x, a, inst_base, meta, k, j, n, base, column);
    const SortPass sp = sort_pass_info(base, a, pass);
    if (!sp.active || j * kSortTile >= n) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    const uint64_t* skey = reinterpret_cast<const uint64_t*>(base + a.key_off[sp.src]);
    const uint32_t* sval = reinterpret_cast<const uint32_t*>(base + a.val_off[sp.src]);
    uint64_t* dkey = reinterpret_cast<uint64_t*>(base + a.key_off[sp.src ^ 1u]);
    uint32_t* dval = reinterpret_cast<uint32_t*>(base + a.val_off[sp.src ^ 1u]);
    {   // offset(d, j) = sum_{d' < d} total(d') + sum_{groups before mine} gsum(g, d) + sum_{earlier tiles of my group} hist(j', d)
        const uint32_t groups = (a.chunks_per_inst + kSortGroup - 1u) / kSortGroup;
        const uint32_t used_groups = ((n + kSortTile - 1u) / kSortTile + kSortGroup - 1u) / kSortGroup;  // groups holding keys
        const uint32_t* gs = reinterpret_cast<const uint32_t*>(base + a.gsum_off) + (size_t)pass * groups * 256u;
        const uint32_t* hist = reinterpret_cast<const uint32_t*>(base + a.hist_off);
        const uint32_t my_group = j / kSortGroup;
        uint32_t total = 0, before = 0;
        for (uint32_t g = 0; g < used_groups; ++g) { const uint32_t v = gs[(size_t)g * 256u + tid]; total += v; if (g < my_group) before += v; }
        for (uint32_t t = my_group * kSortGroup; t < j; ++t) before += hist[(size_t)t * 256u + tid];
        uint32_t incl = total;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
        if (lane == 63) s_base[wave] = incl;  // s_base doubles as scratch for the wave totals
        __syncthreads();
        uint32_t digit_base = incl - total;
        for (uint32_t w = 0; w < wave; ++w) digit_base += s_base[w];
        __syncthreads();
        s_base[tid] = digit_base + before;
    }
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; ++w) s_cnt[w][tid] = 0u;
    __syncthreads();
    for (uint32_t r = 0; r < kSortTile / kBlock; ++r) {
        const uint32_t rbase = j * kSortTile + r * kBlock;
        if (rbase >= n) break;
        const uint32_t i = rbase + tid;
        const bool valid = i < n;
        const uint64_t key = valid ? skey[i] : 0ull;
        const uint32_t val = valid ? sval[i] : 0u;
        const uint32_t digit = (uint32_t)(key >> (8u * pass)) & 0xffu;
        // lanes of this wave holding the same digit (stable: earlier lanes first)
        uint64_t same = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(same & below);
        if (valid && rank == 0u) s_cnt[wave][digit] = (uint32_t)__popcll(same);
        __syncthreads();
        if (valid) {
            uint32_t off = s_base[digit];
            for (uint32_t w = 0; w < wave; ++w) off += s_cnt[w][digit];
            dkey[off + rank] = key;
            dval[off + rank] = val;
        }
        __syncthreads();
        {   // thread d advances digit d's base past this round and clears the round's counters
            uint32_t add = 0;
#pragma unroll
            for (uint32_t w = 0; w < kBlock / 64; ++w) { add += s_cnt[w][tid]; s_cnt[w][tid] = 0u; }
            s_base[tid] += add;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock)
k_sort_copy(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta) {
    uint32_t k, j, n, column; char* base;
    sort_setup(blockIdx.x, a, inst_base, meta, k, j, n, base, column);
    if (j * kSortTile >= n) return;
    const SortPass sp = sort_pass_info(base, a, 8u);  // where the result lives
    const uint64_t* bits = reinterpret_cast<const uint64_t*>(base + a.bits_off) + a.parity * 2u;
    if (bits[0] == bits[1]) return;  // every key equal: no pass ran, the list is already in order
    const uint32_t* vals = reinterpret_cast<const uint32_t*>(base + a.val_off[sp.src]);
    uint32_t* list = reinterpret_cast<uint32_t*>(base + a.alive_off[column]);
    for (uint32_t r = 0; r < kSortTile / kBlock; ++r) {
        const uint32_t i = j * kSortTile + r * kBlock + threadIdx.x;
        if (i >= n) break;
        list[i] = vals[i];
    }
}

}  // namespace hnb
