// Ribbon sort: after the update pass of an effect whose layout has RIBBON_ID, the alive list is
// reordered by (RIBBON_ID, AGE bits) ascending, stable with respect to the list order.
//
// Reference: vfx_sort_fill.wgsl (key = RIBBON_ID, key2 = AGE, value = particle index, in alive-list
// order), vfx_sort.wgsl (ONE thread running an insertion sort that only moves past strictly greater
// elements = stable), vfx_sort_copy.wgsl (values back into the same list column); scheduled after the
// update pass (src/render/mod.rs:4599-4618, 7372-7612).
//
// Here, per instance and frame, every instance of a program in the same launches:
//   k_sort_fill    64-bit keys (RIBBON_ID << 32 | AGE bits) and values in list order; OR / AND of the keys,
//                  over all rows and over the TAIL = the last `spawned` rows (this frame's spawns sit at the
//                  end of the compacted list);
//                  and, in the same pass, whether the HEAD (everything before the tail) is still in order: it is the
//                  list the previous frame's sort produced, minus the casualties, with every age advanced by the
//                  same dt, so normally yes; an update program that rewrites AGE / RIBBON_ID can break it, and
//                  then the whole list is the sort range instead of the tail;
//   8 radix passes least-significant-digit radix sort of the RANGE (tail, or everything), 8 bits per pass:
//                  k_sort_hist (LDS histogram per 4096-key tile + per-group sums) and k_sort_scatter (each
//                  tile derives its digit offsets itself from the group sums and the histograms of the earlier
//                  tiles of its group: no serial spine scan; stable ranks by wave match with ballots). A pass
//                  whose digit is identical in every key of the range (known on the device from OR ^ AND)
//                  returns immediately, and the ping-pong index follows from the passes that did run;
//   k_sort_merge   stable merge of the sorted head and the sorted tail straight into the list column (head
//                  elements first on equal keys, as in the list), by binary search into the other run; with
//                  the whole list as range it is a plain copy.
// Nothing is read back: every decision above is taken on the device from the same few words.
#pragma once
#include "hnb_kernels.hip.h"

namespace hnb {

// Items per workgroup in the sort kernels (measured: 16384-item tiles leave the chip under-filled at 4M keys
// and are 30 % slower overall), and tiles per group of the two-level digit offsets.
constexpr uint32_t kSortTile = 4096;
constexpr uint32_t kSortGroup = 32;

struct SortArgs {
    uint32_t capacity, chunks_per_inst;   // chunks_per_inst = ceil(capacity / kSortTile) here
    soff_t alive_off[2];
    soff_t key_off[2];     // u64[capacity] ping-pong
    soff_t val_off[2];     // u32[capacity] ping-pong
    soff_t hist_off;       // u32[chunks_per_inst][256]: per-tile digit counts of the current pass
    soff_t gsum_off;       // u32[8 passes][groups][256]: digit counts per group of kSortGroup tiles (zeroed by k_sort_fill)
    soff_t bits_off;       // SortState[2]: per frame parity
    soff_t rid_plane, age_plane;    // plane offsets (v == kNoPlane: key half is 0)
    uint32_t parity;       // frame parity of the state double buffer
};

struct SortState {
    uint64_t or_all, and_all;    // over every key of the instance
    uint64_t or_tail, and_tail;  // over the tail rows
    uint32_t head_unsorted;      // set by k_sort_fill
    uint32_t pad[3];
};
static_assert(sizeof(SortState) == 48, "SortState layout");

struct SortRange {
    uint32_t n;        // rows of the list
    uint32_t lo;       // first row of the sort range: rows [lo, n) are radix-sorted, rows [0, lo) are already in order
    uint64_t varying;  // key bits that differ inside the range
};
// n after this frame's update + compaction, and the range to sort
__device__ __forceinline__ SortRange sort_range_of(const SortState& st, const DevMeta& m) {
    SortRange r;
    r.n = m.alive_count;
    const uint32_t tail = m.spawned < r.n ? m.spawned : r.n;
    if (st.head_unsorted) { r.lo = 0u; r.varying = st.or_all ^ st.and_all; }
    else { r.lo = r.n - tail; r.varying = tail ? (st.or_tail ^ st.and_tail) : 0ull; }
    return r;
}
__device__ __forceinline__ SortRange sort_range(const char* base, const SortArgs& a, const DevMeta& m) {
    return sort_range_of(*(reinterpret_cast<const SortState*>(base + a.bits_off) + a.parity), m);
}

struct SortPass { bool active; uint32_t src; };
// Which ping-pong buffer pass p (0..7) reads, and whether it runs at all; p = 8 gives the buffer holding the result.
__device__ __forceinline__ SortPass sort_pass_info(uint64_t varying, uint32_t pass) {
    uint32_t ran = 0;
    for (uint32_t q = 0; q < pass; ++q) ran += ((varying >> (8u * q)) & 0xffull) ? 1u : 0u;
    SortPass r;
    r.active = pass < 8u && ((varying >> (8u * (pass & 7u))) & 0xffull) != 0ull;
    r.src = ran & 1u;
    return r;
}

__device__ __forceinline__ void sort_setup(uint32_t chunk, const SortArgs& a, const uint64_t* inst_base, uint32_t& k, uint32_t& j, char*& base) {
    k = chunk / a.chunks_per_inst;
    j = chunk - k * a.chunks_per_inst;
    base = global_ptr<char>(inst_base[k]);
}

__device__ __forceinline__ uint64_t wave_or(uint64_t v) {
#pragma unroll
    for (uint32_t off = 32; off > 0; off >>= 1) v |= (uint64_t)__shfl_xor((unsigned long long)v, off, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_and(uint64_t v) {
#pragma unroll
    for (uint32_t off = 32; off > 0; off >>= 1) v &= (uint64_t)__shfl_xor((unsigned long long)v, off, 64);
    return v;
}

// One tile of k_sort_fill. `whole`: the tile IS the instance (k_sort_tile1): the state words it would publish for the other tiles and the
// later launches are returned to the caller instead (same values: nobody else contributes).
__device__ __forceinline__ SortState sort_fill_tile(const SortArgs& a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta, uint32_t chunk, bool whole) {
    __shared__ uint64_t s_acc[4][kBlock / 64];
    __shared__ uint32_t s_bad;
    uint32_t k, j; char* base;
    sort_setup(chunk, a, inst_base, k, j, base);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (whole && tid == 0u) s_bad = 0u;
    const uint32_t n = meta[k].alive_count;
    const uint32_t tail = meta[k].spawned < n ? meta[k].spawned : n;
    const uint32_t tail_lo = n - tail;
    SortState* st = reinterpret_cast<SortState*>(base + a.bits_off);
    if (j == 0 && tid == 0) {  // re-arm next frame's state
        SortState z;
        z.or_all = 0ull; z.and_all = ~0ull; z.or_tail = 0ull; z.and_tail = ~0ull; z.head_unsorted = 0u; z.pad[0] = z.pad[1] = z.pad[2] = 0u;
        st[a.parity ^ 1u] = z;
    }
    {   // clear the per-group digit sums of all 8 passes (each tile clears a slice)
        const uint32_t groups = (a.chunks_per_inst + kSortGroup - 1u) / kSortGroup;
        const uint32_t total = 8u * groups * 256u;
        uint32_t* gs = reinterpret_cast<uint32_t*>(base + a.gsum_off);
        for (uint32_t i = j * kBlock + tid; i < total; i += a.chunks_per_inst * kBlock) gs[i] = 0u;
    }
    const uint32_t* list = reinterpret_cast<const uint32_t*>(base + a.alive_off[meta[k].write_index & 1u]);
    uint64_t* keys = reinterpret_cast<uint64_t*>(base + a.key_off[0]);
    uint32_t* vals = reinterpret_cast<uint32_t*>(base + a.val_off[0]);
    const uint32_t* rid = a.rid_plane.v == kNoPlane ? nullptr : reinterpret_cast<const uint32_t*>(base + a.rid_plane);
    const uint32_t* age = a.age_plane.v == kNoPlane ? nullptr : reinterpret_cast<const uint32_t*>(base + a.age_plane);
    uint64_t or_all = 0ull, and_all = ~0ull, or_tail = 0ull, and_tail = ~0ull;
    auto key_of = [&](uint32_t slot) { return ((uint64_t)(rid ? rid[slot] : 0u) << 32) | (uint64_t)(age ? age[slot] : 0u); };
    // Is the HEAD (rows before this frame's spawns) still in non-decreasing key order? (It was a launch of its own, k_sort_check.) Every lane
    // compares its key with its right neighbour's (a shuffle); the last lane of a wave has its neighbour in another wave or round: the first
    // and last key of every (round, wave) go through LDS and are compared after the loop; only the tile's very last row looks at global memory.
    __shared__ uint64_t s_first[kSortTile / kBlock][kBlock / 64], s_last[kSortTile / kBlock][kBlock / 64];
    bool bad = false;
    for (uint32_t r = 0; r < kSortTile / kBlock; ++r) {
        const uint32_t i = j * kSortTile + r * kBlock + tid;
        const bool valid = i < n;
        const uint32_t slot = valid ? list[i] : 0u;
        const uint64_t key = valid ? key_of(slot) : 0ull;
        if (valid) {
            keys[i] = key;
            vals[i] = slot;
            or_all |= key; and_all &= key;
            if (i >= tail_lo) { or_tail |= key; and_tail &= key; }
        }
        const uint64_t next = (uint64_t)__shfl_down((unsigned long long)key, 1, 64);
        if (lane != 63u && i + 1u < tail_lo) bad = bad || key > next;
        if (lane == 0u) s_first[r][wave] = key;
        if (lane == 63u) s_last[r][wave] = key;
    }
    __syncthreads();
    if (tid < (kSortTile / kBlock) * (kBlock / 64)) {   // one thread per (round, wave): its last row against the row that follows it
        const uint32_t r = tid / (kBlock / 64), w = tid % (kBlock / 64);
        const uint32_t i = j * kSortTile + r * kBlock + w * 64u + 63u;   // the row of that wave's last lane
        if (i + 1u < tail_lo) {
            uint64_t next;
            if (w + 1u < kBlock / 64) next = s_first[r][w + 1u];
            else if (r + 1u < kSortTile / kBlock) next = s_first[r + 1u][0];
            else next = key_of(list[i + 1u]);                              // first row of the next tile
            bad = bad || s_last[r][w] > next;
        }
    }
    or_all = wave_or(or_all); and_all = wave_and(and_all); or_tail = wave_or(or_tail); and_tail = wave_and(and_tail);
    if (lane == 0) { s_acc[0][wave] = or_all; s_acc[1][wave] = and_all; s_acc[2][wave] = or_tail; s_acc[3][wave] = and_tail; }
    __syncthreads();
    SortState out;
    out.or_all = 0ull; out.and_all = ~0ull; out.or_tail = 0ull; out.and_tail = ~0ull; out.head_unsorted = 0u; out.pad[0] = out.pad[1] = out.pad[2] = 0u;
    if (whole) {   // (the first barrier above orders the clearing of s_bad before this)
        if (__any(bad) && lane == 0u) atomicOr(&s_bad, 1u);
        __syncthreads();
        for (uint32_t w = 0; w < kBlock / 64; ++w) { out.or_all |= s_acc[0][w]; out.and_all &= s_acc[1][w]; out.or_tail |= s_acc[2][w]; out.and_tail &= s_acc[3][w]; }
        out.head_unsorted = s_bad;
        return out;
    }
    if (__any(bad) && lane == 0u) (st + a.parity)->head_unsorted = 1u;
    if (tid == 0 && j * kSortTile < n) {
        for (uint32_t w = 1; w < kBlock / 64; ++w) { or_all |= s_acc[0][w]; and_all &= s_acc[1][w]; or_tail |= s_acc[2][w]; and_tail &= s_acc[3][w]; }
        SortState* cur = st + a.parity;
        atomicOr(reinterpret_cast<unsigned long long*>(&cur->or_all), (unsigned long long)or_all);
        atomicAnd(reinterpret_cast<unsigned long long*>(&cur->and_all), (unsigned long long)and_all);
        if (j * kSortTile + kSortTile > tail_lo) {  // the tile reaches into the tail
            atomicOr(reinterpret_cast<unsigned long long*>(&cur->or_tail), (unsigned long long)or_tail);
            atomicAnd(reinterpret_cast<unsigned long long*>(&cur->and_tail), (unsigned long long)and_tail);
        }
    }
    return out;
}
__global__ void __launch_bounds__(kBlock)
k_sort_fill(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta) {
    sort_fill_tile(a, inst_base, meta, blockIdx.x, false);
}

__global__ void __launch_bounds__(kBlock)
k_sort_hist(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta, uint32_t pass) {
    __shared__ uint32_t s_hist[256];
    uint32_t k, j; char* base;
    sort_setup(blockIdx.x, a, inst_base, k, j, base);
    const SortRange rg = sort_range(base, a, meta[k]);
    const SortPass sp = sort_pass_info(rg.varying, pass);
    if (!sp.active) return;
    const uint32_t tid = threadIdx.x;
    s_hist[tid] = 0u;
    __syncthreads();
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(base + a.key_off[sp.src]);
    for (uint32_t r = 0; r < kSortTile / kBlock; ++r) {
        const uint32_t i = rg.lo + j * kSortTile + r * kBlock + tid;
        if (i >= rg.n) break;
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
    uint32_t k, j; char* base;
    sort_setup(blockIdx.x, a, inst_base, k, j, base);
    const SortRange rg = sort_range(base, a, meta[k]);
    const SortPass sp = sort_pass_info(rg.varying, pass);
    const uint32_t n = rg.n - rg.lo;  // keys in the range; rows below are relative to its start
    if (!sp.active || j * kSortTile >= n) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    const uint64_t* skey = reinterpret_cast<const uint64_t*>(base + a.key_off[sp.src]) + rg.lo;
    const uint32_t* sval = reinterpret_cast<const uint32_t*>(base + a.val_off[sp.src]) + rg.lo;
    uint64_t* dkey = reinterpret_cast<uint64_t*>(base + a.key_off[sp.src ^ 1u]) + rg.lo;
    uint32_t* dval = reinterpret_cast<uint32_t*>(base + a.val_off[sp.src ^ 1u]) + rg.lo;
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

// "This frame's spawns sort in front of everything else": where the host can prove that every key of the tail is strictly smaller than
// every key of the head (one RIBBON_ID for the whole effect, spawns start at AGE +0 and have been ticked once, every older particle at
// least twice - see HnbProgram::sort_front_* in hanabi_amd.hip), the sorted list is [tail rows in list order | head rows in list
// order]: a rotation of the compacted list by the number of spawns. No key is read and no kernel of this file runs: k_compact writes
// the survivors in that order (CompactArgs::rotate_front in hnb_kernels.hip.h).

// The whole radix sort of one instance's range by ONE workgroup: every pass whose digit varies, histogram -> digit bases ->
// stable scatter over the range's keys in order, the passes separated by workgroup barriers only. The usual ribbon frame
// sorts this frame's spawns (tens of thousands of keys, most often all equal: zero passes) - sixteen launches that each
// find nothing to do cost 4 us apiece; this is one. hnb_simulate uses it when it can bound the range on the host (see there);
// it is correct for any range, just slow for a large one.
constexpr uint32_t kSortSmallMax = 65536;
__device__ __forceinline__ void sort_small_range(const SortArgs& a, char* base, const SortRange rg) {
    __shared__ uint32_t s_base[256];
    __shared__ uint32_t s_cnt[kBlock / 64][256];
    const uint32_t n = rg.n - rg.lo;
    if (n == 0u || rg.varying == 0ull) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t src = 0;
    for (uint32_t pass = 0; pass < 8; ++pass) {
        if (((rg.varying >> (8u * pass)) & 0xffull) == 0ull) continue;
        const uint64_t* skey = reinterpret_cast<const uint64_t*>(base + a.key_off[src]) + rg.lo;
        const uint32_t* sval = reinterpret_cast<const uint32_t*>(base + a.val_off[src]) + rg.lo;
        uint64_t* dkey = reinterpret_cast<uint64_t*>(base + a.key_off[src ^ 1u]) + rg.lo;
        uint32_t* dval = reinterpret_cast<uint32_t*>(base + a.val_off[src ^ 1u]) + rg.lo;
        s_base[tid] = 0u;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) s_cnt[w][tid] = 0u;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kBlock) atomicAdd(&s_base[(uint32_t)(skey[i] >> (8u * pass)) & 0xffu], 1u);
        __syncthreads();
        {   // exclusive scan of the 256 digit counts (thread d owns digit d)
            const uint32_t total = s_base[tid];
            uint32_t incl = total;
#pragma unroll
            for (uint32_t off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
            __syncthreads();
            if (lane == 63) s_base[wave] = incl;  // scratch: the wave totals
            __syncthreads();
            uint32_t digit_base = incl - total;
            for (uint32_t w = 0; w < wave; ++w) digit_base += s_base[w];
            __syncthreads();
            s_base[tid] = digit_base;
        }
        __syncthreads();
        for (uint32_t rbase = 0; rbase < n; rbase += kBlock) {   // rounds of 256 keys in order: stable ranks by wave match (as k_sort_scatter)
            const uint32_t i = rbase + tid;
            const bool valid = i < n;
            const uint64_t key = valid ? skey[i] : 0ull;
            const uint32_t val = valid ? sval[i] : 0u;
            const uint32_t digit = (uint32_t)(key >> (8u * pass)) & 0xffu;
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
            uint32_t add = 0;
#pragma unroll
            for (uint32_t w = 0; w < kBlock / 64; ++w) { add += s_cnt[w][tid]; s_cnt[w][tid] = 0u; }
            s_base[tid] += add;
            __syncthreads();
        }
        src ^= 1u;   // (== sort_pass_info(varying, pass + 1).src: k_sort_merge finds the result where the multi-launch path leaves it)
    }
}
__global__ void __launch_bounds__(kBlock)
k_sort_small(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta) {
    char* base = global_ptr<char>(inst_base[blockIdx.x]);
    sort_small_range(a, base, sort_range(base, a, meta[blockIdx.x]));
}

// number of keys < x (strict) or <= x in the sorted run keys[0..count)
__device__ __forceinline__ uint32_t sorted_rank(const uint64_t* keys, uint32_t count, uint64_t x, bool strict) {
    if (count == 0u) return 0u;
    // the usual ribbon frame: every new particle is younger than every old one, so one of the two ends decides
    const uint64_t first = keys[0], last = keys[count - 1u];
    if (strict ? last < x : last <= x) return count;
    if (strict ? !(first < x) : !(first <= x)) return 0u;
    uint32_t lo = 0, hi = count;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t v = keys[mid];
        if (strict ? v < x : v <= x) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void sort_merge_tile(const SortArgs& a, char* base, const DevMeta* __restrict__ meta, uint32_t k, uint32_t j, const SortRange rg) {
    if (j * kSortTile >= rg.n) return;
    const uint32_t m = rg.n - rg.lo;
    if (m == 0u) return;                              // no spawn, head in order: the list stands
    if (rg.lo == 0u && rg.varying == 0ull) return;    // whole list in range but every key equal: the list stands
    const uint32_t res = sort_pass_info(rg.varying, 8u).src;  // buffer holding the sorted range
    const uint64_t* hkey = reinterpret_cast<const uint64_t*>(base + a.key_off[0]);            // head: untouched fill output
    const uint32_t* hval = reinterpret_cast<const uint32_t*>(base + a.val_off[0]);
    const uint64_t* tkey = reinterpret_cast<const uint64_t*>(base + a.key_off[res]) + rg.lo;  // sorted tail
    const uint32_t* tval = reinterpret_cast<const uint32_t*>(base + a.val_off[res]) + rg.lo;
    uint32_t* list = reinterpret_cast<uint32_t*>(base + a.alive_off[meta[k].write_index & 1u]);
    for (uint32_t r = 0; r < kSortTile / kBlock; ++r) {
        const uint32_t i = j * kSortTile + r * kBlock + threadIdx.x;
        if (i >= rg.n) break;
        if (i < rg.lo) list[i + sorted_rank(tkey, m, hkey[i], true)] = hval[i];                       // head element: strictly smaller tail keys go first
        else list[(i - rg.lo) + sorted_rank(hkey, rg.lo, tkey[i - rg.lo], false)] = tval[i - rg.lo];  // tail element: head keys <= go first
    }
}
__global__ void __launch_bounds__(kBlock)
k_sort_merge(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta) {
    uint32_t k, j; char* base;
    sort_setup(blockIdx.x, a, inst_base, k, j, base);
    sort_merge_tile(a, base, meta, k, j, sort_range(base, a, meta[k]));
}

// The whole sort of an instance that fits ONE tile (capacity <= kSortTile: a lightning bolt, a small trail) by one workgroup and in one launch:
// k_sort_fill's tile, k_sort_small's passes and k_sort_merge's tile back to back. The three communicate through the key / value buffers in
// global memory as they do across launches (a workgroup sees its own stores behind a barrier) and through the state words, which stay in
// registers here. Three dependent launches of microseconds each were 14 us of the 26-effect scene's 55 (profiles/r04t_scene.log).
__global__ void __launch_bounds__(kBlock)
k_sort_tile1(const SortArgs a, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta) {
    const uint32_t k = blockIdx.x;   // (chunks_per_inst == 1: tile == instance)
    char* base = global_ptr<char>(inst_base[k]);
    const SortState st = sort_fill_tile(a, inst_base, meta, k, true);
    const SortRange rg = sort_range_of(st, meta[k]);
    __syncthreads();
    sort_small_range(a, base, rg);
    __syncthreads();
    sort_merge_tile(a, base, meta, k, 0u, rg);
}

}  // namespace hnb
