// merge_cells.hpp -- the HarrisTrack / FastTrack MERGE PASS as a workgroup function: one workgroup marks the cells that hold a
// surviving tracked point, then compacts survivors (input order) and the cells' new corners (cell raster order) with ballot /
// popcount prefix sums.  It is the body of k_merge_compact (corner_track.hip, 8 waves) and of one workgroup of the fused
// FindHomography + merge launch (homography.hip, 4 waves): both only read what PyrLK and the cell pass left, so in a graph that
// has both they share a launch.  Included by those two translation units only; everything here has internal linkage.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "harris_cells.hpp"

namespace vsa {
namespace {

constexpr uint64_t OCCUPIED = ~0ull;

struct MergeArgs {
    int item;                       // ItemKind
    const void *tracked;
    const uint32_t *tracked_count;
    uint32_t tracked_capacity;
    void *out;
    uint32_t *out_count;
    uint32_t out_capacity;
    size_t *num_corners;
    uint32_t *out_count_host;
    uint64_t *num_corners_host;
    uint64_t *cell_best;
    int w, h, cs, ncx, ncy;
    uint32_t cs_magic;              // ceil(2^32 / cs): x / cs == umulhi(x, cs_magic) for x < 65536 (cs > 1)
    uint32_t flag_words;            // > 0: occupied cells are marked in LDS, one byte a cell, this many 32-bit words
};

struct KeypointF {
    float x, y, strength, scale, orientation;
    int32_t tracking_status;
    float error;
};

#ifdef VSA_MG_STAMPS   // development instrumentation: cycle stamps of the merge kernel's phases (thread 0)
__device__ uint64_t g_mg_stamps[16];
#define MG_STAMP(i) do { if (threadIdx.x == 0) g_mg_stamps[i] = clock64(); } while (0)
#else
#define MG_STAMP(i) do { } while (0)
#endif

// The merge is one workgroup, and what it costs a camera stream is mostly the wait for a compute unit with room for it: while
// other streams' PyrLK waves fill the GPU (4 waves x 104 VGPRs on every SIMD) a 16-wave workgroup of 96-VGPR waves waits for a
// whole CU to drain -- 19 us per launch under the bench's load against 8 us alone, with its queue (and the streams behind it
// in that hardware queue) held up meanwhile.  Eight waves of at most 48 VGPRs fit beside a full set of PyrLK waves (2 x 48 =
// the 96 VGPRs they leave per SIMD): bench 28.4 k -> 30.8 k frames/s, single stream 12.2 k -> 12.0 k.  To stay at 48 VGPRs a wave
// walks its range in small blocks and re-reads it for the write-out (L2 hits) instead of keeping it in registers
// (a form that kept its ranges in registers between the counting pass and the write-out needed 96 VGPRs and lost 8 %).  The budget is
// tight enough for the register allocator's mood to matter (the same source as a function instead of a kernel body: 52, and the
// 32-stream bench 17 % slower), so the one place where all 48 were in use -- the scan over the wave totals -- keeps its values in
// scalar registers, and tests/test_kernel_resources.py reads the count from the built library.
constexpr uint32_t MG_THREADS = 512;   // k_merge_compact's workgroup (the fused FindHomography + merge launch runs the pass on 4 waves)
// rounds of 64 items per wave and block: tracked points / cells (1080p, cell 18: a wave owns 8 rounds of points, 13 of cells)
constexpr uint32_t MG_KT = 4, MG_KC = 7;

// Every fetch below is unconditional, from an index clamped into the array, and the validity test follows once the
// whole block of loads is out: a load inside a branch (or a select fed by one) makes the compiler wait for it where
// the branch joins, which strings a block's loads into one memory round trip each.
//
// N rounds of tracked points starting at item `first` (lane = item); bit r of the result: point r is there
// (below `nt`, round below `rounds`), tracked, and inside the image.
template <uint32_t N>
__device__ __forceinline__ uint32_t tracked_fetch(const MergeArgs &a, uint32_t first, uint32_t round0, uint32_t rounds,
                                                  uint32_t nt, float (&x)[N], float (&y)[N])
{
    const uint32_t last = a.tracked_capacity - 1u;      // callers: capacity > 0
    int32_t st[N];
    if (a.item == kPoint2f) {
#pragma unroll
        for (uint32_t r = 0; r < N; ++r) {
            const float2 p = reinterpret_cast<const float2 *>(a.tracked)[min(first + r * 64u, last)];
            x[r] = p.x; y[r] = p.y; st[r] = 1;
        }
    } else {
#pragma unroll
        for (uint32_t r = 0; r < N; ++r) {
            const KeypointF *p = reinterpret_cast<const KeypointF *>(a.tracked) + min(first + r * 64u, last);
            x[r] = p->x; y[r] = p->y; st[r] = p->tracking_status;
        }
    }
    const float xmax = (float)(a.w - 1), ymax = (float)(a.h - 1);
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t r = 0; r < N; ++r) {
        const bool alive = round0 + r < rounds && first + r * 64u < nt && st[r] != 0 &&
                           x[r] >= 0.0f && y[r] >= 0.0f && x[r] <= xmax && y[r] <= ymax;
        bits |= (alive ? 1u : 0u) << r;
    }
    return bits;
}

// N rounds of cell keys starting at cell `first`; cells that are not there read as empty
template <uint32_t N>
__device__ __forceinline__ void keys_fetch(const MergeArgs &a, uint32_t first, uint32_t round0, uint32_t rounds,
                                           uint32_t ncells, uint64_t (&k)[N])
{
#pragma unroll
    for (uint32_t r = 0; r < N; ++r) k[r] = a.cell_best[min(first + r * 64u, ncells - 1u)];
#pragma unroll
    for (uint32_t r = 0; r < N; ++r) k[r] = (round0 + r < rounds && first + r * 64u < ncells) ? k[r] : 0ull;
}

__device__ __forceinline__ uint32_t cell_of(const MergeArgs &a, float x, float y)
{
    // x, y lie inside an image of at most 65535 x 65535: with m = ceil(2^32 / cs) = (2^32 + e) / cs, e < cs < 2^16,
    // x * m / 2^32 = x / cs + x * e / (cs * 2^32) and the excess stays below 1 / cs, so the floor is that of x / cs
    const uint32_t ux = (uint32_t)x, uy = (uint32_t)y;
    const uint32_t cx = min(a.cs == 1 ? ux : __umulhi(ux, a.cs_magic), (uint32_t)a.ncx - 1u);
    const uint32_t cy = min(a.cs == 1 ? uy : __umulhi(uy, a.cs_magic), (uint32_t)a.ncy - 1u);
    return cy * (uint32_t)a.ncx + cx;
}

__device__ __forceinline__ void emit_survivor(const MergeArgs &a, uint32_t pos, uint32_t i, float x, float y)
{
    if (pos >= a.out_capacity) return;
    if (a.item == kPoint2f) reinterpret_cast<float2 *>(a.out)[pos] = make_float2(x, y);
    else reinterpret_cast<KeypointF *>(a.out)[pos] = reinterpret_cast<const KeypointF *>(a.tracked)[i];
}

__device__ __forceinline__ void emit_corner(const MergeArgs &a, uint32_t pos, uint64_t key)
{
    if (pos >= a.out_capacity) return;
    const uint32_t idx = ~(uint32_t)(key & 0xffffffffull);
    const float fx = (float)(idx & 0xffffu), fy = (float)(idx >> 16);
    if (a.item == kPoint2f) {
        reinterpret_cast<float2 *>(a.out)[pos] = make_float2(fx, fy);
    } else {
        KeypointF o;
        o.x = fx; o.y = fy;
        o.strength = from_orderable((uint32_t)(key >> 32));
        o.scale = 0.0f; o.orientation = 0.0f;
        o.tracking_status = 1;
        o.error = 0.0f;
        reinterpret_cast<KeypointF *>(a.out)[pos] = o;
    }
}

// One workgroup, one pass.  Wave w owns the w-th contiguous range of the tracked points and of the cells and walks
// it in rounds of 64 consecutive items (lane = item: coalesced loads), so positions follow input order / cell raster
// order: within a round by the ballot's prefix count, across rounds by a running count, across waves by the wave
// totals (both lists' totals travel as the two halves of one 64-bit word).  The write-out re-reads its ranges (L2 hits)
// instead of keeping them in registers.
__device__ __forceinline__ uint32_t lanes_below(uint64_t ballot)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

// KT / KC: rounds of tracked points / cells per block of a walk (a block's loads are in flight together)
// THREADS: workgroup size (a power-of-two number of waves).  The launch's dynamic LDS is the cell flags, one byte per cell
// (a.flag_words words, at least one).
template <uint32_t THREADS, uint32_t KT, uint32_t KC>
__device__ __forceinline__ void merge_compact_wg(const MergeArgs a)
{
    constexpr uint32_t WAVES = THREADS / 64u;
    __shared__ uint64_t wave_tot[WAVES];
    extern __shared__ uint32_t occ_words[];  // one byte per cell: holds a surviving tracked point
    uint8_t *occ = reinterpret_cast<uint8_t *>(occ_words);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    MG_STAMP(0);
    // Cells holding a surviving tracked point are marked in LDS, one byte a cell (plain stores: tracked points arrive
    // in cell order, so bits would have the lanes of a wave serialize on the same word with atomic ORs); the keys stay
    // untouched.  Only a cell grid too large for LDS is marked in place, in the keys themselves.
    const bool in_lds = a.flag_words != 0u;
    for (uint32_t i = tid; i < max(a.flag_words, 1u); i += THREADS) occ_words[i] = 0u;   // word 0 is always there
    const bool has_t = a.tracked != nullptr && a.tracked_capacity != 0u;
    const uint32_t cap_t = has_t ? a.tracked_capacity : 0u;
    // the live count through a vector load (an index the compiler cannot see is zero): as a scalar load it would
    // have to land before the first vector load issues, one more memory round trip at the head of the kernel
    uint32_t zero = 0u;
    asm volatile("" : "+v"(zero));
    const uint32_t nt = has_t ? min(a.tracked_count[zero], cap_t) : 0u;
    const uint32_t ncells = (uint32_t)a.ncx * (uint32_t)a.ncy;
    // range sizes come from the capacity, not the live count
    const uint32_t rounds_t = (cap_t + THREADS - 1) / THREADS, rounds_c = (ncells + THREADS - 1) / THREADS;
    const uint32_t tbase = wave * rounds_t * 64u + lane, cbase = wave * rounds_c * 64u + lane;

    // phase 0: cells holding a surviving tracked point emit no new corner
    uint32_t n_alive = 0;
    __syncthreads();      // flags cleared
    MG_STAMP(1);
    for (uint32_t r0 = 0; has_t && r0 < rounds_t; r0 += KT) {   // blocks of KT rounds: the loads of a block are in flight together
        float x[KT], y[KT];
        const uint32_t bits = tracked_fetch<KT>(a, tbase + r0 * 64u, r0, rounds_t, nt, x, y);
#pragma unroll
        for (uint32_t r = 0; r < KT; ++r) {
            const bool alive = (bits >> r) & 1u;
            if (alive) {
                const uint32_t cell = cell_of(a, x[r], y[r]);
                if (in_lds) occ[cell] = 1;
                else a.cell_best[cell] = OCCUPIED;
            }
            n_alive += (uint32_t)__popcll(__ballot(alive));
        }
    }
    MG_STAMP(2);
    if (!in_lds) __threadfence_block();
    __syncthreads();
    MG_STAMP(3);

    // a cell emits its corner when it has one and holds no surviving tracked point
    auto fresh_key = [&](uint64_t k, uint32_t c) {
        // no short circuit: a flag read behind a branch would make every round wait for its own LDS round trip
        const uint32_t marked = occ[in_lds ? min(c, ncells - 1u) : 0u];
        return (k != 0ull) & (k != OCCUPIED) & (marked == 0u);
    };
    uint32_t n_new = 0;
    for (uint32_t r0 = 0; r0 < rounds_c; r0 += KC) {
        uint64_t k[KC];
        keys_fetch<KC>(a, cbase + r0 * 64u, r0, rounds_c, ncells, k);
#pragma unroll
        for (uint32_t r = 0; r < KC; ++r) n_new += (uint32_t)__popcll(__ballot(fresh_key(k[r], cbase + (r0 + r) * 64u)));
    }

    // the counts are wave-uniform: exclusive scan over the wave totals
    MG_STAMP(4);
    if (lane == 0u) wave_tot[wave] = (uint64_t)n_alive | ((uint64_t)n_new << 32);
    __syncthreads();
    MG_STAMP(5);
    // (wave-uniform sums kept out of the vector registers: lane i reads wave i's totals, the scan runs on scalar copies of them --
    // read as eight values per lane they are 16 registers at the one place where this kernel's 48 are all in use)
    static_assert((WAVES & (WAVES - 1u)) == 0u, "lane -> wave total by masking");
    const uint64_t tot_lane = wave_tot[lane & (WAVES - 1u)];
    uint32_t off_a = 0, off_n = 0, total_a = 0, total_n = 0;
#pragma unroll
    for (uint32_t i = 0; i < WAVES; ++i) {
        const uint32_t ta = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tot_lane, (int)i);
        const uint32_t tn = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tot_lane >> 32), (int)i);
        off_a += (i < wave) ? ta : 0u;
        off_n += (i < wave) ? tn : 0u;
        total_a += ta;
        total_n += tn;
    }
    const uint64_t wave_off = (uint64_t)off_a | ((uint64_t)off_n << 32);
    const uint64_t total = (uint64_t)total_a | ((uint64_t)total_n << 32);
    const uint32_t total_alive = (uint32_t)total, total_new = (uint32_t)(total >> 32);

    uint32_t pos = (uint32_t)wave_off;                              // survivors, input order
    uint32_t cpos = total_alive + (uint32_t)(wave_off >> 32);       // new corners, cell raster order
    for (uint32_t r0 = 0; has_t && r0 < rounds_t; r0 += KT) {
        float x[KT], y[KT];
        const uint32_t bits = tracked_fetch<KT>(a, tbase + r0 * 64u, r0, rounds_t, nt, x, y);
#pragma unroll
        for (uint32_t r = 0; r < KT; ++r) {
            const bool alive = (bits >> r) & 1u;
            const uint64_t b = __ballot(alive);
            if (alive) emit_survivor(a, pos + lanes_below(b), tbase + (r0 + r) * 64u, x[r], y[r]);
            pos += (uint32_t)__popcll(b);
        }
    }
    MG_STAMP(7);
    for (uint32_t r0 = 0; r0 < rounds_c; r0 += KC) {
        uint64_t k[KC];
        keys_fetch<KC>(a, cbase + r0 * 64u, r0, rounds_c, ncells, k);
#pragma unroll
        for (uint32_t r = 0; r < KC; ++r) {
            const bool fresh = fresh_key(k[r], cbase + (r0 + r) * 64u);
            const uint64_t b = __ballot(fresh);
            if (fresh) emit_corner(a, cpos + lanes_below(b), k[r]);
            cpos += (uint32_t)__popcll(b);
        }
    }
    MG_STAMP(6);
    if (tid == 0) {
        const uint32_t base = total_alive + total_new;
        *a.out_count = min(base, a.out_capacity);
        if (a.num_corners) *a.num_corners = (size_t)base;
        if (a.out_count_host) *a.out_count_host = min(base, a.out_capacity);
        if (a.num_corners_host) *a.num_corners_host = (uint64_t)base;
    }
}

// host side: the merge pass's arguments and dynamic LDS for a detector node's parameters (false: parameters it cannot run with)
inline bool merge_setup(const CornerTrackParams &p, MergeArgs &m, size_t &lds)
{
    const uint32_t cs = p.cell_size;
    if (cs == 0u || p.image.width > 65535u || p.image.height > 65535u) return false;
    m = MergeArgs{};
    m.item = p.item; m.tracked = p.tracked; m.tracked_count = p.tracked_count;
    m.tracked_capacity = p.tracked_capacity;
    m.out = p.out; m.out_count = p.out_count; m.out_capacity = p.out_capacity;
    m.num_corners = p.num_corners; m.cell_best = reinterpret_cast<uint64_t *>(p.scratch);
    m.out_count_host = p.out_count_host; m.num_corners_host = p.num_corners_host;
    m.w = (int)p.image.width; m.h = (int)p.image.height; m.cs = (int)cs;
    m.ncx = (int)((p.image.width + cs - 1) / cs); m.ncy = (int)((p.image.height + cs - 1) / cs);
    const uint32_t ncells = (uint32_t)m.ncx * (uint32_t)m.ncy;
    if (ncells == 0u) return false;
    m.flag_words = ncells <= 60u * 1024u ? (ncells + 3u) / 4u : 0u;   // beyond 60 K cells the keys are marked in place
    lds = (size_t)(m.flag_words ? m.flag_words : 1u) * 4u;
    m.cs_magic = cs > 1u ? (uint32_t)((0x100000000ull + cs - 1u) / cs) : 0u;
    return true;
}

}  // namespace
}  // namespace vsa
