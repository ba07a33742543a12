// homography.hip -- nvxFindHomography (RANSAC) on the PyrLK matches.
//
// Reference call sites replaced: nvxFindHomographyNode(graph, srcPoints, dstPoints, homography, RANSAC, threshold,
// maxEstimateIters, maxRefineIters, confidence, outlierRatio, mask)  (demos/video_stabilizer/stabilizer.cpp:264-273,
// VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp:1576-1578); output conventions from the consumer
// demos/video_stabilizer/homography_filter_node.cpp:36-159.  Semantics and the exact floating-point order:
// oracle/vxoracle_homography.c -- results are bit-identical.
//
// One launch, no host round trip:
//   * workgroup k evaluates hypothesis k (all hypotheses of the a-priori RANSAC count are independent, so they run as
//     one batch): wave 0 draws the 4-point sample and maps it exactly in binary64 while the other waves fetch points;
//     the workgroup then counts the hypothesis' inliers (binary32, division-free) and publishes count and matrix;
//   * the workgroup that takes the last ticket goes on: it picks the best hypothesis, normalises coordinates, then alternates
//     least-squares refits (29 binary64 sums in 256 interleaved partials + butterfly, 8x8 LDL^T) with inlier recounts; writes
//     matrix, mask and counts (also to pinned mirrors).
// The work is tiny (56 x 4096 point tests): latency is what matters to the caller -- and to the other camera streams on the GPU
// the kernel's footprint, which decides whether its workgroups start beside their PyrLK waves or wait for them to drain.  Hence
//   * both kernels fit in the 96 registers and 40 KB of LDS that four PyrLK waves per SIMD leave (the refit's sums in three passes,
//     the solve reading them from LDS).  k_find_homography_lean, used when the arrays' capacities bound the point count by
//     16 * 256 (the tracker graphs'): a thread's 16 correspondences stay on chip for the whole kernel (destination points in
//     registers, source points in LDS) and the loads do not wait for the item counts; k_find_homography (any size) streams the
//     points from memory in every pass, four per thread in flight;
//   * all sums of a pass go through ONE barrier pair: DPP butterfly inside each wave, four wave totals through LDS;
//   * the serial binary64 parts (sample -> hypothesis, LDL^T, denormalisation) run on wave 0 with lanes as the
//     parallel dimension: independent divisions share one instruction sequence, row i of the factorisation lives on
//     lane i.  Every element still sees the operations of the sequential algorithm in the same order;
//   * what the hypothesis workgroups hand to the refining one goes past the L2s word by word (device-scope relaxed atomics)
//     instead of through a device-scope release / acquire, which on this GPU writes back and invalidates whole L2s.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "kernels.hpp"
#include "merge_cells.hpp"

namespace vsa {
namespace {

#ifdef VSA_HG_STAMPS   // development instrumentation: cycle stamps of the phases (thread 0 of workgroup 0)
__device__ uint64_t g_hg_stamps[64];
#define HG_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_hg_stamps[i] = clock64(); } while (0)
#define HG_STAMP_ANY(i) do { if (threadIdx.x == 0) g_hg_stamps[i] = clock64(); } while (0)      // the one refining workgroup
#else
#define HG_STAMP(i) do { } while (0)
#define HG_STAMP_ANY(i) do { } while (0)
#endif

constexpr int HG_T = 256;      // threads = interleaved partial sums (the oracle's HG_P)
constexpr int HG_NACC = 29;
constexpr int HG_KPRE = 64;    // up to this many hypotheses are staged in LDS by the refinement

struct HgArgs {
    const float2 *src, *dst;
    const uint32_t *src_count, *dst_count;
    uint32_t src_cap, dst_cap;
    float thr2;
    uint32_t K, max_refine;
    float *matrix, *matrix_host;
    uint8_t *mask;
    uint32_t *mask_count, *mask_count_host;
    uint32_t mask_cap;
    uint32_t *ticket;   // workgroups that have published their hypothesis; 0 between launches.  Behind it (homography_scratch_bytes):
                        // keys, per hypothesis: inlier count + 1, or 0 when the hypothesis is invalid; hyp, per hypothesis: 9 floats, h22 = 1
};
// (derived where they are used rather than passed: two pointers are four scalar registers for the whole kernel)
__device__ __forceinline__ uint32_t *hg_keys(const HgArgs &a) { return a.ticket + 4; }
__device__ __forceinline__ float *hg_hyp(const HgArgs &a) { return reinterpret_cast<float *>(a.ticket + 4 + ((a.K + 3u) & ~3u)); }

__device__ __forceinline__ uint32_t hg_n(const HgArgs &a)
{
    uint32_t n = min(min(*a.src_count, a.src_cap), min(*a.dst_count, a.dst_cap));
    if (a.mask) n = min(n, a.mask_cap);
    return n;
}

__device__ __forceinline__ bool hg_usable(float2 s, float2 d) { return s.x >= 0.0f && s.y >= 0.0f && d.x >= 0.0f && d.y >= 0.0f; }

__device__ __forceinline__ uint32_t hg_draw(uint32_t &s, uint32_t n)
{
    s = s * 1664525u + 1013904223u;
    return (uint32_t)(((uint64_t)(s >> 8) * n) >> 24);
}

// reprojection error <= threshold, evaluated without a division: |(X,Y) - W (u,v)|^2 <= thr^2 W^2, W > 0
__device__ __forceinline__ bool hg_inlier(const float *H, float2 s, float2 d, float thr2)
{
    const float X = fmaf(H[0], s.x, fmaf(H[1], s.y, H[2]));
    const float Y = fmaf(H[3], s.x, fmaf(H[4], s.y, H[5]));
    const float W = fmaf(H[6], s.x, fmaf(H[7], s.y, H[8]));
    const float ex = fmaf(-d.x, W, X), ey = fmaf(-d.y, W, Y);
    return (W > 1e-6f) & (fmaf(ex, ex, ey * ey) <= thr2 * (W * W));
}

// value of lane `src_lane` (a constant once unrolled) in every lane
__device__ __forceinline__ double lane_bcast(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

template <int N>
__device__ __forceinline__ double lane_select(const double (&v)[N], uint32_t lane)
{
    double r = v[0];
#pragma unroll
    for (int i = 1; i < N; ++i) {
        r = lane == (uint32_t)i ? v[i] : r;
        asm volatile("" : "+v"(r));     // a chain of selects, not a table in scratch memory
    }
    return r;
}

// DPP move of a binary64 value; lanes the row mask excludes (or with no source lane) hold an undefined value, which
// the callers never use
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    int r = __builtin_amdgcn_mov_dpp((int)v, CTRL, ROW_MASK, 0xf, false);
    asm volatile("" : "+v"(r));       // keep the move a move (not folded into the add)
    return (uint32_t)r;
}
// Pairwise butterfly over the 64 lanes (offsets 1..32), total valid in lane 63 ONLY.  Inside a row of 16 the partner
// of steps 4 and 8 may be any lane of the other group (the groups are uniform by then); rows combine as
// (r0 + r1) + (r2 + r3), exactly the butterfly's pairing (addition commutes): row_bcast:15 adds row 0 into row 1 and
// row 2 into row 3, row_bcast:31 then adds row 1 into row 3.  All 64 lanes must be active.
template <typename T>
__device__ __forceinline__ T wave_sum63(T v)
{
    v = v + dpp_mov<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141, 0xf>(v);   // row_half_mirror
    v = v + dpp_mov<0x140, 0xf>(v);   // row_mirror
    v = v + dpp_mov<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = v + dpp_mov<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// the largest of the 64 lanes' values, valid in lane 63 only (the moves of wave_sum63)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_max63(unsigned long long v)
{
    v = dpp_max<0xB1, 0xf>(v);
    v = dpp_max<0x4E, 0xf>(v);
    v = dpp_max<0x141, 0xf>(v);
    v = dpp_max<0x140, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v);
    v = dpp_max<0x143, 0xc>(v);
    return v;
}

// The butterfly over the 256 per-thread partials (offsets 1..128) for N sums at once, one barrier pair: offsets
// 1..32 inside each wave, offsets 64 and 128 pair the four wave totals as (W0 + W1) + (W2 + W3).
template <int N>
__device__ __forceinline__ void block_tree_n(double (&v)[N], double (*w)[4])
{
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = wave_sum63(v[q]);
    __syncthreads();      // the previous round's readers are done with w
    if ((threadIdx.x & 63) == 63) {
#pragma unroll
        for (int q = 0; q < N; ++q) w[q][threadIdx.x >> 6] = v[q];
    }
    __syncthreads();
}
template <int N>
__device__ __forceinline__ void block_tree_read(double (&v)[N], double (*w)[4])
{
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = (w[q][0] + w[q][1]) + (w[q][2] + w[q][3]);
}

// block-wide integer sum of a per-thread count (order-free)
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *lds4)
{
    v = wave_sum63(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// H (binary64, any scale) -> binary32 with h22 = 1, element `lane` (< 9) in lane `lane`: the nine quotients are one
// division across nine lanes.  false if H cannot be normalised (oracle: hg_to_f32).
__device__ __forceinline__ bool hg_to_f32_lanes(const double (&H)[9], uint32_t lane, float &elem)
{
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { const double v = fabs(H[i]); if (v > m) m = v; }
    const bool ok = (fabs(H[8]) > 1e-9 * m) && (m < INFINITY);
    elem = (float)(lane_select<9>(H, lane) / H[8]);
    return ok;
}

// Executed by every lane of wave 0 with identical values (a wave instruction costs the same for one lane or 64), which
// lets independent divisions share one instruction sequence.  Hypothesis k: the 4-point sample and its exact map
// (binary64) rounded to binary32; element `lane` of it in lanes 0..8.
__device__ __forceinline__ bool hg_make_hypothesis(const HgArgs &a, uint32_t k, uint32_t n, uint32_t lane, float &elem)
{
    elem = (lane % 4u == 0u) ? 1.0f : 0.0f;
    if (n < 4) return false;
    uint32_t idx[4];
    float2 sp[4], dp[4];
    // the common case: the first four draws are usable and distinct -- fetch them together, then check
    uint32_t s = (k + 1u) * 0x9E3779B9u;
#pragma unroll
    for (int j = 0; j < 4; ++j) idx[j] = hg_draw(s, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sp[j] = a.src[idx[j]]; dp[j] = a.dst[idx[j]]; }
    bool fast = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        fast = fast && hg_usable(sp[j], dp[j]);
#pragma unroll
        for (int q = 0; q < j; ++q) fast = fast && idx[q] != idx[j];
    }
    if (!fast) {   // the general procedure, from the start (uniform: every lane takes the same path)
        s = (k + 1u) * 0x9E3779B9u;
        bool ok = true;
        for (int j = 0; j < 4 && ok; ++j) {
            bool got = false;
            for (int t = 0; t < 64 && !got; ++t) {
                const uint32_t i = hg_draw(s, n);
                const float2 ss = a.src[i], dd = a.dst[i];
                if (!hg_usable(ss, dd)) continue;
                bool dup = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) dup |= (q < j) & (idx[q] == i);
                if (dup) continue;
                // (register arrays indexed by a loop counter: selects, not scratch)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q == j) { idx[q] = i; sp[q] = ss; dp[q] = dd; }
                got = true;
            }
            ok = got;
        }
        if (!ok) return false;
    }
    // unit square -> quadrilateral for both point sets (Heckbert); the four quotients g, h share one division.  Written in stages
    // (source side, destination side, quotients, adjugate of A, then H = B adj(A) row by row) with the coordinates converted where
    // they are used, and the stages kept apart for the instruction scheduler: few binary64 values are live at any time, which is
    // what lets the 96-register kernel build its hypothesis beside a thread's points.  Every value is the same expression as ever.
    auto X = [&](int j) { return (double)sp[j].x; };
    auto Y = [&](int j) { return (double)sp[j].y; };
    auto U = [&](int j) { return (double)dp[j].x; };
    auto V = [&](int j) { return (double)dp[j].y; };
    double num[4], aden, bden;
    {
        const double adx1 = X(1) - X(2), adx2 = X(3) - X(2), ady1 = Y(1) - Y(2), ady2 = Y(3) - Y(2);
        const double asx = ((X(0) - X(1)) + X(2)) - X(3), asy = ((Y(0) - Y(1)) + Y(2)) - Y(3);
        aden = adx1 * ady2 - adx2 * ady1;
        num[0] = asx * ady2 - adx2 * asy;
        num[1] = adx1 * asy - asx * ady1;
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const double bdx1 = U(1) - U(2), bdx2 = U(3) - U(2), bdy1 = V(1) - V(2), bdy2 = V(3) - V(2);
        const double bsx = ((U(0) - U(1)) + U(2)) - U(3), bsy = ((V(0) - V(1)) + V(2)) - V(3);
        bden = bdx1 * bdy2 - bdx2 * bdy1;
        num[2] = bsx * bdy2 - bdx2 * bsy;
        num[3] = bdx1 * bsy - bsx * bdy1;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(aden != 0.0) || !(bden != 0.0)) return false;
    const double quo = lane_select<4>(num, lane) / (lane < 2u ? aden : bden);
    const double ag = lane_bcast(quo, 0), ah = lane_bcast(quo, 1), bg = lane_bcast(quo, 2), bh = lane_bcast(quo, 3);
    __builtin_amdgcn_sched_barrier(0);
    double adj[9], H[9];
    {
        double A[9];
        A[0] = (X(1) - X(0)) + ag * X(1); A[1] = (X(3) - X(0)) + ah * X(3); A[2] = X(0);
        A[3] = (Y(1) - Y(0)) + ag * Y(1); A[4] = (Y(3) - Y(0)) + ah * Y(3); A[5] = Y(0);
        A[6] = ag; A[7] = ah; A[8] = 1.0;
        adj[0] = A[4] * A[8] - A[5] * A[7]; adj[1] = A[2] * A[7] - A[1] * A[8]; adj[2] = A[1] * A[5] - A[2] * A[4];
        adj[3] = A[5] * A[6] - A[3] * A[8]; adj[4] = A[0] * A[8] - A[2] * A[6]; adj[5] = A[2] * A[3] - A[0] * A[5];
        adj[6] = A[3] * A[7] - A[4] * A[6]; adj[7] = A[1] * A[6] - A[0] * A[7]; adj[8] = A[0] * A[4] - A[1] * A[3];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double B[3];
        if (r == 0) { B[0] = (U(1) - U(0)) + bg * U(1); B[1] = (U(3) - U(0)) + bh * U(3); B[2] = U(0); }
        else if (r == 1) { B[0] = (V(1) - V(0)) + bg * V(1); B[1] = (V(3) - V(0)) + bh * V(3); B[2] = V(0); }
        else { B[0] = bg; B[1] = bh; B[2] = 1.0; }
#pragma unroll
        for (int c = 0; c < 3; ++c) H[3 * r + c] = (B[0] * adj[c] + B[1] * adj[3 + c]) + B[2] * adj[6 + c];
        __builtin_amdgcn_sched_barrier(0);
    }
    float e;
    if (!hg_to_f32_lanes(H, lane, e)) return false;
    elem = e;
    return true;
}

// index into the 29 sums of element (i, c), c <= i, of the normal-equation matrix (-1: zero)
__host__ __device__ constexpr int hg_midx(int i, int c)
{
    return (i < 3) ? (c == 0 ? i : (c == 1 ? (i == 1 ? 3 : 4) : 5))                        // pp, lower triangle
         : (i < 6) ? (c < 3 ? -1 : hg_midx(i - 3, c - 3))
         : (c < 3) ? 6 + 2 * c + (i - 6)
         : (c < 6) ? 12 + 2 * (c - 3) + (i - 6)
         : 18 + (i - 6) + (c - 6);
}
static_assert(hg_midx(0, 0) == 0 && hg_midx(1, 0) == 1 && hg_midx(2, 0) == 2 && hg_midx(1, 1) == 3 && hg_midx(2, 1) == 4 &&
              hg_midx(2, 2) == 5 && hg_midx(4, 3) == 1 && hg_midx(5, 5) == 5 && hg_midx(3, 2) == -1 && hg_midx(6, 0) == 6 &&
              hg_midx(7, 0) == 7 && hg_midx(6, 2) == 10 && hg_midx(7, 5) == 17 && hg_midx(6, 3) == 12 && hg_midx(6, 6) == 18 &&
              hg_midx(7, 6) == 19 && hg_midx(7, 7) == 20, "normal-equation layout");

struct HgNorm { double cx, cy, cu, cv, ss, sd; };

// ---- the kernels: at most 96 registers, so that a workgroup starts beside four PyrLK waves per SIMD --------------------------------
// Rounds 1 and 2 kept a thread's 16 correspondences in registers (298 in all, 19.8 us alone).  What that kernel cost the OTHER
// camera streams was its footprint, not its work: a workgroup of four 298-register waves needs a compute unit whose four SIMDs
// each hold at most two PyrLK waves, so it waits for the other streams' PyrLK launches to drain, and the four hardware queues fall
// into step -- a do-nothing kernel with that footprint takes the 32-stream bench from 38.3 k to 30.3 k frames/s, the real work
// another 1.7 k (profiles/r03_homography_dummy_kernels.txt).
// Four PyrLK waves leave 96 registers per SIMD and 40 KB of LDS per compute unit; both forms below fit in them:
//   * the refit's 29 binary64 sums in three passes over the inliers (10 + 10 + 9 sums; every sum still sees its products in the
//     same order), one barrier pair for all three; the current matrix waits in LDS meanwhile;
//   * the 8 x 8 solve reads the sums from LDS where it needs them (row i on lane i), keeps the pivots and the back-substituted
//     solution in scalar registers and takes L's columns from the rows by lane reads instead of holding a transposed copy;
//   * arrays of at most HG_NPT * 256 correspondences (the tracker graphs'): a thread's 16 correspondences stay ON CHIP -- destination
//     points in registers (32), source points in LDS (32 KB per workgroup, fetched straight into it, 16 bytes a lane; the barrier
//     behind the hypothesis stands between those loads and the first read), usable and inlier bits in one register;
//   * larger arrays: every pass streams the points from memory again (L2 hits) and tests them against the current matrix.
// Results are bit-identical to the oracle (and to the 298-register kernel of rounds 1 and 2).
constexpr int HG_NPT = 16;
static_assert(HG_NPT <= 16, "usable and inlier bits of a thread's points share one 32-bit register");

// a wave-uniform value, moved to scalar registers (the first lane's copy)
__device__ __forceinline__ double uniform64(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// wave-level part of block_tree_n for one value: lane 63 of each wave stores its wave's total
__device__ __forceinline__ void wave_total_to_lds(double v, double (*w)[4], int q, uint32_t lane, uint32_t wv)
{
    v = wave_sum63(v);
    if (lane == 63u) w[q][wv] = v;
}

// The same totals for several sums in ONE butterfly (the scheme of pyrlk.hip's wave_sum2/3, in binary64): after the step of
// offset 1 a sum needs only every other lane, after offset 2 every fourth, so lanes l % 4 == 0..3 carry sums a..d from there
// on.  The remaining steps must pair lanes of equal l % 4: rotations by 4 and 8 inside a row (canonical order in lanes
// 12..15), then the row swaps, lane i of row r with lane i of row r ^ 1 and of row r ^ 2.  Every sum keeps the pairing of
// wave_sum63 -- ((l0 + l1) + (l2 + l3)) per quad, (q0 + q1) + (q2 + q3) per row, (r0 + r1) + (r2 + r3) -- and binary64
// addition commutes exactly, so the totals are the same bits.  Lanes 60..63 end with the totals of a..d.
typedef unsigned hg_v2u __attribute__((ext_vector_type(2)));
template <bool ROWS32>
__device__ __forceinline__ double row_swap_sum(double z)
{
    const unsigned lo = (unsigned)__double2loint(z), hi = (unsigned)__double2hiint(z);
    const hg_v2u l = ROWS32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const hg_v2u h = ROWS32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
}
__device__ __forceinline__ double wave_sum_rows(double z)
{
    z = z + dpp_mov<0x124, 0xf>(z);   // row_ror:4
    z = z + dpp_mov<0x128, 0xf>(z);   // row_ror:8
    z = row_swap_sum<false>(z);       // rows 0,1: r0 + r1; rows 2,3: r2 + r3
    return row_swap_sum<true>(z);     // (r0 + r1) + (r2 + r3) in every row
}
__device__ __forceinline__ double wave_sum_pair(double a, double b, bool lane_bit0)
{
    a = a + dpp_mov<0xB1, 0xf>(a);    // quad_perm [1,0,3,2]
    b = b + dpp_mov<0xB1, 0xf>(b);
    double z = lane_bit0 ? b : a;
    z = z + dpp_mov<0x4E, 0xf>(z);    // quad_perm [2,3,0,1]: a partner of the same parity
    return wave_sum_rows(z);          // a in lane 62, b in lane 63
}
__device__ __forceinline__ double wave_sum_quad(double a, double b, double c, double d, bool lane_bit0, bool lane_bit1)
{
    a = a + dpp_mov<0xB1, 0xf>(a);
    b = b + dpp_mov<0xB1, 0xf>(b);
    c = c + dpp_mov<0xB1, 0xf>(c);
    d = d + dpp_mov<0xB1, 0xf>(d);
    double y = lane_bit0 ? b : a, z = lane_bit0 ? d : c;
    y = y + dpp_mov<0x4E, 0xf>(y);
    z = z + dpp_mov<0x4E, 0xf>(z);
    z = lane_bit1 ? z : y;            // lanes 0, 1 of a quad: a, b; lanes 2, 3: c, d
    return wave_sum_rows(z);          // a..d in lanes 60..63
}
// Which of the 29 sums a refit pass accumulates (hg_refit_point): a function of constants, so that the indices fold into the code
// (as a table in memory they are scalar loads in the refining workgroup's chain)
//   pass 0: sums 0..5 (x, y only), 21, 22, 24, 25 (x u, y u, x v, y v);  pass 1: 6..11, 23, 26 (u, v), 27, 28;  pass 2: 12..17, 18..20
__host__ __device__ constexpr int hg_pass_sum(int pass, int q)
{
    return pass == 0 ? (q < 6 ? q : (q == 6 ? 21 : (q == 7 ? 22 : (q == 8 ? 24 : 25))))
         : pass == 1 ? (q < 6 ? 6 + q : (q == 6 ? 23 : (q == 7 ? 26 : (q == 8 ? 27 : 28))))
         : 12 + q;
}
template <int PASS> constexpr int hg_pass_n() { return PASS == 2 ? 9 : 10; }
static_assert(hg_pass_sum(0, 9) == 25 && hg_pass_sum(1, 0) == 6 && hg_pass_sum(1, 9) == 28 && hg_pass_sum(2, 8) == 20, "the three passes cover the 29 sums");
// wave totals of a pass's sums acc[0..N-1] into w[sum][wv]: quads, then a pair or a single one
template <int PASS, int N>
__device__ __forceinline__ void wave_totals_to_lds(const double (&acc)[N], double (*w)[4], uint32_t lane, uint32_t wv)
{
    asm volatile("" : "+v"(lane));    // (the lane-bit masks are made here, not kept in scalar registers across the kernel)
    const bool b0 = lane & 1u, b1 = lane & 2u;
    constexpr int Q = N / 4, R = N % 4;
    static_assert(R != 3, "quads, then a pair or a single sum");
#pragma unroll
    for (int g = 0; g < Q; ++g) {
        const double z = wave_sum_quad(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], b0, b1);
        const int q = b1 ? (b0 ? hg_pass_sum(PASS, 4 * g + 3) : hg_pass_sum(PASS, 4 * g + 2)) : (b0 ? hg_pass_sum(PASS, 4 * g + 1) : hg_pass_sum(PASS, 4 * g));
        if (lane >= 60u) w[q][wv] = z;
    }
    if (R == 2) {
        const double z = wave_sum_pair(acc[4 * Q], acc[4 * Q + 1], b0);
        if (lane >= 62u) w[b0 ? hg_pass_sum(PASS, 4 * Q + 1) : hg_pass_sum(PASS, 4 * Q)][wv] = z;
    } else if (R == 1) {
        wave_total_to_lds(acc[4 * Q], w, hg_pass_sum(PASS, 4 * Q), lane, wv);
    }
}

// LDL^T solve of the 8x8 normal equations on lanes 0..7 of one wave: lane i owns row i of M and of L; every element sees exactly
// the operations of the sequential algorithm (oracle/vxoracle_homography.c, hg_fit), in the same order.  The 29 sums are read
// from LDS (w[q][0..3]: the four wave totals of sum q) by the lane that needs them.
__device__ __forceinline__ bool hg_solve_lanes_lds(const double (*w)[4], uint32_t lane, double (&h)[8])
{
    const int i = (int)(lane & 7u);           // lanes 8..63 shadow rows 0..7 (their results are never used)
    auto S = [&](int q) { return (w[q][0] + w[q][1]) + (w[q][2] + w[q][3]); };
    double Lrow[8], D[8], Dself = 1.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) Lrow[c] = 0.0;
    const double brow = S(21 + i);
    bool pd = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = hg_midx(i, j <= i ? j : i);           // element (i, j) of the lower triangle; 0 above it and where hg_midx says so
        const double m = S(q >= 0 ? q : 0);
        double t = (j <= i && q >= 0) ? m : 0.0;
#pragma unroll
        for (int k = 0; k < j; ++k) t -= (Lrow[k] * lane_bcast(Lrow[k], j)) * D[k];
        const double d = lane_bcast(t, j);
        pd = pd && (d > 1e-12);
        D[j] = d;
        const double l = t / d;
        Lrow[j] = lane > (uint32_t)j ? l : 0.0;
        if (lane == (uint32_t)j) Dself = d;
    }
    double z = brow;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double zk = lane_bcast(z, k);
        if (lane > (uint32_t)k) z -= Lrow[k] * zk;
    }
    const double s0 = z / Dself;
#pragma unroll
    for (int r_ = 7; r_ >= 0; --r_) {
        double r = s0;
#pragma unroll
        for (int k = r_ + 1; k < 8; ++k) r -= lane_bcast(Lrow[r_], k) * h[k];     // L[k][r_] lives in row k, on lane k
        h[r_] = lane_bcast(r, r_);
    }
    return pd;
}

__device__ __forceinline__ void hg_fit_publish_lds(const double (*w)[4], const HgNorm &nm, uint32_t lane, float *Hnew, int *fit_ok)
{
    // a copy of the lane number the compiler cannot see through: the solve's per-lane indices and lane masks are loop invariants of
    // the refit loop, and hoisted out of it they hold dozens of registers across the passes over the points
    asm volatile("" : "+v"(lane));
    double h[8];
    bool ok = hg_solve_lanes_lds(w, lane, h);
    const double Hn[9] = { h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 1.0 };
    double G[9], Hd[9];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        G[3 * rr] = Hn[3 * rr] * nm.ss;
        G[3 * rr + 1] = Hn[3 * rr + 1] * nm.ss;
        G[3 * rr + 2] = (Hn[3 * rr + 2] - G[3 * rr] * nm.cx) - G[3 * rr + 1] * nm.cy;
    }
    const double g6[6] = { G[0], G[1], G[2], G[3], G[4], G[5] };
    const double gq = lane_select<6>(g6, lane) / nm.sd;      // G[c] / sd, c = 0..5, in one division
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Hd[c] = lane_bcast(gq, c) + nm.cu * G[6 + c];
        Hd[3 + c] = lane_bcast(gq, 3 + c) + nm.cv * G[6 + c];
        Hd[6 + c] = G[6 + c];
    }
    float e;
    ok = hg_to_f32_lanes(Hd, lane, e) && ok;
    if (lane < 9) Hnew[lane] = ok ? e : (lane % 4u == 0u ? 1.0f : 0.0f);
    if (lane == 0) *fit_ok = ok ? 1 : 0;
}

// The source points (32 KB) are the launch's DYNAMIC LDS: declared statically, the compiler sees that LDS allows four such
// workgroups per compute unit, concludes that fewer than 97 registers buy no occupancy, and takes 97 whatever it is asked for --
// one more than fits beside four PyrLK waves.
constexpr size_t HG_LEAN_DYN_LDS = (size_t)HG_NPT * HG_T * sizeof(float2);
struct HgLeanLds {                      // the static part
    double wsum[HG_NACC][4];
    unsigned long long best_w[4];
    float hyp_s[HG_KPRE * 9];
    float Hs[9], Hnew[9], Hcur[9];
    uint32_t red[4];
    int valid, fit_ok;
    uint32_t is_last;
};

// One pass of the refit: a third of the 29 sums, each one the same chain of fused multiply-adds over a thread's points in the same
// order as in a one-pass form (two passes of 15 + 14 sums do not fit in the registers beside sixteen destination points)
// (which sums a pass takes: hg_pass_sum above)
// ... one correspondence's products
template <int PASS, int N>
__device__ __forceinline__ void hg_refit_point(double (&acc)[N], float sx, float sy, float dx, float dy, const HgNorm &nm)
{
    // opaque copies: keeps the compiler from hoisting the points' binary64 terms out of the pass (and across passes)
    asm volatile("" : "+v"(sx), "+v"(sy), "+v"(dx), "+v"(dy));
    const double x = ((double)sx - nm.cx) * nm.ss, y = ((double)sy - nm.cy) * nm.ss;
    const double u = ((double)dx - nm.cu) * nm.sd, v = ((double)dy - nm.cv) * nm.sd;
#define ACC(k, p, q) acc[k] = fma(p, q, acc[k])
    if (PASS == 0) {
        ACC(0, x, x); ACC(1, x, y); ACC(2, x, 1.0); ACC(3, y, y); ACC(4, y, 1.0); ACC(5, 1.0, 1.0);
        ACC(6, x, u); ACC(7, y, u); ACC(8, x, v); ACC(9, y, v);
    } else if (PASS == 1) {
        const double q0 = -(x * u), q1 = -(y * u), r0 = -(x * v), r1 = -(y * v);
        ACC(0, x, q0); ACC(1, x, q1); ACC(2, y, q0); ACC(3, y, q1); ACC(4, 1.0, q0); ACC(5, 1.0, q1);
        ACC(6, 1.0, u); ACC(7, 1.0, v);
        ACC(8, q0, u); ACC(8, r0, v); ACC(9, q1, u); ACC(9, r1, v);
    } else {
        const double q0 = -(x * u), q1 = -(y * u), r0 = -(x * v), r1 = -(y * v);
        ACC(0, x, r0); ACC(1, x, r1); ACC(2, y, r0); ACC(3, y, r1); ACC(4, 1.0, r0); ACC(5, 1.0, r1);
        ACC(6, q0, q0); ACC(6, r0, r0); ACC(7, q0, q1); ACC(7, r0, r1); ACC(8, q1, q1); ACC(8, r1, r1);
    }
#undef ACC
}
// ... over the inliers among a thread's sixteen on-chip correspondences
template <int PASS, typename LDS>
__device__ __forceinline__ void hg_refit_pass(const LDS &L, const float2 (&pd)[HG_NPT], uint32_t inl, const HgNorm &nm,
                                              uint32_t tid, uint32_t lane, uint32_t wv, double (*w)[4])
{
    constexpr int N = hg_pass_n<PASS>();
    double acc[N];
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = 0.0;
#pragma unroll
    for (int j = 0; j < HG_NPT; ++j) {
        if (!((inl >> j) & 1u)) continue;
        const float2 s = L.src[tid + (uint32_t)j * HG_T];
        hg_refit_point<PASS>(acc, s.x, s.y, pd[j].x, pd[j].y, nm);
    }
    wave_totals_to_lds<PASS, N>(acc, w, lane, wv);
    __builtin_amdgcn_sched_barrier(0);      // the passes one after the other: interleaved they would all hold their sums at once
}
// A thread's correspondences in memory -- items t, t + 256, ... in that order -- B at a time with their loads in flight together.
// Items past n come back as lost points (-1, -1), which nothing counts.  (Left to the compiler, these loops are either unrolled
// with everything they touch held in registers, 244 of them, or not at all, with one trip to memory per item.)
template <int B, typename F>
__device__ __forceinline__ void hg_for_each_point(const HgArgs &a, uint32_t n, uint32_t tid, F &&f)
{
    asm volatile("" : "+v"(tid));     // (every walk derives its addresses afresh: hoisted, they are held across the whole refinement)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)      // (interleaved, the counting walks hold two copies of their matrices)
    for (uint32_t i = tid; i < n; i += (uint32_t)B * HG_T) {
        float2 s[B], d[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const uint32_t idx = i + (uint32_t)j * HG_T, c = min(idx, n - 1u);
            s[j] = a.src[c];
            d[j] = a.dst[c];
            if (idx >= n) s[j] = d[j] = make_float2(-1.0f, -1.0f);
        }
#pragma unroll
        for (int j = 0; j < B; ++j) f(i + (uint32_t)j * HG_T, s[j], d[j]);
    }
}
// ... over the inliers of H among the correspondences in memory
template <int PASS>
__device__ __forceinline__ void hg_refit_pass_stream(const HgArgs &a, const float (&H)[9], uint32_t n, const HgNorm &nm,
                                                     uint32_t tid, uint32_t lane, uint32_t wv, double (*w)[4])
{
    constexpr int N = hg_pass_n<PASS>();
    double acc[N];
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = 0.0;
    hg_for_each_point<4>(a, n, tid, [&](uint32_t, float2 s, float2 d) {
        if (hg_usable(s, d) && hg_inlier(H, s, d, a.thr2)) hg_refit_point<PASS>(acc, s.x, s.y, d.x, d.y, nm);
    });
    wave_totals_to_lds<PASS, N>(acc, w, lane, wv);
    __builtin_amdgcn_sched_barrier(0);
}

template <typename LDS>
__device__ __forceinline__ void hg_refine_lean(const HgArgs &a, LDS &L, const float2 (&pd)[HG_NPT], uint32_t us, uint32_t n)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    // the mask's item count is known already (written here, the two pointers are not carried through the refinement)
    if (tid == 0 && a.mask_count) {
        *a.mask_count = n;
        if (a.mask_count_host) *a.mask_count_host = n;
    }
    // all hypotheses travel with the keys, so the winner's matrix is not one more dependent trip to memory
    // (keys and hypotheses were published by workgroups on other XCDs: read past this XCD's L2, see find_homography_lean_wg)
    const bool pre = a.K <= (uint32_t)HG_KPRE;
    if (pre)
        for (uint32_t e = tid; e < 9u * a.K; e += HG_T) L.hyp_s[e] = __hip_atomic_load(&hg_hyp(a)[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // best hypothesis: most inliers, ties to the lowest index
    unsigned long long best = 0;
    for (uint32_t k = tid; k < a.K; k += HG_T) {
        const uint32_t kk = __hip_atomic_load(&hg_keys(a)[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long key = ((unsigned long long)kk << 32) | (unsigned long long)(0xffffffffu - k);
        best = key > best ? key : best;
    }
    uint32_t usable = wave_sum63((uint32_t)__popc(us));
    best = wave_max63(best);
    HG_STAMP_ANY(1);
    if (lane == 63u) { L.best_w[wv] = best; L.red[wv] = usable; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) best = L.best_w[w] > best ? L.best_w[w] : best;
    usable = L.red[0] + L.red[1] + L.red[2] + L.red[3];
    const uint32_t key = (uint32_t)(best >> 32), kbest = 0xffffffffu - (uint32_t)(best & 0xffffffffu);

    // the current matrix lives in LDS (L.Hcur, written by threads 0..8) and is in registers only where it is used; the barriers
    // of the sums below stand between its writes and its reads
    if (tid < 9) L.Hcur[tid] = (tid % 4u == 0u) ? 1.0f : 0.0f;
    // from here on `us` carries the inlier bits of the current matrix in its upper half (one register for both sets)
    uint32_t cnt = 0;
    if (usable >= 4 && key >= 5) {      // a valid hypothesis with at least 4 inliers
        cnt = key - 1u;
        if (tid < 9) L.Hcur[tid] = pre ? L.hyp_s[9 * kbest + tid] : __hip_atomic_load(&hg_hyp(a)[9 * kbest + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        HG_STAMP_ANY(2);
        // normalisation over all usable correspondences (two passes, interleaved partials)
        double c4[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int j = 0; j < HG_NPT; ++j) {
            const float2 s = L.src[tid + (uint32_t)j * HG_T];      // (read whether used or not: the reads go out together)
            if ((us >> j) & 1u) { c4[0] += (double)s.x; c4[1] += (double)s.y; c4[2] += (double)pd[j].x; c4[3] += (double)pd[j].y; }
        }
        block_tree_n<4>(c4, L.wsum);
        block_tree_read<4>(c4, L.wsum);
        const double cntd = (double)usable;
        HgNorm nm;
        nm.cx = uniform64(c4[0] / cntd); nm.cy = uniform64(c4[1] / cntd); nm.cu = uniform64(c4[2] / cntd); nm.cv = uniform64(c4[3] / cntd);
        HG_STAMP_ANY(3);
        double d2[2] = { 0, 0 };
#pragma unroll
        for (int j = 0; j < HG_NPT; ++j) {
            const float2 s = L.src[tid + (uint32_t)j * HG_T];
            if ((us >> j) & 1u) {
                d2[0] += fabs((double)s.x - nm.cx) + fabs((double)s.y - nm.cy);
                d2[1] += fabs((double)pd[j].x - nm.cu) + fabs((double)pd[j].y - nm.cv);
            }
        }
        block_tree_n<2>(d2, L.wsum);
        block_tree_read<2>(d2, L.wsum);
        HG_STAMP_ANY(4);
        const bool norm_ok = d2[0] > 0.0 && d2[1] > 0.0;
        nm.ss = uniform64(norm_ok ? cntd / d2[0] : 0.0);
        nm.sd = uniform64(norm_ok ? cntd / d2[1] : 0.0);
        {
            float H[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) H[i] = L.Hcur[i];
#pragma unroll
            for (int j = 0; j < HG_NPT; ++j)
                us |= (((us >> j) & 1u) & (uint32_t)hg_inlier(H, L.src[tid + (uint32_t)j * HG_T], pd[j], a.thr2)) << (16 + j);
        }

        HG_STAMP_ANY(5);
        for (uint32_t r = 0; r < a.max_refine && norm_ok; ++r) {
            HG_STAMP_ANY(8 + 6 * (r < 3 ? r : 3));
            __syncthreads();      // the previous round's readers are done with the sums
            hg_refit_pass<0>(L, pd, us >> 16, nm, tid, lane, wv, L.wsum);
            hg_refit_pass<1>(L, pd, us >> 16, nm, tid, lane, wv, L.wsum);
            hg_refit_pass<2>(L, pd, us >> 16, nm, tid, lane, wv, L.wsum);
            HG_STAMP_ANY(9 + 6 * (r < 3 ? r : 3));
            __syncthreads();
            HG_STAMP_ANY(10 + 6 * (r < 3 ? r : 3));
            if (wv == 0) hg_fit_publish_lds(L.wsum, nm, lane, L.Hnew, &L.fit_ok);
            HG_STAMP_ANY(12 + 6 * (r < 3 ? r : 3));
            __syncthreads();
            if (!L.fit_ok) break;     // block-uniform
            float Hn[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Hn[i] = L.Hnew[i];
            uint32_t is = 0;
#pragma unroll
            for (int j = 0; j < HG_NPT; ++j)
                is |= (((us >> j) & 1u) & (uint32_t)hg_inlier(Hn, L.src[tid + (uint32_t)j * HG_T], pd[j], a.thr2)) << j;
            uint32_t c = __popc(is), diff = __popc(is ^ (us >> 16));
            c = block_sum_u32(c | (diff << 16), L.red);     // both counts are at most 4096: one sum, 16 bits each
            diff = c >> 16;
            c &= 0xffffu;
            HG_STAMP_ANY(13 + 6 * (r < 3 ? r : 3));
            if (c < cnt) break;
            if (tid < 9) L.Hcur[tid] = L.Hnew[tid];     // (Hnew is next written after the next round's barriers)
            cnt = c;
            us = (us & 0xffffu) | (is << 16);
            if (diff == 0) break;
        }
    }
    HG_STAMP_ANY(6);
    // (an opaque copy of the thread number for the write-out: the sixteen "i < n" lane masks, "tid < 9" and "tid == 0" would otherwise
    // be kept in scalar registers from the kernel's head to here)
    uint32_t t2 = tid;
    asm volatile("" : "+v"(t2));
    uint8_t *mask = a.mask;
    asm volatile("" : "+s"(mask));      // (likewise: "is there a mask" is first asked where the point count is read)
    if (mask) {
#pragma unroll
        for (int j = 0; j < HG_NPT; ++j) {
            const uint32_t i = t2 + (uint32_t)j * HG_T;
            if (i < n) mask[i] = (cnt != 0 && ((us >> (16 + j)) & 1u)) ? 255 : 0;
        }
    }
    __syncthreads();      // L.Hcur's last update
    if (t2 < 9) {   // transposed on the way out (homography_filter_node.cpp:87-91)
        const float v = L.Hcur[3u * (t2 % 3u) + t2 / 3u];
        a.matrix[t2] = v;
        if (a.matrix_host) a.matrix_host[t2] = v;
    }
    HG_STAMP_ANY(7);
}

__device__ __forceinline__ void find_homography_lean_wg(const HgArgs &a, uint32_t bound, const uint32_t k)
{
    __shared__ HgLeanLds Ls;
    extern __shared__ __attribute__((aligned(16))) float2 hg_src_points[];
    struct View {       // what the functions above see: the static part by reference, the points through the dynamic array
        float2 *src; double (&wsum)[HG_NACC][4]; unsigned long long (&best_w)[4]; float (&hyp_s)[HG_KPRE * 9];
        float (&Hs)[9], (&Hnew)[9], (&Hcur)[9]; uint32_t (&red)[4]; int &valid, &fit_ok; uint32_t &is_last;
    };
    View L{ hg_src_points, Ls.wsum, Ls.best_w, Ls.hyp_s, Ls.Hs, Ls.Hnew, Ls.Hcur, Ls.red, Ls.valid, Ls.fit_ok, Ls.is_last };
    const uint32_t tid = threadIdx.x;
    HG_STAMP(32);
    // The loads do not wait for the item counts: the host guarantees bound <= HG_NPT * HG_T <= the arrays' capacities.  Source points go
    // from memory straight into LDS (global_load_lds, 16 bytes = two points per lane, no registers in between); what lies at and
    // beyond `bound` is never read back (the usable test below starts with i < n <= bound).
    {
        const uint32_t wave_first = tid & ~63u;
#pragma unroll
        for (int m = 0; m < HG_NPT / 2; ++m) {
            const uint32_t f = (uint32_t)m * HG_T + tid;                 // this lane's pair of points: 2 f, 2 f + 1
            if (2u * f + 1u < bound)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const float4 *>(a.src) + f),
                                                 (__attribute__((address_space(3))) void *)(L.src + 2u * ((uint32_t)m * HG_T + wave_first)), 16, 0, 0);
            else if (2u * f < bound) L.src[2u * f] = a.src[2u * f];
        }
    }
    float2 pd[HG_NPT];
#pragma unroll
    for (int j = 0; j < HG_NPT; ++j) {
        const uint32_t i = tid + (uint32_t)j * HG_T;
        pd[j] = make_float2(-1.0f, -1.0f);
        if (i < bound) pd[j] = a.dst[i];
    }
    const uint32_t n = hg_n(a);
    HG_STAMP(33);
    if (tid < 64) {
        float e;
        const bool ok = hg_make_hypothesis(a, k, n, tid, e);
        if (tid < 9) L.Hs[tid] = ok ? e : (tid % 4u == 0u ? 1.0f : 0.0f);
        if (tid == 0) L.valid = ok ? 1 : 0;
    }
    HG_STAMP(34);
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this wave's part of the source points is in LDS (the other waves': the barrier)
    __syncthreads();
    HG_STAMP(35);
    float H[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = L.Hs[i];
    const bool valid = L.valid != 0;
    uint32_t us = 0, cnt = 0;
    uint32_t t1 = tid;
    asm volatile("" : "+v"(t1));      // (the points' indices are re-derived here, not held in sixteen registers since the loads above)
#pragma unroll
    for (int j = 0; j < HG_NPT; ++j) {
        const float2 s = L.src[t1 + (uint32_t)j * HG_T];
        const uint32_t u = (uint32_t)(t1 + (uint32_t)j * HG_T < n) & (uint32_t)hg_usable(s, pd[j]);     // (no branch: the sixteen reads go out together)
        us |= u << j;
        cnt += u & (uint32_t)hg_inlier(H, s, pd[j], a.thr2);
    }
    HG_STAMP(36);
    cnt = block_sum_u32(valid ? cnt : 0u, L.red);
    // Publish, then take a ticket.  The other workgroups may sit on other XCDs, behind other L2s.  A release / acquire pair at
    // device scope would do, but on this GPU it is a write-back of the XCD's whole L2 in every workgroup and an invalidation of it
    // in the last -- with the other camera streams' PyrLK waves living on their tiles in those L2s (57 such operations per
    // frame).  Instead the ten words a workgroup publishes and the reads of them go past the L2s one by one (device-scope relaxed
    // atomics: write-through stores, loads that do not hit a stale line), and the order is kept by hand: every thread waits for
    // its own stores to be acknowledged (the workgroup-scope release of the barrier) before thread 0 takes the ticket.
    if (tid == 0) __hip_atomic_store(&hg_keys(a)[k], valid ? cnt + 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 9) __hip_atomic_store(&hg_hyp(a)[9 * k + tid], L.Hs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);       // vmcnt = lgkmcnt = expcnt = 0: the stores above have been acknowledged
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.is_last = (t == a.K - 1u) ? 1u : 0u;
        if (t == a.K - 1u) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    HG_STAMP(37);
    if (!L.is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    HG_STAMP_ANY(0);
    hg_refine_lean(a, L, pd, us, n);
}

// ---- arrays of any size: the same steps with the correspondences streamed from memory in every pass -----------------------------
// (the bits a thread's register holds above -- usable, inlier under the current matrix -- are recomputed where they are needed)
__device__ __forceinline__ void hg_refine_stream(const HgArgs &a, HgLeanLds &L, uint32_t n)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0 && a.mask_count) {
        *a.mask_count = n;
        if (a.mask_count_host) *a.mask_count_host = n;
    }
    // (keys and hypotheses were published by workgroups on other XCDs: read past this XCD's L2, see find_homography_lean_wg)
    const bool pre = a.K <= (uint32_t)HG_KPRE;
    if (pre)
        for (uint32_t e = tid; e < 9u * a.K; e += HG_T) L.hyp_s[e] = __hip_atomic_load(&hg_hyp(a)[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long best = 0;
    for (uint32_t k = tid; k < a.K; k += HG_T) {
        const uint32_t kk = __hip_atomic_load(&hg_keys(a)[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long key = ((unsigned long long)kk << 32) | (unsigned long long)(0xffffffffu - k);
        best = key > best ? key : best;
    }
    uint32_t usable = 0;
    hg_for_each_point<4>(a, n, tid, [&](uint32_t, float2 s, float2 d) { usable += hg_usable(s, d) ? 1u : 0u; });
    usable = wave_sum63(usable);
    best = wave_max63(best);
    if (lane == 63u) { L.best_w[wv] = best; L.red[wv] = usable; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) best = L.best_w[w] > best ? L.best_w[w] : best;
    usable = L.red[0] + L.red[1] + L.red[2] + L.red[3];
    const uint32_t key = (uint32_t)(best >> 32), kbest = 0xffffffffu - (uint32_t)(best & 0xffffffffu);

    // the current matrix lives in LDS (L.Hcur) and is in registers only where it is used
    if (tid < 9) L.Hcur[tid] = (tid % 4u == 0u) ? 1.0f : 0.0f;
    uint32_t cnt = 0;
    if (usable >= 4 && key >= 5) {      // a valid hypothesis with at least 4 inliers
        cnt = key - 1u;
        if (tid < 9) L.Hcur[tid] = pre ? L.hyp_s[9 * kbest + tid] : __hip_atomic_load(&hg_hyp(a)[9 * kbest + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // normalisation over all usable correspondences (two passes, interleaved partials)
        double c4[4] = { 0, 0, 0, 0 };
        hg_for_each_point<4>(a, n, tid, [&](uint32_t, float2 s, float2 d) {
            if (hg_usable(s, d)) { c4[0] += (double)s.x; c4[1] += (double)s.y; c4[2] += (double)d.x; c4[3] += (double)d.y; }
        });
        block_tree_n<4>(c4, L.wsum);
        block_tree_read<4>(c4, L.wsum);
        const double cntd = (double)usable;
        HgNorm nm;
        nm.cx = uniform64(c4[0] / cntd); nm.cy = uniform64(c4[1] / cntd); nm.cu = uniform64(c4[2] / cntd); nm.cv = uniform64(c4[3] / cntd);
        double d2[2] = { 0, 0 };
        hg_for_each_point<4>(a, n, tid, [&](uint32_t, float2 s, float2 d) {
            if (hg_usable(s, d)) {
                d2[0] += fabs((double)s.x - nm.cx) + fabs((double)s.y - nm.cy);
                d2[1] += fabs((double)d.x - nm.cu) + fabs((double)d.y - nm.cv);
            }
        });
        block_tree_n<2>(d2, L.wsum);
        block_tree_read<2>(d2, L.wsum);
        const bool norm_ok = d2[0] > 0.0 && d2[1] > 0.0;
        nm.ss = uniform64(norm_ok ? cntd / d2[0] : 0.0);
        nm.sd = uniform64(norm_ok ? cntd / d2[1] : 0.0);

        for (uint32_t r = 0; r < a.max_refine && norm_ok; ++r) {
            __syncthreads();      // L.Hcur is written; the previous round's readers are done with the sums
            {
                float H[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) H[i] = L.Hcur[i];
                hg_refit_pass_stream<0>(a, H, n, nm, tid, lane, wv, L.wsum);
                hg_refit_pass_stream<1>(a, H, n, nm, tid, lane, wv, L.wsum);
                hg_refit_pass_stream<2>(a, H, n, nm, tid, lane, wv, L.wsum);
            }
            __syncthreads();
            if (wv == 0) hg_fit_publish_lds(L.wsum, nm, lane, L.Hnew, &L.fit_ok);
            __syncthreads();
            if (!L.fit_ok) break;     // block-uniform
            float H[9], Hn[9];        // (the current matrix is read again rather than held in registers across the solve)
#pragma unroll
            for (int i = 0; i < 9; ++i) { H[i] = L.Hcur[i]; Hn[i] = L.Hnew[i]; }
            uint32_t c = 0, diff = 0;
            hg_for_each_point<4>(a, n, tid, [&](uint32_t, float2 s, float2 d) {
                const bool us = hg_usable(s, d);
                const bool was = us && hg_inlier(H, s, d, a.thr2), now = us && hg_inlier(Hn, s, d, a.thr2);
                c += now ? 1u : 0u;
                diff += (was != now) ? 1u : 0u;
            });
            c = block_sum_u32(c, L.red);
            diff = block_sum_u32(diff, L.red);
            if (c < cnt) break;
            if (tid < 9) L.Hcur[tid] = L.Hnew[tid];     // (its readers passed the barriers of the two sums above)
            cnt = c;
            if (diff == 0) break;
        }
    }
    __syncthreads();      // L.Hcur's last update
    if (a.mask) {
        float H[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) H[i] = L.Hcur[i];
        hg_for_each_point<4>(a, n, tid, [&](uint32_t i, float2 s, float2 d) {
            if (i < n) a.mask[i] = (cnt != 0 && hg_usable(s, d) && hg_inlier(H, s, d, a.thr2)) ? 255 : 0;
        });
    }
    if (tid < 9) {   // transposed on the way out (homography_filter_node.cpp:87-91)
        const float v = L.Hcur[3u * (tid % 3u) + tid / 3u];
        a.matrix[tid] = v;
        if (a.matrix_host) a.matrix_host[tid] = v;
    }
}

__device__ __forceinline__ void find_homography_stream_wg(const HgArgs &a, const uint32_t k)
{
    __shared__ HgLeanLds L;
    const uint32_t tid = threadIdx.x;
    const uint32_t n = hg_n(a);
    if (tid < 64) {
        float e;
        const bool ok = hg_make_hypothesis(a, k, n, tid, e);
        if (tid < 9) L.Hs[tid] = ok ? e : (tid % 4u == 0u ? 1.0f : 0.0f);
        if (tid == 0) L.valid = ok ? 1 : 0;
    }
    __syncthreads();
    float H[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = L.Hs[i];
    const bool valid = L.valid != 0;
    uint32_t cnt = 0;
    if (valid)
        hg_for_each_point<4>(a, n, tid, [&](uint32_t, float2 s, float2 d) { cnt += (hg_usable(s, d) && hg_inlier(H, s, d, a.thr2)) ? 1u : 0u; });
    cnt = block_sum_u32(cnt, L.red);
    // publish past the L2s, then take a ticket (see find_homography_lean_wg)
    if (tid == 0) __hip_atomic_store(&hg_keys(a)[k], valid ? cnt + 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 9) __hip_atomic_store(&hg_hyp(a)[9 * k + tid], L.Hs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.is_last = (t == a.K - 1u) ? 1u : 0u;
        if (t == a.K - 1u) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!L.is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // compiler: nothing below moves above the ticket
    hg_refine_stream(a, L, n);
}


// 48: the attribute counts the unified register file in halves (pyrlk.hip)
__global__ __launch_bounds__(HG_T) __attribute__((amdgpu_num_vgpr(48))) void k_find_homography_lean(HgArgs a, uint32_t bound)
{
    find_homography_lean_wg(a, bound, blockIdx.x);
}
// ... for arrays of any size
__global__ __launch_bounds__(HG_T) __attribute__((amdgpu_num_vgpr(48))) void k_find_homography(HgArgs a)
{
    find_homography_stream_wg(a, blockIdx.x);
}
constexpr uint32_t HG_MERGE_KT = 8, HG_MERGE_KC = 13;   // rounds per block of the merge's walks (four waves, up to 96 registers)
// ... and with a detector node's merge pass as workgroup 0 (k_find_homography_merge below is the any-size form)
__global__ __launch_bounds__(HG_T) __attribute__((amdgpu_num_vgpr(48))) void k_find_homography_lean_merge(HgArgs a, uint32_t bound, MergeArgs m)
{
    if (blockIdx.x == 0) {
        VSA_SHORT_KERNEL_PRIORITY();
        merge_compact_wg<(uint32_t)HG_T, HG_MERGE_KT, HG_MERGE_KC>(m);
        return;
    }
    find_homography_lean_wg(a, bound, blockIdx.x - 1u);
}

// FindHomography and a detector node's merge pass in one launch (launch_find_homography_with_merge): both only read what PyrLK
// (and, for the merge, the cell pass) left, neither reads what the other writes.  Workgroup 0 runs the merge on its four waves
// (merge_cells.hpp) and is long done when the refinement ends; workgroups 1 .. K are the hypotheses.  What a camera stream saves is
// the merge's launch: its wait for a compute unit with room, its run and the gap in front of it, while the stream's hardware
// queue is held by the homography's latency chain anyway.
__global__ __launch_bounds__(HG_T) __attribute__((amdgpu_num_vgpr(48))) void k_find_homography_merge(HgArgs a, MergeArgs m)
{
    if (blockIdx.x == 0) {
        VSA_SHORT_KERNEL_PRIORITY();
        merge_compact_wg<(uint32_t)HG_T, HG_MERGE_KT, HG_MERGE_KC>(m);
        return;
    }
    find_homography_stream_wg(a, blockIdx.x - 1u);
}

}  // namespace

uint32_t homography_hypotheses(float confidence, float outlier_ratio, uint32_t max_estimate_iters)
{
    double w = 1.0 - (double)outlier_ratio;
    if (!(w >= 0.01)) w = 0.01;
    if (w > 1.0) w = 1.0;
    double c = (double)confidence;
    if (!(c >= 0.0)) c = 0.0;
    if (c > 0.999999) c = 0.999999;
    const double p = w * w * w * w;
    double k = 1.0;
    if (p < 1.0) k = std::ceil(std::log(1.0 - c) / std::log(1.0 - p));
    if (!(k >= 1.0)) k = 1.0;
    if (max_estimate_iters < 1) max_estimate_iters = 1;
    if (k > (double)max_estimate_iters) k = (double)max_estimate_iters;
    return (uint32_t)k;
}

size_t homography_scratch_bytes(uint32_t hypotheses)
{
    return (size_t)hypotheses * (sizeof(uint32_t) + 9 * sizeof(float)) + 32;     // ticket word (16 B), keys padded to 16 B, hypotheses
}

namespace {
// kernel arguments for p; `bound`: what the arrays' capacities allow the point count to be (it lives in device memory)
bool hg_setup(const HomographyParams &p, HgArgs &a, uint32_t &bound)
{
    if (p.hypotheses == 0) return false;
    a = HgArgs{};
    a.src = static_cast<const float2 *>(p.src); a.dst = static_cast<const float2 *>(p.dst);
    a.src_count = p.src_count; a.dst_count = p.dst_count; a.src_cap = p.src_capacity; a.dst_cap = p.dst_capacity;
    a.thr2 = p.threshold * p.threshold;
    a.K = p.hypotheses; a.max_refine = p.max_refine_iters;
    a.matrix = p.matrix; a.matrix_host = p.matrix_host;
    a.mask = p.mask; a.mask_count = p.mask_count; a.mask_count_host = p.mask_count_host; a.mask_cap = p.mask_capacity;
    a.ticket = static_cast<uint32_t *>(p.scratch);          // at a fixed place, whatever the hypothesis count
    bound = p.src_capacity < p.dst_capacity ? p.src_capacity : p.dst_capacity;
    if (p.mask && p.mask_capacity < bound) bound = p.mask_capacity;
    return true;
}
// the any-size kernel: arrays that may hold more than HG_NPT * 256 correspondences, or a source array that is not 16-byte aligned
// (the on-chip kernel fetches it in 16-byte pieces; arrays the runtime allocates always are)
bool hg_generic(const HomographyParams &p, uint32_t bound)
{
    const bool force_generic = getenv("VISUALSLAM_AMD_HG_GENERIC") != nullptr;   // tests: exercise the any-size kernel
    return bound > (uint32_t)HG_NPT * HG_T || (reinterpret_cast<uintptr_t>(p.src) & 15u) != 0 || force_generic;
}
constexpr size_t HG_MERGE_MAX_LDS = 32u * 1024u;   // the merge's cell flags, beside the refinement's static arrays
}  // namespace

hipError_t launch_find_homography(const HomographyParams &p, hipStream_t stream)
{
    HgArgs a{};
    uint32_t bound = 0;
    if (!hg_setup(p, a, bound)) return hipErrorInvalidValue;
    if (!hg_generic(p, bound)) hipLaunchKernelGGL(k_find_homography_lean, dim3(p.hypotheses), dim3(HG_T), HG_LEAN_DYN_LDS, stream, a, bound);
    else hipLaunchKernelGGL(k_find_homography, dim3(p.hypotheses), dim3(HG_T), 0, stream, a);
    return hipGetLastError();
}

bool find_homography_merge_fusable(const HomographyParams &p, const CornerTrackParams &ct)
{
    HgArgs a{};
    MergeArgs m{};
    uint32_t bound = 0;
    size_t lds = 0;
    return hg_setup(p, a, bound) && merge_setup(ct, m, lds) && lds <= HG_MERGE_MAX_LDS;
}

hipError_t launch_find_homography_with_merge(const HomographyParams &p, const CornerTrackParams &ct, hipStream_t stream)
{
    HgArgs a{};
    MergeArgs m{};
    uint32_t bound = 0;
    size_t lds = 0;
    if (!hg_setup(p, a, bound) || !merge_setup(ct, m, lds) || lds > HG_MERGE_MAX_LDS) return hipErrorInvalidValue;
    const dim3 grid(p.hypotheses + 1u);
    if (!hg_generic(p, bound))
        hipLaunchKernelGGL(k_find_homography_lean_merge, grid, dim3(HG_T), lds > HG_LEAN_DYN_LDS ? lds : HG_LEAN_DYN_LDS, stream, a, bound, m);
    else hipLaunchKernelGGL(k_find_homography_merge, grid, dim3(HG_T), lds, stream, a, m);
    return hipGetLastError();
}

}  // namespace vsa

#ifdef VSA_HG_STAMPS
extern "C" int visualslam_amd_debug_hg_stamps(uint64_t *host, size_t n_u64)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vsa::g_hg_stamps), n_u64 * sizeof(uint64_t));
}
#endif
