// homography.hip -- nvxFindHomography (RANSAC) on the PyrLK matches.
//
// Reference call sites replaced: nvxFindHomographyNode(graph, srcPoints, dstPoints, homography, RANSAC, threshold,
// maxEstimateIters, maxRefineIters, confidence, outlierRatio, mask)  (demos/video_stabilizer/stabilizer.cpp:264-273,
// VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp:1576-1578); output conventions from the consumer
// demos/video_stabilizer/homography_filter_node.cpp:36-159.  Semantics and the exact floating-point order:
// oracle/vxoracle_homography.c -- results are bit-identical.
//
// One launch, no host round trip (k_find_homography):
//   * workgroup k evaluates hypothesis k (all hypotheses of the a-priori RANSAC count are independent, so they run as
//     one batch): wave 0 draws the 4-point sample and maps it exactly in binary64 while the other waves fetch points;
//     the workgroup then counts the hypothesis' inliers (binary32, division-free) and publishes count and matrix;
//   * the workgroup that takes the last ticket (device-scope release / acquire) goes on with its points still in
//     registers: it picks the best hypothesis, normalises coordinates, then alternates least-squares refits (29 binary64
//     sums in 256 interleaved partials + butterfly, 8x8 LDL^T) with inlier recounts; writes matrix, mask and counts
//     (also to pinned mirrors).
// The work is tiny (56 x 4096 point tests): latency, not throughput, is what matters.  Hence
//   * NPT > 0 variants, used when the arrays' capacities bound the point count by NPT*256: every point stays in
//     registers for the whole kernel, and the loads do not wait for the item counts (NPT = 0: any size, global loads);
//   * all sums of a pass go through ONE barrier pair: DPP butterfly inside each wave, four wave totals through LDS;
//   * the serial binary64 parts (sample -> hypothesis, LDL^T, denormalisation) run on wave 0 with lanes as the
//     parallel dimension: independent divisions share one instruction sequence, row i of the factorisation lives on
//     lane i.  Every element still sees the operations of the sequential algorithm in the same order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "kernels.hpp"

namespace vsa {
namespace {

#ifdef VSA_HG_STAMPS   // development instrumentation: cycle stamps of the phases (thread 0 of workgroup 0)
__device__ uint64_t g_hg_stamps[64];
#define HG_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_hg_stamps[i] = clock64(); } while (0)
#else
#define HG_STAMP(i) do { } while (0)
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
    uint32_t *keys;     // per hypothesis: inlier count + 1, or 0 when the hypothesis is invalid
    float *hyp;         // per hypothesis: 9 floats, h22 = 1
    uint32_t *ticket;   // workgroups that have published their hypothesis; 0 between launches
};

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
    // unit square -> quadrilateral for both point sets (Heckbert); the four quotients g, h share one division
    double x[4], y[4], u[4], v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = sp[j].x; y[j] = sp[j].y; u[j] = dp[j].x; v[j] = dp[j].y; }
    const double adx1 = x[1] - x[2], adx2 = x[3] - x[2], ady1 = y[1] - y[2], ady2 = y[3] - y[2];
    const double asx = ((x[0] - x[1]) + x[2]) - x[3], asy = ((y[0] - y[1]) + y[2]) - y[3];
    const double aden = adx1 * ady2 - adx2 * ady1;
    const double bdx1 = u[1] - u[2], bdx2 = u[3] - u[2], bdy1 = v[1] - v[2], bdy2 = v[3] - v[2];
    const double bsx = ((u[0] - u[1]) + u[2]) - u[3], bsy = ((v[0] - v[1]) + v[2]) - v[3];
    const double bden = bdx1 * bdy2 - bdx2 * bdy1;
    if (!(aden != 0.0) || !(bden != 0.0)) return false;
    const double num[4] = { asx * ady2 - adx2 * asy, adx1 * asy - asx * ady1, bsx * bdy2 - bdx2 * bsy, bdx1 * bsy - bsx * bdy1 };
    const double quo = lane_select<4>(num, lane) / (lane < 2u ? aden : bden);
    const double ag = lane_bcast(quo, 0), ah = lane_bcast(quo, 1), bg = lane_bcast(quo, 2), bh = lane_bcast(quo, 3);
    double A[9], B[9], adj[9], H[9];
    A[0] = (x[1] - x[0]) + ag * x[1]; A[1] = (x[3] - x[0]) + ah * x[3]; A[2] = x[0];
    A[3] = (y[1] - y[0]) + ag * y[1]; A[4] = (y[3] - y[0]) + ah * y[3]; A[5] = y[0];
    A[6] = ag; A[7] = ah; A[8] = 1.0;
    B[0] = (u[1] - u[0]) + bg * u[1]; B[1] = (u[3] - u[0]) + bh * u[3]; B[2] = u[0];
    B[3] = (v[1] - v[0]) + bg * v[1]; B[4] = (v[3] - v[0]) + bh * v[3]; B[5] = v[0];
    B[6] = bg; B[7] = bh; B[8] = 1.0;
    adj[0] = A[4] * A[8] - A[5] * A[7]; adj[1] = A[2] * A[7] - A[1] * A[8]; adj[2] = A[1] * A[5] - A[2] * A[4];
    adj[3] = A[5] * A[6] - A[3] * A[8]; adj[4] = A[0] * A[8] - A[2] * A[6]; adj[5] = A[2] * A[3] - A[0] * A[5];
    adj[6] = A[3] * A[7] - A[4] * A[6]; adj[7] = A[1] * A[6] - A[0] * A[7]; adj[8] = A[0] * A[4] - A[1] * A[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            H[3 * r + c] = (B[3 * r] * adj[c] + B[3 * r + 1] * adj[3 + c]) + B[3 * r + 2] * adj[6 + c];
    float e;
    if (!hg_to_f32_lanes(H, lane, e)) return false;
    elem = e;
    return true;
}

// The points of thread `tid`: NPT > 0 keeps points tid, tid + 256, ... in registers (the host guarantees
// n <= NPT*256 <= the arrays' capacities, so the loads need not wait for the item counts); NPT == 0 reads global memory.
template <int NPT>
struct HgPoints {
    float2 ps[NPT > 0 ? NPT : 1], pd[NPT > 0 ? NPT : 1];
    uint32_t us;      // bit j: point j of this thread exists and is usable

    __device__ __forceinline__ void fetch(const HgArgs &a, uint32_t bound)
    {
        if (NPT == 0) return;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const uint32_t i = threadIdx.x + (uint32_t)j * HG_T;
            ps[j] = pd[j] = make_float2(-1.0f, -1.0f);
            if (i < bound) { ps[j] = a.src[i]; pd[j] = a.dst[i]; }
        }
    }
    __device__ __forceinline__ void flag(uint32_t n)
    {
        us = 0;
        if (NPT == 0) return;
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            us |= (threadIdx.x + (uint32_t)j * HG_T < n && hg_usable(ps[j], pd[j])) ? (1u << j) : 0u;
    }
};

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

// LDL^T solve of the 8x8 normal equations on lanes 0..7 of one wave: lane i owns row i of M and of L; every element
// sees exactly the operations of the sequential algorithm (oracle/vxoracle_homography.c, hg_fit), in the same order.
__device__ __forceinline__ bool hg_solve_lanes(const double (&S)[HG_NACC], uint32_t lane, double (&h)[8])
{
    double Mrow[8], Lrow[8], LT[8], D[8], brow = 0.0, Dself = 1.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) { Mrow[c] = 0.0; Lrow[c] = 0.0; LT[c] = 0.0; }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane == (uint32_t)i) {
#pragma unroll
            for (int c = 0; c <= i; ++c) Mrow[c] = hg_midx(i, c) >= 0 ? S[hg_midx(i, c) >= 0 ? hg_midx(i, c) : 0] : 0.0;
            brow = S[21 + i];
        }
    bool pd = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double t = Mrow[j];
#pragma unroll
        for (int k = 0; k < j; ++k) t -= (Lrow[k] * lane_bcast(Lrow[k], j)) * D[k];
        const double d = lane_bcast(t, j);
        pd = pd && (d > 1e-12);
        D[j] = d;
        const double l = t / d;
        Lrow[j] = lane > (uint32_t)j ? l : 0.0;
        if (lane == (uint32_t)j) Dself = d;
#pragma unroll
        for (int i = j + 1; i < 8; ++i) {
            const double c = lane_bcast(l, i);
            if (lane == (uint32_t)j) LT[i] = c;
        }
    }
    double z = brow;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double zk = lane_bcast(z, k);
        if (lane > (uint32_t)k) z -= Lrow[k] * zk;
    }
    const double s0 = z / Dself;
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        double r = s0;
#pragma unroll
        for (int k = i + 1; k < 8; ++k) r -= LT[k] * h[k];
        h[i] = lane_bcast(r, i);
    }
    return pd;
}

struct HgNorm { double cx, cy, cu, cv, ss, sd; };

// wave 0, all lanes: the 29 sums -> least-squares H (h22 = 1), denormalised, rounded to binary32, published in LDS
__device__ __forceinline__ void hg_fit_publish(const double (&S)[HG_NACC], const HgNorm &nm, uint32_t lane, float *Hnew, int *fit_ok)
{
    double h[8];
    bool ok = hg_solve_lanes(S, lane, h);
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

// Picks the best hypothesis, normalises, refits; one workgroup.  P: this thread's points, already fetched and flagged.
template <int NPT>
__device__ __forceinline__ void hg_refine(const HgArgs &a, HgPoints<NPT> &P, uint32_t n)
{
    static_assert(NPT >= 0 && NPT <= 32, "one bit per point in a 32-bit mask");
    __shared__ double wsum[HG_NACC][4];
    __shared__ unsigned long long best_w[4];
    __shared__ uint32_t red[4];
    __shared__ float Hnew[9];
    __shared__ int fit_ok;
    __shared__ float hyp_s[HG_KPRE * 9];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;

    // all hypotheses travel with the keys, so the winner's matrix is not one more dependent trip to memory
    const bool pre = a.K <= (uint32_t)HG_KPRE;
    if (pre)
        for (uint32_t e = tid; e < 9u * a.K; e += HG_T) hyp_s[e] = a.hyp[e];
    // best hypothesis: most inliers, ties to the lowest index
    unsigned long long best = 0;
    for (uint32_t k = tid; k < a.K; k += HG_T) {
        const unsigned long long key = ((unsigned long long)a.keys[k] << 32) | (unsigned long long)(0xffffffffu - k);
        best = key > best ? key : best;
    }
    uint32_t usable = 0;
    if (NPT > 0) usable = __popc(P.us);
    else
        for (uint32_t i = tid; i < n; i += HG_T) usable += hg_usable(a.src[i], a.dst[i]) ? 1u : 0u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
        usable += __shfl_xor(usable, off);
    }
    HG_STAMP(1);
    if (lane == 0) { best_w[wv] = best; red[wv] = usable; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) best = best_w[w] > best ? best_w[w] : best;
    usable = red[0] + red[1] + red[2] + red[3];
    const uint32_t key = (uint32_t)(best >> 32), kbest = 0xffffffffu - (uint32_t)(best & 0xffffffffu);

    float H[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    uint32_t cnt = 0, inl = 0;
    if (usable >= 4 && key >= 5) {      // a valid hypothesis with at least 4 inliers
        cnt = key - 1u;
#pragma unroll
        for (int i = 0; i < 9; ++i) H[i] = pre ? hyp_s[9 * kbest + i] : a.hyp[9 * kbest + i];
        HG_STAMP(2);

        // normalisation over all usable correspondences (two passes, interleaved partials)
        double c4[4] = { 0, 0, 0, 0 };
        if (NPT > 0) {
#pragma unroll
            for (int j = 0; j < NPT; ++j)
                if ((P.us >> j) & 1u) {
                    c4[0] += (double)P.ps[j].x; c4[1] += (double)P.ps[j].y; c4[2] += (double)P.pd[j].x; c4[3] += (double)P.pd[j].y;
                }
        } else {
            for (uint32_t i = tid; i < n; i += HG_T) {
                const float2 s = a.src[i], d = a.dst[i];
                if (!hg_usable(s, d)) continue;
                c4[0] += (double)s.x; c4[1] += (double)s.y; c4[2] += (double)d.x; c4[3] += (double)d.y;
            }
        }
        block_tree_n<4>(c4, wsum);
        block_tree_read<4>(c4, wsum);
        const double cntd = (double)usable;
        HgNorm nm;
        nm.cx = c4[0] / cntd; nm.cy = c4[1] / cntd; nm.cu = c4[2] / cntd; nm.cv = c4[3] / cntd;
        HG_STAMP(3);
        double d2[2] = { 0, 0 };
        if (NPT > 0) {
#pragma unroll
            for (int j = 0; j < NPT; ++j)
                if ((P.us >> j) & 1u) {
                    d2[0] += fabs((double)P.ps[j].x - nm.cx) + fabs((double)P.ps[j].y - nm.cy);
                    d2[1] += fabs((double)P.pd[j].x - nm.cu) + fabs((double)P.pd[j].y - nm.cv);
                }
        } else {
            for (uint32_t i = tid; i < n; i += HG_T) {
                const float2 s = a.src[i], d = a.dst[i];
                if (!hg_usable(s, d)) continue;
                d2[0] += fabs((double)s.x - nm.cx) + fabs((double)s.y - nm.cy);
                d2[1] += fabs((double)d.x - nm.cu) + fabs((double)d.y - nm.cv);
            }
        }
        block_tree_n<2>(d2, wsum);
        block_tree_read<2>(d2, wsum);
        HG_STAMP(4);
        const bool norm_ok = d2[0] > 0.0 && d2[1] > 0.0;
        nm.ss = norm_ok ? cntd / d2[0] : 0.0;
        nm.sd = norm_ok ? cntd / d2[1] : 0.0;

        if (NPT > 0) {
#pragma unroll
            for (int j = 0; j < NPT; ++j) inl |= (((P.us >> j) & 1u) & (uint32_t)hg_inlier(H, P.ps[j], P.pd[j], a.thr2)) << j;
        }
        HG_STAMP(5);

        for (uint32_t r = 0; r < a.max_refine && norm_ok; ++r) {
            HG_STAMP(8 + 6 * (r < 3 ? r : 3));
            // least-squares refit over the inliers of H
            double acc[HG_NACC];
#pragma unroll
            for (int q = 0; q < HG_NACC; ++q) acc[q] = 0.0;
#define HG_ACCUMULATE(sx, sy, dx, dy)                                                                                          \
    {                                                                                                                          \
        const double x = ((double)(sx) - nm.cx) * nm.ss, y = ((double)(sy) - nm.cy) * nm.ss;                                    \
        const double u = ((double)(dx) - nm.cu) * nm.sd, v = ((double)(dy) - nm.cv) * nm.sd;                                    \
        const double q0 = -(x * u), q1 = -(y * u), r0 = -(x * v), r1 = -(y * v);                                                \
        ACC(0, x, x); ACC(1, x, y); ACC(2, x, 1.0); ACC(3, y, y); ACC(4, y, 1.0); ACC(5, 1.0, 1.0);                              \
        ACC(6, x, q0); ACC(7, x, q1); ACC(8, y, q0); ACC(9, y, q1); ACC(10, 1.0, q0); ACC(11, 1.0, q1);                          \
        ACC(12, x, r0); ACC(13, x, r1); ACC(14, y, r0); ACC(15, y, r1); ACC(16, 1.0, r0); ACC(17, 1.0, r1);                      \
        ACC(18, q0, q0); ACC(18, r0, r0); ACC(19, q0, q1); ACC(19, r0, r1); ACC(20, q1, q1); ACC(20, r1, r1);                    \
        ACC(21, x, u); ACC(22, y, u); ACC(23, 1.0, u);                                                                         \
        ACC(24, x, v); ACC(25, y, v); ACC(26, 1.0, v);                                                                         \
        ACC(27, q0, u); ACC(27, r0, v); ACC(28, q1, u); ACC(28, r1, v);                                                         \
    }
#define ACC(k, p, q) acc[k] = fma(p, q, acc[k])
            if (NPT > 0) {
#pragma unroll
                for (int j = 0; j < NPT; ++j) {
                    if (!((inl >> j) & 1u)) continue;
                    // opaque copies: keeps the compiler from hoisting every point's binary64 terms out of the refit
                    // loop (they do not fit in registers next to the 29 sums)
                    float sx = P.ps[j].x, sy = P.ps[j].y, dx = P.pd[j].x, dy = P.pd[j].y;
                    asm volatile("" : "+v"(sx), "+v"(sy), "+v"(dx), "+v"(dy));
                    HG_ACCUMULATE(sx, sy, dx, dy)
                }
            } else {
                for (uint32_t i = tid; i < n; i += HG_T) {
                    const float2 s = a.src[i], d = a.dst[i];
                    if (!(hg_usable(s, d) && hg_inlier(H, s, d, a.thr2))) continue;
                    HG_ACCUMULATE(s.x, s.y, d.x, d.y)
                }
            }
#undef ACC
#undef HG_ACCUMULATE
            HG_STAMP(9 + 6 * (r < 3 ? r : 3));
            block_tree_n<HG_NACC>(acc, wsum);
            HG_STAMP(10 + 6 * (r < 3 ? r : 3));
            if (wv == 0) {
                block_tree_read<HG_NACC>(acc, wsum);
                hg_fit_publish(acc, nm, lane, Hnew, &fit_ok);
            }
            HG_STAMP(12 + 6 * (r < 3 ? r : 3));
            __syncthreads();
            if (!fit_ok) break;     // block-uniform
            float Hn[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Hn[i] = Hnew[i];
            uint32_t is = 0, c = 0, diff = 0;
            if (NPT > 0) {
#pragma unroll
                for (int j = 0; j < NPT; ++j) is |= (((P.us >> j) & 1u) & (uint32_t)hg_inlier(Hn, P.ps[j], P.pd[j], a.thr2)) << j;
                c = __popc(is);
                diff = __popc(is ^ inl);
            } else {
                for (uint32_t i = tid; i < n; i += HG_T) {
                    const float2 s = a.src[i], d = a.dst[i];
                    const bool us = hg_usable(s, d);
                    const bool was = us && hg_inlier(H, s, d, a.thr2), now = us && hg_inlier(Hn, s, d, a.thr2);
                    c += now ? 1u : 0u;
                    diff += (was != now) ? 1u : 0u;
                }
            }
            c = block_sum_u32(c, red);
            diff = block_sum_u32(diff, red);
            HG_STAMP(13 + 6 * (r < 3 ? r : 3));
            if (c < cnt) break;
#pragma unroll
            for (int i = 0; i < 9; ++i) H[i] = Hn[i];
            cnt = c;
            inl = is;
            if (diff == 0) break;
        }
    }
    HG_STAMP(6);

    if (a.mask) {
        if (NPT > 0) {
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const uint32_t i = tid + (uint32_t)j * HG_T;
                if (i < n) a.mask[i] = (cnt != 0 && ((inl >> j) & 1u)) ? 255 : 0;
            }
        } else {
            for (uint32_t i = tid; i < n; i += HG_T) {
                const float2 s = a.src[i], d = a.dst[i];
                a.mask[i] = (cnt != 0 && hg_usable(s, d) && hg_inlier(H, s, d, a.thr2)) ? 255 : 0;
            }
        }
    }
    if (tid < 9) {   // transposed on the way out (homography_filter_node.cpp:87-91)
        const float v = H[3 * (tid % 3) + tid / 3];
        a.matrix[tid] = v;
        if (a.matrix_host) a.matrix_host[tid] = v;
    }
    if (tid == 0 && a.mask_count) {
        *a.mask_count = n;
        if (a.mask_count_host) *a.mask_count_host = n;
    }
    HG_STAMP(7);
}

// One launch: workgroup k evaluates hypothesis k; the workgroup that publishes last (a ticket counter) goes on to
// the refinement with its points still in registers.
template <int NPT>
__global__ __launch_bounds__(HG_T) void k_find_homography(HgArgs a, uint32_t bound)
{
    __shared__ float Hs[9];
    __shared__ int valid;
    __shared__ uint32_t red[4];
    __shared__ uint32_t is_last;
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    HG_STAMP(32);
    HgPoints<NPT> P;
    P.fetch(a, bound);
    const uint32_t n = hg_n(a);
    HG_STAMP(33);
    if (tid < 64) {
        float e;
        const bool ok = hg_make_hypothesis(a, k, n, tid, e);
        if (tid < 9) Hs[tid] = ok ? e : (tid % 4u == 0u ? 1.0f : 0.0f);
        if (tid == 0) valid = ok ? 1 : 0;
    }
    P.flag(n);
    HG_STAMP(34);
    __syncthreads();
    HG_STAMP(35);
    float H[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = Hs[i];
    uint32_t cnt = 0;
    if (valid) {
        if (NPT > 0) {
#pragma unroll
            for (int j = 0; j < NPT; ++j) cnt += (((P.us >> j) & 1u) & (uint32_t)hg_inlier(H, P.ps[j], P.pd[j], a.thr2));
        } else {
            for (uint32_t i = tid; i < n; i += HG_T) {
                const float2 s = a.src[i], d = a.dst[i];
                cnt += (hg_usable(s, d) && hg_inlier(H, s, d, a.thr2)) ? 1u : 0u;
            }
        }
    }
    HG_STAMP(36);
    cnt = block_sum_u32(cnt, red);
    if (tid == 0) a.keys[k] = valid ? cnt + 1u : 0u;
    if (tid < 9) a.hyp[9 * k + tid] = Hs[tid];
    HG_STAMP(37);
    // publish (release at device scope: the other workgroups may sit on other XCDs, behind other L2s), take a ticket
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == a.K - 1u) ? 1u : 0u;
        if (t == a.K - 1u) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!is_last) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);     // every thread of the last workgroup: see all published keys / hypotheses
    HG_STAMP(0);
    hg_refine<NPT>(a, P, n);
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

hipError_t launch_find_homography(const HomographyParams &p, hipStream_t stream)
{
    if (p.hypotheses == 0) return hipErrorInvalidValue;
    HgArgs a{};
    a.src = static_cast<const float2 *>(p.src); a.dst = static_cast<const float2 *>(p.dst);
    a.src_count = p.src_count; a.dst_count = p.dst_count; a.src_cap = p.src_capacity; a.dst_cap = p.dst_capacity;
    a.thr2 = p.threshold * p.threshold;
    a.K = p.hypotheses; a.max_refine = p.max_refine_iters;
    a.matrix = p.matrix; a.matrix_host = p.matrix_host;
    a.mask = p.mask; a.mask_count = p.mask_count; a.mask_count_host = p.mask_count_host; a.mask_cap = p.mask_capacity;
    a.ticket = static_cast<uint32_t *>(p.scratch);          // at a fixed place, whatever the hypothesis count
    a.keys = a.ticket + 4;
    a.hyp = reinterpret_cast<float *>(a.keys + ((p.hypotheses + 3u) & ~3u));
    // the point count lives in device memory; the capacities bound it
    uint32_t bound = p.src_capacity < p.dst_capacity ? p.src_capacity : p.dst_capacity;
    if (p.mask && p.mask_capacity < bound) bound = p.mask_capacity;
    const bool force_generic = getenv("VISUALSLAM_AMD_HG_GENERIC") != nullptr;   // tests: exercise the any-size kernel
    if (bound <= 16u * HG_T && !force_generic) hipLaunchKernelGGL(k_find_homography<16>, dim3(p.hypotheses), dim3(HG_T), 0, stream, a, bound);
    else hipLaunchKernelGGL(k_find_homography<0>, dim3(p.hypotheses), dim3(HG_T), 0, stream, a, bound);
    return hipGetLastError();
}

}  // namespace vsa

#ifdef VSA_HG_STAMPS
extern "C" int visualslam_amd_debug_hg_stamps(uint64_t *host, size_t n_u64)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vsa::g_hg_stamps), n_u64 * sizeof(uint64_t));
}
#endif
