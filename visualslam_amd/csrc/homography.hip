// homography.hip -- nvxFindHomography (RANSAC) on the PyrLK matches.
//
// Reference call sites replaced: nvxFindHomographyNode(graph, srcPoints, dstPoints, homography, RANSAC, threshold,
// maxEstimateIters, maxRefineIters, confidence, outlierRatio, mask)  (demos/video_stabilizer/stabilizer.cpp:264-273,
// VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp:1576-1578); output conventions from the consumer
// demos/video_stabilizer/homography_filter_node.cpp:36-159.  Semantics and the exact floating-point order:
// oracle/vxoracle_homography.c -- results are bit-identical.
//
// Two launches, no host round trip:
//   k_hg_hypotheses  one workgroup per hypothesis (all hypotheses of the a-priori RANSAC count are independent, so
//                    they run as one batch): thread 0 draws the 4-point sample and maps it exactly in binary64,
//                    the workgroup counts its inliers over all correspondences (binary32, ballot/popcount);
//   k_hg_refine      one workgroup: picks the best hypothesis, normalises coordinates, then alternates
//                    least-squares refits (29 binary64 sums in 256 interleaved partials + butterfly, 8x8 LDL^T on
//                    one lane) with inlier recounts; writes matrix, mask and counts (also to pinned mirrors).
// The work is tiny (56 x 4096 point tests); it is latency, not throughput, that matters here.
#include <hip/hip_runtime.h>

#include <cmath>

#include "kernels.hpp"

namespace vsa {
namespace {

constexpr int HG_T = 256;      // threads = interleaved partial sums (the oracle's HG_P)
constexpr int HG_NACC = 29;

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

__device__ bool hg_square_to_quad(const double *x, const double *y, double *A)
{
    const double dx1 = x[1] - x[2], dx2 = x[3] - x[2], dy1 = y[1] - y[2], dy2 = y[3] - y[2];
    const double sx = ((x[0] - x[1]) + x[2]) - x[3], sy = ((y[0] - y[1]) + y[2]) - y[3];
    const double den = dx1 * dy2 - dx2 * dy1;
    if (!(den != 0.0)) return false;
    const double g = (sx * dy2 - dx2 * sy) / den, h = (dx1 * sy - sx * dy1) / den;
    A[0] = (x[1] - x[0]) + g * x[1]; A[1] = (x[3] - x[0]) + h * x[3]; A[2] = x[0];
    A[3] = (y[1] - y[0]) + g * y[1]; A[4] = (y[3] - y[0]) + h * y[3]; A[5] = y[0];
    A[6] = g; A[7] = h; A[8] = 1.0;
    return true;
}

__device__ bool hg_to_f32(const double *H, float *Hf)
{
    double m = 0.0;
    for (int i = 0; i < 9; ++i) { const double v = fabs(H[i]); if (v > m) m = v; }
    if (!(fabs(H[8]) > 1e-9 * m) || !(m < INFINITY)) return false;
    for (int i = 0; i < 9; ++i) Hf[i] = (float)(H[i] / H[8]);
    return true;
}

__device__ __forceinline__ bool hg_inlier(const float *H, float2 s, float2 d, float thr2)
{
    const float X = fmaf(H[0], s.x, fmaf(H[1], s.y, H[2]));
    const float Y = fmaf(H[3], s.x, fmaf(H[4], s.y, H[5]));
    const float W = fmaf(H[6], s.x, fmaf(H[7], s.y, H[8]));
    if (!(W > 1e-6f)) return false;
    const float du = X / W - d.x, dv = Y / W - d.y;
    return fmaf(du, du, dv * dv) <= thr2;
}

// block-wide integer sum of a per-thread count (order-free)
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *lds4)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

__global__ __launch_bounds__(HG_T) void k_hg_hypotheses(HgArgs a)
{
    __shared__ float Hs[9];
    __shared__ int valid;
    __shared__ uint32_t red[4];
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = hg_n(a);
    if (tid == 0) {
        bool ok = n >= 4;
        uint32_t idx[4] = { 0, 0, 0, 0 };
        uint32_t s = (k + 1u) * 0x9E3779B9u;
        for (int j = 0; j < 4 && ok; ++j) {
            bool got = false;
            for (int t = 0; t < 64 && !got; ++t) {
                const uint32_t i = hg_draw(s, n);
                if (!hg_usable(a.src[i], a.dst[i])) continue;
                bool dup = false;
                for (int q = 0; q < j; ++q) dup |= idx[q] == i;
                if (dup) continue;
                idx[j] = i;
                got = true;
            }
            ok = got;
        }
        float Hf[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        if (ok) {
            double x[4], y[4], u[4], v[4], A[9], B[9], adj[9], H[9];
            for (int j = 0; j < 4; ++j) {
                const float2 sp = a.src[idx[j]], dp = a.dst[idx[j]];
                x[j] = sp.x; y[j] = sp.y; u[j] = dp.x; v[j] = dp.y;
            }
            ok = hg_square_to_quad(x, y, A) && hg_square_to_quad(u, v, B);
            if (ok) {
                adj[0] = A[4] * A[8] - A[5] * A[7]; adj[1] = A[2] * A[7] - A[1] * A[8]; adj[2] = A[1] * A[5] - A[2] * A[4];
                adj[3] = A[5] * A[6] - A[3] * A[8]; adj[4] = A[0] * A[8] - A[2] * A[6]; adj[5] = A[2] * A[3] - A[0] * A[5];
                adj[6] = A[3] * A[7] - A[4] * A[6]; adj[7] = A[1] * A[6] - A[0] * A[7]; adj[8] = A[0] * A[4] - A[1] * A[3];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c)
                        H[3 * r + c] = (B[3 * r] * adj[c] + B[3 * r + 1] * adj[3 + c]) + B[3 * r + 2] * adj[6 + c];
                ok = hg_to_f32(H, Hf);
            }
        }
        for (int i = 0; i < 9; ++i) Hs[i] = Hf[i];
        valid = ok ? 1 : 0;
    }
    __syncthreads();
    float H[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = Hs[i];
    uint32_t cnt = 0;
    if (valid)
        for (uint32_t i = tid; i < n; i += HG_T) {
            const float2 s = a.src[i], d = a.dst[i];
            cnt += (hg_usable(s, d) && hg_inlier(H, s, d, a.thr2)) ? 1u : 0u;
        }
    cnt = block_sum_u32(cnt, red);
    if (tid == 0) a.keys[k] = valid ? cnt + 1u : 0u;
    if (tid < 9) a.hyp[9 * k + tid] = Hs[tid];
}

// pairwise butterfly over the 256 per-thread partials (offsets 1..128): every thread ends with the total
__device__ __forceinline__ double block_tree(double v, double *lds)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off);
#pragma unroll
    for (int off = 64; off < HG_T; off <<= 1) {
        __syncthreads();
        lds[threadIdx.x] = v;
        __syncthreads();
        v = v + lds[threadIdx.x ^ off];
    }
    return v;
}

__global__ __launch_bounds__(HG_T) void k_hg_refine(HgArgs a)
{
    __shared__ double lds[HG_T];
    __shared__ double S[HG_NACC];
    __shared__ unsigned long long best_w[4];
    __shared__ uint32_t red[4];
    __shared__ float Hnew[9];
    __shared__ int fit_ok;
    const uint32_t tid = threadIdx.x;
    const uint32_t n = hg_n(a);

    // best hypothesis: most inliers, ties to the lowest index
    unsigned long long best = 0;
    for (uint32_t k = tid; k < a.K; k += HG_T) {
        const unsigned long long key = ((unsigned long long)a.keys[k] << 32) | (unsigned long long)(0xffffffffu - k);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if ((tid & 63) == 0) best_w[tid >> 6] = best;
    uint32_t usable = 0;
    for (uint32_t i = tid; i < n; i += HG_T) usable += hg_usable(a.src[i], a.dst[i]) ? 1u : 0u;
    usable = block_sum_u32(usable, red);      // (its barriers also publish best_w)
    for (int w = 0; w < 4; ++w) best = best_w[w] > best ? best_w[w] : best;
    const uint32_t key = (uint32_t)(best >> 32), kbest = 0xffffffffu - (uint32_t)(best & 0xffffffffu);

    float H[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    uint32_t cnt = 0;
    if (usable >= 4 && key >= 5) {   // a valid hypothesis with at least 4 inliers
        cnt = key - 1u;
#pragma unroll
        for (int i = 0; i < 9; ++i) H[i] = a.hyp[9 * kbest + i];

        // normalisation over all usable correspondences (two passes, interleaved partials)
        double sx = 0, sy = 0, su = 0, sv = 0;
        for (uint32_t i = tid; i < n; i += HG_T) {
            const float2 s = a.src[i], d = a.dst[i];
            if (!hg_usable(s, d)) continue;
            sx += (double)s.x; sy += (double)s.y; su += (double)d.x; sv += (double)d.y;
        }
        const double cntd = (double)usable;
        const double cx = block_tree(sx, lds) / cntd, cy = block_tree(sy, lds) / cntd;
        const double cu = block_tree(su, lds) / cntd, cv = block_tree(sv, lds) / cntd;
        double ds = 0, dd = 0;
        for (uint32_t i = tid; i < n; i += HG_T) {
            const float2 s = a.src[i], d = a.dst[i];
            if (!hg_usable(s, d)) continue;
            ds += fabs((double)s.x - cx) + fabs((double)s.y - cy);
            dd += fabs((double)d.x - cu) + fabs((double)d.y - cv);
        }
        ds = block_tree(ds, lds);
        dd = block_tree(dd, lds);
        const bool norm_ok = ds > 0.0 && dd > 0.0;
        const double ss = norm_ok ? cntd / ds : 0.0, sd = norm_ok ? cntd / dd : 0.0;

        for (uint32_t r = 0; r < a.max_refine && norm_ok; ++r) {
            // least-squares refit over the inliers of H
            double acc[HG_NACC];
#pragma unroll
            for (int q = 0; q < HG_NACC; ++q) acc[q] = 0.0;
            for (uint32_t i = tid; i < n; i += HG_T) {
                const float2 s = a.src[i], d = a.dst[i];
                if (!(hg_usable(s, d) && hg_inlier(H, s, d, a.thr2))) continue;
                const double x = ((double)s.x - cx) * ss, y = ((double)s.y - cy) * ss;
                const double u = ((double)d.x - cu) * sd, v = ((double)d.y - cv) * sd;
                const double q0 = -(x * u), q1 = -(y * u), r0 = -(x * v), r1 = -(y * v);
#define ACC(k, p, q) acc[k] = fma(p, q, acc[k])
                ACC(0, x, x); ACC(1, x, y); ACC(2, x, 1.0); ACC(3, y, y); ACC(4, y, 1.0); ACC(5, 1.0, 1.0);
                ACC(6, x, q0); ACC(7, x, q1); ACC(8, y, q0); ACC(9, y, q1); ACC(10, 1.0, q0); ACC(11, 1.0, q1);
                ACC(12, x, r0); ACC(13, x, r1); ACC(14, y, r0); ACC(15, y, r1); ACC(16, 1.0, r0); ACC(17, 1.0, r1);
                ACC(18, q0, q0); ACC(18, r0, r0); ACC(19, q0, q1); ACC(19, r0, r1); ACC(20, q1, q1); ACC(20, r1, r1);
                ACC(21, x, u); ACC(22, y, u); ACC(23, 1.0, u);
                ACC(24, x, v); ACC(25, y, v); ACC(26, 1.0, v);
                ACC(27, q0, u); ACC(27, r0, v); ACC(28, q1, u); ACC(28, r1, v);
#undef ACC
            }
#pragma unroll
            for (int q = 0; q < HG_NACC; ++q) {
                const double t = block_tree(acc[q], lds);
                if (tid == 0) S[q] = t;
            }
            if (tid == 0) {
                double M[8][8], b[8], h[8], Lm[8][8], D[8];
                for (int i = 0; i < 8; ++i)
                    for (int j = 0; j < 8; ++j) { M[i][j] = 0.0; Lm[i][j] = 0.0; }
                const double pp[3][3] = { { S[0], S[1], S[2] }, { S[1], S[3], S[4] }, { S[2], S[4], S[5] } };
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) { M[i][j] = pp[i][j]; M[3 + i][3 + j] = pp[i][j]; }
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 2; ++j) {
                        M[i][6 + j] = M[6 + j][i] = S[6 + 2 * i + j];
                        M[3 + i][6 + j] = M[6 + j][3 + i] = S[12 + 2 * i + j];
                    }
                M[6][6] = S[18]; M[6][7] = M[7][6] = S[19]; M[7][7] = S[20];
                b[0] = S[21]; b[1] = S[22]; b[2] = S[23]; b[3] = S[24]; b[4] = S[25]; b[5] = S[26]; b[6] = S[27]; b[7] = S[28];
                bool pd = true;
                for (int j = 0; j < 8 && pd; ++j) {
                    double d = M[j][j];
                    for (int k = 0; k < j; ++k) d -= (Lm[j][k] * Lm[j][k]) * D[k];
                    if (!(d > 1e-12)) { pd = false; break; }
                    D[j] = d;
                    for (int i = j + 1; i < 8; ++i) {
                        double s = M[i][j];
                        for (int k = 0; k < j; ++k) s -= (Lm[i][k] * Lm[j][k]) * D[k];
                        Lm[i][j] = s / d;
                    }
                }
                bool ok = false;
                float Hf[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
                if (pd) {
                    double z[8];
                    for (int i = 0; i < 8; ++i) {
                        double s = b[i];
                        for (int k = 0; k < i; ++k) s -= Lm[i][k] * z[k];
                        z[i] = s;
                    }
                    for (int i = 7; i >= 0; --i) {
                        double s = z[i] / D[i];
                        for (int k = i + 1; k < 8; ++k) s -= Lm[k][i] * h[k];
                        h[i] = s;
                    }
                    const double Hn[9] = { h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 1.0 };
                    double G[9], Hd[9];
                    for (int rr = 0; rr < 3; ++rr) {
                        G[3 * rr] = Hn[3 * rr] * ss;
                        G[3 * rr + 1] = Hn[3 * rr + 1] * ss;
                        G[3 * rr + 2] = (Hn[3 * rr + 2] - G[3 * rr] * cx) - G[3 * rr + 1] * cy;
                    }
                    for (int c = 0; c < 3; ++c) {
                        Hd[c] = G[c] / sd + cu * G[6 + c];
                        Hd[3 + c] = G[3 + c] / sd + cv * G[6 + c];
                        Hd[6 + c] = G[6 + c];
                    }
                    ok = hg_to_f32(Hd, Hf);
                }
                for (int i = 0; i < 9; ++i) Hnew[i] = Hf[i];
                fit_ok = ok ? 1 : 0;
            }
            __syncthreads();
            if (!fit_ok) break;     // block-uniform
            float Hn[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Hn[i] = Hnew[i];
            uint32_t c = 0, diff = 0;
            for (uint32_t i = tid; i < n; i += HG_T) {
                const float2 s = a.src[i], d = a.dst[i];
                const bool us = hg_usable(s, d);
                const bool was = us && hg_inlier(H, s, d, a.thr2), is = us && hg_inlier(Hn, s, d, a.thr2);
                c += is ? 1u : 0u;
                diff += (was != is) ? 1u : 0u;
            }
            c = block_sum_u32(c, red);
            diff = block_sum_u32(diff, red);
            if (c < cnt) break;
#pragma unroll
            for (int i = 0; i < 9; ++i) H[i] = Hn[i];
            cnt = c;
            if (diff == 0) break;
        }
    }

    if (a.mask)
        for (uint32_t i = tid; i < n; i += HG_T) {
            const float2 s = a.src[i], d = a.dst[i];
            a.mask[i] = (cnt != 0 && hg_usable(s, d) && hg_inlier(H, s, d, a.thr2)) ? 255 : 0;
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
    return (size_t)hypotheses * (sizeof(uint32_t) + 9 * sizeof(float)) + 16;
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
    a.keys = static_cast<uint32_t *>(p.scratch);
    a.hyp = reinterpret_cast<float *>(a.keys + ((p.hypotheses + 3u) & ~3u));
    hipLaunchKernelGGL(k_hg_hypotheses, dim3(p.hypotheses), dim3(HG_T), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_hg_refine, dim3(1), dim3(HG_T), 0, stream, a);
    return hipGetLastError();
}

}  // namespace vsa
