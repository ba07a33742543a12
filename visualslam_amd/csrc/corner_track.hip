// corner_track.hip -- nvxHarrisTrack / nvxFastTrack: corner score, 3x3 non-maximum suppression,
// strongest corner per detector cell, merge with the tracked points, deterministic output order.
//
// Reference call sites replaced: nvxHarrisTrackNode / nvxcuHarrisTrack and nvxFastTrackNode /
// nvxcuFastTrack (demos/feature_tracker/feature_tracker.cpp:418-430,
// demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:307-317).  Semantics: oracle/vxoracle.c
// (cell selection/merge, Harris response and FAST strength functions).
//
// Two launches, no atomics on global memory, no host round trip:
//   k_corner_cells<DET>  one workgroup OWNS a rectangle of whole detector cells; it walks that
//                        rectangle in LDS-staged tiles (image read once + halo), computes the
//                        score, the NMS and each cell's best (score, lowest raster index) as a
//                        64-bit key, and stores one key per cell (0 = no corner).
//   k_merge_compact      one workgroup marks cells that hold a surviving tracked point, then
//                        compacts survivors (input order) and new corners (cell raster order)
//                        with ballot/popcount prefix sums.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace vsa {
namespace {

constexpr int TW = 64, TH = 40;         // tile interior (pixels)
constexpr int HALO = 4;                 // Harris needs 3, FAST (radius 3 + NMS) needs 4
constexpr int PXW = TW + 2 * HALO, PXH = TH + 2 * HALO;
constexpr int GW = TW + 4, GH = TH + 4; // gradient grid  (interior + 2)
constexpr int SW = TW + 2, SH = TH + 2; // score grid     (interior + 1)
constexpr uint64_t OCCUPIED = ~0ull;

struct CornerArgs {
    const uint8_t *img;
    uint32_t pitch;
    int w, h;
    const uint8_t *mask;
    uint32_t mask_pitch;
    int cs, ncx, ncy, nbx, nby;
    float k, thr;
    int fast_n, fast_t;
    uint64_t *cell_best;
};

__device__ __forceinline__ uint32_t orderable(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ __forceinline__ float harris_r(int a, int b, int c, float k)
{
    const float S = 1.0f / 144.0f;
    const float A = (float)a * S, B = (float)b * S, C = (float)c * S;
    const float det = A * C - B * B;
    const float tr = A + C;
    return det - k * tr * tr;
}

// FAST-N: -inf unless the pixel is a corner at threshold t; then the exact strength (largest
// threshold for which it still is one).  Pixels below t cannot beat a candidate in the NMS, so
// their exact strength is never needed.
__device__ __forceinline__ float fast_score(const uint8_t *c /* centre in LDS */, int n, int t)
{
    const int ip = c[0];
    int d[16];
    d[0]  = c[-3 * PXW + 0] - ip;  d[1]  = c[-3 * PXW + 1] - ip;
    d[2]  = c[-2 * PXW + 2] - ip;  d[3]  = c[-1 * PXW + 3] - ip;
    d[4]  = c[3] - ip;             d[5]  = c[1 * PXW + 3] - ip;
    d[6]  = c[2 * PXW + 2] - ip;   d[7]  = c[3 * PXW + 1] - ip;
    d[8]  = c[3 * PXW + 0] - ip;   d[9]  = c[3 * PXW - 1] - ip;
    d[10] = c[2 * PXW - 2] - ip;   d[11] = c[1 * PXW - 3] - ip;
    d[12] = c[-3] - ip;            d[13] = c[-1 * PXW - 3] - ip;
    d[14] = c[-2 * PXW - 2] - ip;  d[15] = c[-3 * PXW - 1] - ip;
    uint32_t mb = 0, md = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        mb |= (d[i] > t ? 1u : 0u) << i;
        md |= (d[i] < -t ? 1u : 0u) << i;
    }
    mb |= mb << 16;
    md |= md << 16;
    uint32_t rb = mb, rd = md;
    for (int i = 1; i < n; ++i) {
        rb &= mb >> i;
        rd &= md >> i;
    }
    if (((rb | rd) & 0xffffu) == 0) return -INFINITY;
    int best = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        int lo = 255, hi = 255;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (i < n) {
                const int v = d[(s + i) & 15];
                lo = min(lo, v);
                hi = min(hi, -v);
            }
        }
        best = max(best, max(lo, hi));
    }
    return (float)(best - 1);
}

template <int DET>
__global__ __launch_bounds__(256) void k_corner_cells(CornerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // carve: keys | score | grad (Harris only) | pixels
    const int ncl = a.nbx * a.nby;
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    float *sc = reinterpret_cast<float *>(smem + (size_t)((ncl * 8 + 15) & ~15));
    uint32_t *grad = reinterpret_cast<uint32_t *>(sc + SW * SH);
    uint8_t *px = reinterpret_cast<uint8_t *>(DET == kHarris ? (void *)(grad + GW * GH) : (void *)grad);

    const int tid = threadIdx.x;
    // each XCD owns a contiguous band of cell blocks (halo rows/columns are shared inside one L2)
    const uint32_t item = xcd_contiguous_item(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bx = (int)(item % gridDim.x), by = (int)(item / gridDim.x);
    const int rx0 = bx * a.nbx * a.cs, ry0 = by * a.nby * a.cs;
    const int rw = min(a.nbx * a.cs, a.w - rx0), rh = min(a.nby * a.cs, a.h - ry0);

    for (int i = tid; i < ncl; i += 256) keys[i] = 0;

    for (int ty = 0; ty < rh; ty += TH) {
        for (int tx = 0; tx < rw; tx += TW) {
            const int tw = min(TW, rw - tx), th = min(TH, rh - ty);
            const int gx0 = rx0 + tx, gy0 = ry0 + ty;
            __syncthreads();  // previous tile fully consumed (also orders the key init)

            // S1: pixels, replicate-clamped (scores that would use clamped pixels are masked off below)
            for (int i = tid; i < PXW * PXH; i += 256) {
                const int r = i / PXW, c = i % PXW;
                if (r < th + 2 * HALO && c < tw + 2 * HALO) {
                    const int gy = min(max(gy0 - HALO + r, 0), a.h - 1);
                    const int gx = min(max(gx0 - HALO + c, 0), a.w - 1);
                    px[i] = a.img[(size_t)gy * a.pitch + gx];
                }
            }
            __syncthreads();

            if (DET == kHarris) {
                // S2: Sobel gradients on the (tw+4)x(th+4) grid, packed (ix | iy << 16)
                for (int i = tid; i < GW * GH; i += 256) {
                    const int r = i / GW, c = i % GW;
                    if (r < th + 4 && c < tw + 4) {
                        const uint8_t *p = px + (r + HALO - 2) * PXW + (c + HALO - 2);
                        const int p00 = p[-PXW - 1], p01 = p[-PXW], p02 = p[-PXW + 1];
                        const int p10 = p[-1], p12 = p[1];
                        const int p20 = p[PXW - 1], p21 = p[PXW], p22 = p[PXW + 1];
                        const int ix = (p02 + 2 * p12 + p22) - (p00 + 2 * p10 + p20);
                        const int iy = (p20 + 2 * p21 + p22) - (p00 + 2 * p01 + p02);
                        grad[i] = ((uint32_t)ix & 0xffffu) | ((uint32_t)iy << 16);
                    }
                }
                __syncthreads();
            }

            // S3: score on the (tw+2)x(th+2) grid
            for (int i = tid; i < SW * SH; i += 256) {
                const int r = i / SW, c = i % SW;
                if (r < th + 2 && c < tw + 2) {
                    const int gx = gx0 - 1 + c, gy = gy0 - 1 + r;
                    float s = -INFINITY;
                    if (DET == kHarris) {
                        if (gx >= 2 && gy >= 2 && gx < a.w - 2 && gy < a.h - 2) {
                            int sa = 0, sb = 0, sc2 = 0;
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx) {
                                    const uint32_t g = grad[(r + dy) * GW + (c + dx)];
                                    const int ix = (int)(short)(g & 0xffffu), iy = (int)g >> 16;
                                    sa += ix * ix;
                                    sb += ix * iy;
                                    sc2 += iy * iy;
                                }
                            s = harris_r(sa, sb, sc2, a.k);
                        }
                    } else {
                        if (gx >= 3 && gy >= 3 && gx < a.w - 3 && gy < a.h - 3)
                            s = fast_score(px + (r + HALO - 1) * PXW + (c + HALO - 1), a.fast_n, a.fast_t);
                    }
                    sc[i] = s;
                }
            }
            __syncthreads();

            // S4: threshold, 3x3 NMS with raster tie-break, mask, per-cell maximum
            for (int i = tid; i < TW * TH; i += 256) {
                const int ly = i / TW, lx = i % TW;
                bool cand = false;
                float s = 0.0f;
                if (ly < th && lx < tw) {
                    const float *q = sc + (ly + 1) * SW + (lx + 1);
                    s = q[0];
                    cand = (s >= a.thr) && (s != -INFINITY);
                    if (cand) {
                        cand = !(q[-SW - 1] >= s) && !(q[-SW] >= s) && !(q[-SW + 1] >= s) && !(q[-1] >= s) &&
                               !(q[1] > s) && !(q[SW - 1] > s) && !(q[SW] > s) && !(q[SW + 1] > s);
                    }
                }
                if (__ballot(cand) == 0ull) continue;  // wave-uniform early out: candidates are sparse
                if (cand) {
                    const int gx = gx0 + lx, gy = gy0 + ly;
                    if (a.mask && a.mask[(size_t)gy * a.mask_pitch + gx] == 0) continue;
                    const int cell = ((ty + ly) / a.cs) * a.nbx + (tx + lx) / a.cs;
                    const uint32_t idx = (uint32_t)gy * (uint32_t)a.w + (uint32_t)gx;
                    const uint64_t key = ((uint64_t)orderable(s) << 32) | (uint64_t)(~idx);
                    atomicMax(reinterpret_cast<unsigned long long *>(&keys[cell]), (unsigned long long)key);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < ncl; i += 256) {
        const int cx = bx * a.nbx + i % a.nbx, cy = by * a.nby + i / a.nbx;
        if (cx < a.ncx && cy < a.ncy) a.cell_best[(size_t)cy * a.ncx + cx] = keys[i];
    }
}

// ---- merge + compaction ---------------------------------------------------------------------
struct MergeArgs {
    int item;                       // ItemKind
    const void *tracked;
    const uint32_t *tracked_count;
    uint32_t tracked_capacity;
    void *out;
    uint32_t *out_count;
    uint32_t out_capacity;
    size_t *num_corners;
    uint64_t *cell_best;
    int w, h, cs, ncx, ncy;
};

struct KeypointF {
    float x, y, strength, scale, orientation;
    int32_t tracking_status;
    float error;
};

__device__ __forceinline__ uint32_t block_scan_flag(bool flag, uint32_t *wave_tot, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const uint32_t pre = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(b);
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t t = wave_tot[i];
        off += (i < wave) ? t : 0u;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return off + pre;
}

__device__ __forceinline__ bool tracked_alive(const MergeArgs &a, uint32_t i, float &x, float &y)
{
    if (a.item == kPoint2f) {
        const float2 p = reinterpret_cast<const float2 *>(a.tracked)[i];
        x = p.x; y = p.y;
    } else {
        const KeypointF *p = reinterpret_cast<const KeypointF *>(a.tracked) + i;
        x = p->x; y = p->y;
        if (p->tracking_status == 0) return false;
    }
    return x >= 0.0f && y >= 0.0f && x <= (float)(a.w - 1) && y <= (float)(a.h - 1);
}

__global__ __launch_bounds__(1024) void k_merge_compact(MergeArgs a)
{
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t nt = a.tracked ? min(*a.tracked_count, a.tracked_capacity) : 0u;
    const uint32_t ncells = (uint32_t)a.ncx * (uint32_t)a.ncy;

    // phase 0: cells holding a surviving tracked point emit no new corner
    for (uint32_t i = tid; i < nt; i += 1024u) {
        float x, y;
        if (tracked_alive(a, i, x, y)) {
            const uint32_t cx = min((uint32_t)x / (uint32_t)a.cs, (uint32_t)a.ncx - 1u);
            const uint32_t cy = min((uint32_t)y / (uint32_t)a.cs, (uint32_t)a.ncy - 1u);
            a.cell_best[(size_t)cy * a.ncx + cx] = OCCUPIED;
        }
    }
    __threadfence_block();
    __syncthreads();

    uint32_t base = 0;
    // phase 1: survivors, input order
    for (uint32_t i0 = 0; i0 < nt; i0 += 1024u) {
        const uint32_t i = i0 + tid;
        float x = 0, y = 0;
        const bool alive = (i < nt) && tracked_alive(a, i, x, y);
        uint32_t total;
        const uint32_t pos = base + block_scan_flag(alive, wave_tot, total);
        if (alive && pos < a.out_capacity) {
            if (a.item == kPoint2f) {
                reinterpret_cast<float2 *>(a.out)[pos] = make_float2(x, y);
            } else {
                reinterpret_cast<KeypointF *>(a.out)[pos] = reinterpret_cast<const KeypointF *>(a.tracked)[i];
            }
        }
        base += total;
    }
    // phase 2: new corners, cell raster order
    for (uint32_t c0 = 0; c0 < ncells; c0 += 1024u) {
        const uint32_t c = c0 + tid;
        const uint64_t key = (c < ncells) ? a.cell_best[c] : 0ull;
        const bool has = key != 0ull && key != OCCUPIED;
        uint32_t total;
        const uint32_t pos = base + block_scan_flag(has, wave_tot, total);
        if (has && pos < a.out_capacity) {
            const uint32_t idx = ~(uint32_t)(key & 0xffffffffull);
            const float fx = (float)(idx % (uint32_t)a.w), fy = (float)(idx / (uint32_t)a.w);
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
        base += total;
    }
    if (tid == 0) {
        *a.out_count = min(base, a.out_capacity);
        if (a.num_corners) *a.num_corners = (size_t)base;
    }
}

inline void cell_block(uint32_t cs, int &nbx, int &nby)
{
    nbx = (int)(TW / cs);
    nby = (int)(TH / cs);
    if (nbx < 1) nbx = 1;
    if (nby < 1) nby = 1;
}

}  // namespace

size_t corner_track_scratch_bytes(uint32_t width, uint32_t height, uint32_t cell_size)
{
    if (cell_size == 0) return 0;
    const size_t ncx = (width + cell_size - 1) / cell_size, ncy = (height + cell_size - 1) / cell_size;
    return ncx * ncy * sizeof(uint64_t);
}

hipError_t launch_corner_track(const CornerTrackParams &p, hipStream_t stream)
{
    const uint32_t cs = p.cell_size;
    CornerArgs a{};
    a.img = p.image.ptr; a.pitch = p.image.pitch; a.w = (int)p.image.width; a.h = (int)p.image.height;
    a.mask = p.mask; a.mask_pitch = p.mask_pitch;
    a.cs = (int)cs;
    a.ncx = (int)((p.image.width + cs - 1) / cs);
    a.ncy = (int)((p.image.height + cs - 1) / cs);
    cell_block(cs, a.nbx, a.nby);
    a.k = p.harris_k; a.thr = p.threshold;
    a.fast_n = (int)p.fast_n; a.fast_t = (int)p.threshold;
    a.cell_best = reinterpret_cast<uint64_t *>(p.scratch);

    const dim3 grid((a.ncx + a.nbx - 1) / a.nbx, (a.ncy + a.nby - 1) / a.nby);
    const size_t keys_b = ((size_t)a.nbx * a.nby * 8 + 15) & ~(size_t)15;
    if (p.detector == kHarris) {
        const size_t lds = keys_b + SW * SH * 4 + GW * GH * 4 + PXW * PXH;
        hipLaunchKernelGGL(k_corner_cells<kHarris>, grid, dim3(256), lds, stream, a);
    } else {
        const size_t lds = keys_b + SW * SH * 4 + PXW * PXH;
        hipLaunchKernelGGL(k_corner_cells<kFast>, grid, dim3(256), lds, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    MergeArgs m{};
    m.item = p.item; m.tracked = p.tracked; m.tracked_count = p.tracked_count;
    m.tracked_capacity = p.tracked_capacity;
    m.out = p.out; m.out_count = p.out_count; m.out_capacity = p.out_capacity;
    m.num_corners = p.num_corners; m.cell_best = a.cell_best;
    m.w = a.w; m.h = a.h; m.cs = a.cs; m.ncx = a.ncx; m.ncy = a.ncy;
    hipLaunchKernelGGL(k_merge_compact, dim3(1), dim3(1024), 0, stream, m);
    return hipGetLastError();
}

}  // namespace vsa
