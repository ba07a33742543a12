// corner_track.hip -- nvxHarrisTrack / nvxFastTrack: corner score, 3x3 non-maximum suppression,
// strongest corner per detector cell, merge with the tracked points, deterministic output order.
//
// Reference call sites replaced: nvxHarrisTrackNode / nvxcuHarrisTrack and nvxFastTrackNode /
// nvxcuFastTrack (demos/feature_tracker/feature_tracker.cpp:418-430,
// demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:307-317).  Semantics: oracle/vxoracle.c
// (cell selection/merge, Harris response and FAST strength functions).
//
// Two launches, no atomics on global memory, no host round trip:
//   k_harris_cells /     one workgroup OWNS a rectangle of whole detector cells (image read once +
//   k_fast_cells         halo), computes the score, the NMS and each cell's best (score, lowest
//                        raster index) as a 64-bit key, and stores one key per cell (0 = no corner).
//                        Harris: one wave per rectangle, lane = image column, rows streamed through
//                        register sliding windows, horizontal neighbours by DPP wave shifts.
//                        FAST: 256-thread workgroup over LDS-staged tiles.
//   k_merge_compact      one workgroup marks cells that hold a surviving tracked point, then
//                        compacts survivors (input order) and new corners (cell raster order)
//                        with ballot/popcount prefix sums.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace vsa {
namespace {

constexpr int TW = 64, TH = 40;         // FAST tile interior (pixels)
constexpr int HALO = 4;                 // FAST: radius 3 + NMS
constexpr int PXW = TW + 2 * HALO, PXH = TH + 2 * HALO;
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

// FAST-N strength: the largest threshold for which the pixel still passes the segment test, i.e.
// max over the 16 arcs of N consecutive circle pixels of min(d) (brighter) or min(-d) (darker), minus one; -inf when
// no arc is uniformly brighter or darker (strength < 0).  Branch-free: arcs are built from 3-pixel minima/maxima
// (v_min3 / v_max3), every lane does the same ~90 integer operations, so a wave's cost does not depend on whether
// one of its 64 pixels happens to be a corner.
template <int N>
__device__ __forceinline__ float fast_score(const uint8_t *c /* centre in LDS */)
{
    static_assert(N >= 9 && N <= 12, "FAST-9 .. FAST-12");
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
    int lo3[16], hi3[16];          // min / max of circle pixels i, i+1, i+2
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        lo3[i] = min(min(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
        hi3[i] = max(max(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
    }
    int bright = -256, dark = 256;  // max over arcs of min(d); min over arcs of max(d)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int lo = min(min(lo3[i], lo3[(i + 3) & 15]), lo3[(i + 6) & 15]);     // 9 consecutive pixels from i
        int hi = max(max(hi3[i], hi3[(i + 3) & 15]), hi3[(i + 6) & 15]);
        if (N == 10) { lo = min(lo, d[(i + 9) & 15]); hi = max(hi, d[(i + 9) & 15]); }
        if (N == 11) { lo = min(min(lo, d[(i + 9) & 15]), d[(i + 10) & 15]); hi = max(max(hi, d[(i + 9) & 15]), d[(i + 10) & 15]); }
        if (N == 12) { lo = min(lo, lo3[(i + 9) & 15]); hi = max(hi, hi3[(i + 9) & 15]); }
        bright = max(bright, lo);
        dark = min(dark, hi);
    }
    const int best = max(max(bright, -dark), 0);
    return best > 0 ? (float)(best - 1) : -INFINITY;
}

// ---- Harris: one wave per rectangle of cells, lane = column ----------------------------------
constexpr int HS_USE = 55;    // owned columns per strip: 64 lanes - 3 right halo - (3..6) left halo (dword-aligned origin)
constexpr int HS_ROWS = 48;   // owned rows per chunk (LDS tile height - 6)
constexpr int HS_BAND = 24;   // preferred owned rows per wave: shorter bands = more waves per SIMD
constexpr int HS_R = HS_ROWS + 6 + 2;   // LDS tile rows: owned + halo + the partial last triple
constexpr int HS_NLD = (HS_R * 16 + 63) / 64;

// Wave shifts are only ever issued as v_mov_b32_dpp: folded into an arithmetic instruction (what the
// compiler's DPP combiner does with `right - left`) a wave_shr source read as zero on gfx950, so the
// moved value is pinned in a register before it is used.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    int r = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}
// value of the lane holding column x-1 / x+1 (0 beyond the wave; those lanes are halo)
__device__ __forceinline__ int left_i(int v) { return dpp_i<0x138>(v); }    // wave_shr:1
__device__ __forceinline__ int right_i(int v) { return dpp_i<0x130>(v); }   // wave_shl:1
__device__ __forceinline__ float left_f(float v) { return __int_as_float(left_i(__float_as_int(v))); }
__device__ __forceinline__ float right_f(float v) { return __int_as_float(right_i(__float_as_int(v))); }


struct HarrisRows {
    int hd[3], hs[3];             // per pixel row: right-left and left+2*centre+right   (ring, index = row % 3)
    int hxx[3], hxy[3], hyy[3];   // per gradient row: horizontal 3-sums of the gradient products (ring)
};

// Per-tile constants of the streamed pass
struct HarrisTile {
    const uint8_t *col;        // LDS: this lane's column of the pixel tile (row stride 64)
    int gx;                    // image column of this lane
    int gy_tile0;              // image row of tile row 0
    int th;                    // owned rows
    bool col_valid, col_owned;
    int cell_x, cell_row0;     // cell column of this lane, owned-row offset inside the rectangle (ty)
    uint64_t *keys;
};

// One streamed tile row.  PH = r % 3 (compile time, so the rings live in registers without moves);
// STAGES = how much of the pipeline is live (warm-up rows run fewer stages).  Pixel row r completes
// gradient row r-1 and returns the score of row r-2.  All integer stages are exact, so the separable
// evaluation order is free; the response is harris_r() as in the oracle.
template <int PH, int STAGES>
__device__ __forceinline__ float harris_row(HarrisRows &st, const CornerArgs &a, const HarrisTile &t, int r)
{
    constexpr int P0 = (PH + 1) % 3, P1 = (PH + 2) % 3, P2 = PH;   // ring slots of rows r-2, r-1, r
    {
        const int p = t.col[r * 64];
        const int l = left_i(p), rt = right_i(p);
        st.hd[P2] = rt - l;
        st.hs[P2] = l + 2 * p + rt;
    }
    if (STAGES >= 2) {   // Sobel at gradient row r-1 -> ring slot P1; |ix|,|iy| <= 1020: 24-bit multiplies
        const int ix = st.hd[P0] + 2 * st.hd[P1] + st.hd[P2];
        const int iy = st.hs[P2] - st.hs[P0];
        const int xx = __mul24(ix, ix), xy = __mul24(ix, iy), yy = __mul24(iy, iy);
        st.hxx[P1] = left_i(xx) + xx + right_i(xx);
        st.hxy[P1] = left_i(xy) + xy + right_i(xy);
        st.hyy[P1] = left_i(yy) + yy + right_i(yy);
    }
    float s = -INFINITY;
    if (STAGES >= 3) {   // score of row r-2 (gradient rows r-3, r-2, r-1 = all three ring slots)
        const int gy = t.gy_tile0 + r - 2;
        const float v = harris_r(st.hxx[0] + st.hxx[1] + st.hxx[2], st.hxy[0] + st.hxy[1] + st.hxy[2],
                                 st.hyy[0] + st.hyy[1] + st.hyy[2], a.k);
        s = (t.col_valid && gy >= 2 && gy < a.h - 2) ? v : -INFINITY;
    }
    return s;
}

// is the centre of (up, c, dn) a corner candidate that survives the 3x3 NMS?  `own` already holds
// ownership and threshold.  Every lane takes part in the shifts (no divergence around them).
__device__ __forceinline__ bool harris_nms(bool own, float up, float c, float dn)
{
    const float ul = left_f(up), ur = right_f(up), cl = left_f(c), cr = right_f(c), dl = left_f(dn), dr = right_f(dn);
    return own && !(ul >= c) && !(up >= c) && !(ur >= c) && !(cl >= c) && !(cr > c) && !(dl > c) && !(dn > c) &&
           !(dr > c);
}

__device__ __forceinline__ void harris_emit(const CornerArgs &a, const HarrisTile &t, int n /* owned row */, float c)
{
    const int gy = t.gy_tile0 + 3 + n;
    if (a.mask && a.mask[(size_t)gy * a.mask_pitch + t.gx] == 0) return;
    const int cell = ((t.cell_row0 + n) / a.cs) * a.nbx + t.cell_x;
    const uint32_t idx = ((uint32_t)gy << 16) | (uint32_t)t.gx;     // raster order for images up to 65535 x 65535, decoded by shifts
    const uint64_t key = ((uint64_t)orderable(c) << 32) | (uint64_t)(~idx);
    atomicMax(reinterpret_cast<unsigned long long *>(&t.keys[cell]), (unsigned long long)key);
}

__global__ __launch_bounds__(64) void k_harris_cells(CornerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    VSA_SHORT_KERNEL_PRIORITY();
    // carve: pixel tile (HS_R rows x 64 columns) | keys
    uint8_t *tile = smem;
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem + HS_R * 64);
    const int ncl = a.nbx * a.nby;
    const int lane = threadIdx.x;

    const uint32_t gxn = (uint32_t)(a.ncx + a.nbx - 1) / (uint32_t)a.nbx;
    // each XCD owns a contiguous band of cell blocks (halo rows/columns are shared inside one L2)
    const uint32_t item = xcd_contiguous_item(blockIdx.x, gridDim.x);
    const int bx = (int)(item % gxn), by = (int)(item / gxn);
    const int rx0 = bx * a.nbx * a.cs, ry0 = by * a.nby * a.cs;
    const int rw = min(a.nbx * a.cs, a.w - rx0), rh = min(a.nby * a.cs, a.h - ry0);
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.img) | a.pitch) & 3u) == 0;

    for (int i = lane; i < ncl; i += 64) keys[i] = 0;

    for (int ty = 0; ty < rh; ty += HS_ROWS) {
        for (int tx = 0; tx < rw; tx += HS_USE) {
            const int tw = min(HS_USE, rw - tx), th = min(HS_ROWS, rh - ty);
            const int gx0 = rx0 + tx, gy0 = ry0 + ty;
            const int xa = (gx0 - 3) & ~3;          // column of lane 0 (multiple of 4, may be negative)
            const int R = th + 6;                   // tile rows: image rows gy0-3 .. gy0+th+2
            __syncthreads();                        // previous tile fully consumed (orders the key init too)

            // stage the tile: rows replicate-clamped; columns outside the image hold arbitrary in-bounds
            // bytes (every score whose support touches them is invalid by definition)
            if (aligned) {
                const int xmax = (int)a.pitch - 4;
                uint32_t v[HS_NLD];
#pragma unroll
                for (int i = 0; i < HS_NLD; ++i) {
                    const int idx = lane + 64 * i, r = idx >> 4, d = idx & 15;   // load i covers tile rows 4i .. 4i+3
                    const int gy = min(max(gy0 - 3 + r, 0), a.h - 1);
                    const int X = min(max(xa + 4 * d, 0), xmax);
                    v[i] = 0;
                    if (4 * i < R + 2)   // wave-uniform
                        v[i] = *reinterpret_cast<const uint32_t *>(a.img + (uint32_t)gy * a.pitch + (uint32_t)X);
                }
#pragma unroll
                for (int i = 0; i < HS_NLD; ++i)
                    if (4 * i < R + 2) reinterpret_cast<uint32_t *>(tile)[lane + 64 * i] = v[i];
            } else {
                const int gxc = min(max(xa + lane, 0), a.w - 1);
                for (int r = 0; r < R + 2; ++r) {
                    const int gy = min(max(gy0 - 3 + r, 0), a.h - 1);
                    tile[r * 64 + lane] = a.img[(size_t)gy * a.pitch + gxc];
                }
            }
            __syncthreads();

            HarrisTile t;
            t.col = tile + lane;
            t.gx = xa + lane;
            t.gy_tile0 = gy0 - 3;
            t.th = th;
            t.col_owned = t.gx >= gx0 && t.gx < gx0 + tw;
            t.col_valid = t.gx >= 2 && t.gx < a.w - 2;
            t.cell_x = t.col_owned ? (tx + t.gx - gx0) / a.cs : 0;
            t.cell_row0 = ty;
            t.keys = keys;
            HarrisRows st;
            harris_row<0, 1>(st, a, t, 0);   // warm-up: the pipeline fills over six rows
            harris_row<1, 1>(st, a, t, 1);
            harris_row<2, 2>(st, a, t, 2);
            harris_row<0, 2>(st, a, t, 3);
            float sm2 = harris_row<1, 3>(st, a, t, 4);   // scores of tile rows 2 and 3
            float sm1 = harris_row<2, 3>(st, a, t, 5);
            // steady state: three rows per branch-free iteration (their chains interleave), then the NMS of
            // owned rows n, n+1, n+2 (tile rows r-3 .. r-1); candidates are sparse, so one wave-uniform test
            // covers the triple.  A last partial triple runs on staged rows and fails the owned-row test.
            for (int r = 6; r < R; r += 3) {
                const int n = r - 6;
                const float sa = harris_row<0, 3>(st, a, t, r);
                const float sb = harris_row<1, 3>(st, a, t, r + 1);
                const float sc = harris_row<2, 3>(st, a, t, r + 2);
                const bool o0 = t.col_owned && (sm1 >= a.thr) && (sm1 != -INFINITY);
                const bool o1 = t.col_owned && (n + 1 < th) && (sa >= a.thr) && (sa != -INFINITY);
                const bool o2 = t.col_owned && (n + 2 < th) && (sb >= a.thr) && (sb != -INFINITY);
                if (__ballot(o0 || o1 || o2) != 0ull) {
                    const bool c0 = harris_nms(o0, sm2, sm1, sa);
                    const bool c1 = harris_nms(o1, sm1, sa, sb);
                    const bool c2 = harris_nms(o2, sa, sb, sc);
                    if (c0) harris_emit(a, t, n, sm1);
                    if (c1) harris_emit(a, t, n + 1, sa);
                    if (c2) harris_emit(a, t, n + 2, sb);
                }
                sm2 = sb;
                sm1 = sc;
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < ncl; i += 64) {
        const int cx = bx * a.nbx + i % a.nbx, cy = by * a.nby + i / a.nbx;
        if (cx < a.ncx && cy < a.ncy) a.cell_best[(size_t)cy * a.ncx + cx] = keys[i];
    }
}

// ---- FAST: 256-thread workgroup over LDS-staged tiles ------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void k_fast_cells(CornerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // carve: keys | score | pixels
    const int ncl = a.nbx * a.nby;
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    float *sc = reinterpret_cast<float *>(smem + (size_t)((ncl * 8 + 15) & ~15));
    uint8_t *px = reinterpret_cast<uint8_t *>(sc + SW * SH);

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

            // S3: score on the (tw+2)x(th+2) grid
            for (int i = tid; i < SW * SH; i += 256) {
                const int r = i / SW, c = i % SW;
                if (r < th + 2 && c < tw + 2) {
                    const int gx = gx0 - 1 + c, gy = gy0 - 1 + r;
                    float s = -INFINITY;
                    if (gx >= 3 && gy >= 3 && gx < a.w - 3 && gy < a.h - 3)
                        s = fast_score<N>(px + (r + HALO - 1) * PXW + (c + HALO - 1));
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
                    const uint32_t idx = ((uint32_t)gy << 16) | (uint32_t)gx;
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
// (a form that kept its ranges in registers between the counting pass and the write-out needed 96 VGPRs and lost 8 %).
constexpr uint32_t MG_THREADS = 512, MG_WAVES = MG_THREADS / 64;
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
template <uint32_t KT, uint32_t KC>
__global__ __launch_bounds__(MG_THREADS) void k_merge_compact(MergeArgs a)
{
    __shared__ uint64_t wave_tot[MG_WAVES];
    extern __shared__ uint32_t occ_words[];  // one byte per cell: holds a surviving tracked point
    VSA_SHORT_KERNEL_PRIORITY();
    uint8_t *occ = reinterpret_cast<uint8_t *>(occ_words);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    MG_STAMP(0);
    // Cells holding a surviving tracked point are marked in LDS, one byte a cell (plain stores: tracked points arrive
    // in cell order, so bits would have the lanes of a wave serialize on the same word with atomic ORs); the keys stay
    // untouched.  Only a cell grid too large for LDS is marked in place, in the keys themselves.
    const bool in_lds = a.flag_words != 0u;
    for (uint32_t i = tid; i < max(a.flag_words, 1u); i += MG_THREADS) occ_words[i] = 0u;   // word 0 is always there
    const bool has_t = a.tracked != nullptr && a.tracked_capacity != 0u;
    const uint32_t cap_t = has_t ? a.tracked_capacity : 0u;
    // the live count through a vector load (an index the compiler cannot see is zero): as a scalar load it would
    // have to land before the first vector load issues, one more memory round trip at the head of the kernel
    uint32_t zero = 0u;
    asm volatile("" : "+v"(zero));
    const uint32_t nt = has_t ? min(a.tracked_count[zero], cap_t) : 0u;
    const uint32_t ncells = (uint32_t)a.ncx * (uint32_t)a.ncy;
    // range sizes come from the capacity, not the live count
    const uint32_t rounds_t = (cap_t + MG_THREADS - 1) / MG_THREADS, rounds_c = (ncells + MG_THREADS - 1) / MG_THREADS;
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
    uint64_t wave_off = 0, total = 0;
#pragma unroll
    for (uint32_t i = 0; i < MG_WAVES; ++i) {
        const uint64_t t = wave_tot[i];
        wave_off += (i < wave) ? t : 0ull;
        total += t;
    }
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

inline void cell_block(uint32_t cs, int span_x, int span_y, int &nbx, int &nby)
{
    nbx = (int)((uint32_t)span_x / cs);
    nby = (int)((uint32_t)span_y / cs);
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
    if (p.image.width > 65535u || p.image.height > 65535u) return hipErrorInvalidValue;   // cell keys hold (y << 16) | x
    CornerArgs a{};
    a.img = p.image.ptr; a.pitch = p.image.pitch; a.w = (int)p.image.width; a.h = (int)p.image.height;
    a.mask = p.mask; a.mask_pitch = p.mask_pitch;
    a.cs = (int)cs;
    a.ncx = (int)((p.image.width + cs - 1) / cs);
    a.ncy = (int)((p.image.height + cs - 1) / cs);
    if (p.detector == kHarris) cell_block(cs, HS_USE, HS_BAND, a.nbx, a.nby);
    else cell_block(cs, TW, TH, a.nbx, a.nby);
    a.k = p.harris_k; a.thr = p.threshold;
    a.fast_n = (int)p.fast_n; a.fast_t = (int)p.threshold;
    a.cell_best = reinterpret_cast<uint64_t *>(p.scratch);

    const dim3 grid((a.ncx + a.nbx - 1) / a.nbx, (a.ncy + a.nby - 1) / a.nby);
    const size_t keys_b = ((size_t)a.nbx * a.nby * 8 + 15) & ~(size_t)15;
    if (p.detector == kHarris) {
        const size_t lds = (size_t)HS_R * 64 + keys_b;
        hipLaunchKernelGGL(k_harris_cells, dim3(grid.x * grid.y), dim3(64), lds, stream, a);
    } else {
        const size_t lds = keys_b + SW * SH * 4 + PXW * PXH;
        switch (p.fast_n) {
        case 9: hipLaunchKernelGGL(k_fast_cells<9>, grid, dim3(256), lds, stream, a); break;
        case 10: hipLaunchKernelGGL(k_fast_cells<10>, grid, dim3(256), lds, stream, a); break;
        case 11: hipLaunchKernelGGL(k_fast_cells<11>, grid, dim3(256), lds, stream, a); break;
        case 12: hipLaunchKernelGGL(k_fast_cells<12>, grid, dim3(256), lds, stream, a); break;
        default: return hipErrorInvalidValue;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    MergeArgs m{};
    m.item = p.item; m.tracked = p.tracked; m.tracked_count = p.tracked_count;
    m.tracked_capacity = p.tracked_capacity;
    m.out = p.out; m.out_count = p.out_count; m.out_capacity = p.out_capacity;
    m.num_corners = p.num_corners; m.cell_best = a.cell_best;
    m.out_count_host = p.out_count_host; m.num_corners_host = p.num_corners_host;
    m.w = a.w; m.h = a.h; m.cs = a.cs; m.ncx = a.ncx; m.ncy = a.ncy;
    const uint32_t ncells = (uint32_t)a.ncx * (uint32_t)a.ncy;
    if (ncells == 0u) return hipErrorInvalidValue;
    m.flag_words = ncells <= 60u * 1024u ? (ncells + 3u) / 4u : 0u;   // beyond 60 K cells the keys are marked in place
    const size_t lds = (size_t)(m.flag_words ? m.flag_words : 1u) * 4u;
    m.cs_magic = cs > 1u ? (uint32_t)((0x100000000ull + cs - 1u) / cs) : 0u;
    hipLaunchKernelGGL((k_merge_compact<MG_KT, MG_KC>), dim3(1), dim3(MG_THREADS), lds, stream, m);
    return hipGetLastError();
}

}  // namespace vsa

#ifdef VSA_MG_STAMPS
extern "C" int visualslam_amd_debug_mg_stamps(uint64_t *host, size_t n_u64)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vsa::g_mg_stamps), n_u64 * sizeof(uint64_t));
}
#endif

