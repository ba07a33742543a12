// harris_cells.hpp -- the Harris-track CELL PASS as a device function: Harris response, 3x3 non-maximum suppression and each
// detector cell's best (score, lowest raster index) key, one wave per rectangle of whole cells.  It is the body of
// k_harris_cells (corner_track.hip) and of the second half of the fused PyrLK + cell-pass launch (pyrlk.hip): the cell pass
// reads only the image, so it does not have to wait for PyrLK -- only the merge pass needs the tracked points.
// Included by those two translation units only; everything here has internal linkage.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace vsa {
namespace {

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

// ---- Harris: one wave per rectangle of cells, lane = column ----------------------------------
constexpr int HS_USE = 55;    // owned columns per strip: 64 lanes - 3 right halo - (3..6) left halo (dword-aligned origin)
constexpr int HS_ROWS = 48;   // owned rows per chunk (LDS tile height - 6)
// Preferred owned rows per wave (a launch parameter: the kernel takes the cell block from its arguments).  A wave computes 6 halo
// rows beside the rows it owns, so taller bands mean less work in total (cell 18: bands of 36 rows stream 42 rows for 36, bands of
// 18 stream 24 for 18: -12 % vector instructions) and fewer, longer waves.
//   HS_BAND       the cell pass launched by itself (thin layer, immediate mode): short waves, the launch ends sooner (12 us at
//                 1080p against 17 us with 36-row bands);
//   HS_BAND_FUSED the cell pass inside the PyrLK launch of a graph: with many camera streams on the GPU the instruction count is
//                 what matters (+5 % frames/s on the 32-stream bench for 36 against 24; 54 gains nothing more) --
//                 profiles/r02_kernel_experiments.txt.
constexpr int HS_BAND = 24, HS_BAND_FUSED = 36;
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

// One wave = workgroup `bid` of `nblocks`: the rectangle of cells it owns, streamed tile by tile.  smem: HS_R * 64 bytes of
// pixel tile followed by one 64-bit key per owned cell (harris_cells_lds()).
__device__ __forceinline__ void harris_cells_wave(const CornerArgs &a, uint32_t bid, uint32_t nblocks, uint8_t *smem)
{
    // carve: pixel tile (HS_R rows x 64 columns) | keys
    uint8_t *tile = smem;
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem + HS_R * 64);
    const int ncl = a.nbx * a.nby;
    const int lane = threadIdx.x;

    const uint32_t gxn = (uint32_t)(a.ncx + a.nbx - 1) / (uint32_t)a.nbx;
    // each XCD owns a contiguous band of cell blocks (halo rows/columns are shared inside one L2)
    const uint32_t item = xcd_contiguous_item(bid, nblocks);
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

inline void cell_block(uint32_t cs, int span_x, int span_y, int &nbx, int &nby)
{
    nbx = (int)((uint32_t)span_x / cs);
    nby = (int)((uint32_t)span_y / cs);
    if (nbx < 1) nbx = 1;
    if (nby < 1) nby = 1;
}

// grid and LDS of the cell pass for an image (host side)
inline void harris_cells_setup(const CornerTrackParams &p, CornerArgs &a, uint32_t &blocks, size_t &lds, int band = HS_BAND)
{
    const uint32_t cs = p.cell_size;
    a = CornerArgs{};
    a.img = p.image.ptr; a.pitch = p.image.pitch; a.w = (int)p.image.width; a.h = (int)p.image.height;
    a.mask = p.mask; a.mask_pitch = p.mask_pitch;
    a.cs = (int)cs;
    a.ncx = (int)((p.image.width + cs - 1) / cs);
    a.ncy = (int)((p.image.height + cs - 1) / cs);
    cell_block(cs, HS_USE, band, a.nbx, a.nby);
    a.k = p.harris_k; a.thr = p.threshold;
    a.fast_n = (int)p.fast_n; a.fast_t = (int)p.threshold;
    a.cell_best = reinterpret_cast<uint64_t *>(p.scratch);
    blocks = (uint32_t)((a.ncx + a.nbx - 1) / a.nbx) * (uint32_t)((a.ncy + a.nby - 1) / a.nby);
    lds = (size_t)HS_R * 64 + (((size_t)a.nbx * a.nby * 8 + 15) & ~(size_t)15);
}

}  // namespace
}  // namespace vsa
