// fast_cells.hpp -- the FAST-track CELL PASS as a device function: FAST-N strength, 3x3 non-maximum suppression and each detector
// cell's best (strength, lowest raster index) key, one wave per rectangle of whole cells, lane = image column.  It is the body of
// k_fast_cells (corner_track.hip) and of the second half of the fused PyrLK + cell-pass launch (pyrlk.hip), like the Harris pass
// in harris_cells.hpp, whose CornerArgs, key format and cell-block geometry it shares.
//
// Reference call sites replaced: nvxFastTrackNode / nvxuFastTrack / nvxcuFastTrack (demos/feature_tracker/feature_tracker.cpp:
// 334-338,426-430, demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:314-317).  Semantics: oracle/vxoracle.c (fast strength,
// cell selection); results are bit-identical.
//
// Design (gfx950).  The pass is bound by vector-instruction issue: the strength of one pixel is ~80 three-input minima / maxima,
// whatever the image holds.  So everything around them is arranged to cost little:
//   * a wave owns a strip of 56 columns (64 lanes minus a halo of 4 on either side: circle radius 3 + 1 for the NMS) and streams
//     down its rows; the pixel tile is staged once in LDS (aligned dwords, row stride 68 bytes so that lane 0 can sit on any
//     byte) and every pixel is read from LDS exactly once, by the lane of its column;
//   * the 16 circle pixels come from a register ring of the last rows and their copies shifted by 1, 2 and 3 lanes (DPP wave
//     shifts), not from 16 LDS reads with their address arithmetic;
//   * two rows are processed at once in the 16-bit halves of every register (packed minima / maxima: tools/minmax_rate.hip);
//   * the arcs' minima and maxima are running minima over the circle's halves (57 two-input operations per polarity, not 79);
//   * minima and maxima run on the raw pixel values and the centre is subtracted once at the end (min(v_i - p) = min(v_i) - p);
//   * strengths stay integers (-1 = no corner); row pairs are unrolled by the ring length so every ring index is a constant.
#pragma once

#include <hip/hip_runtime.h>

#include "harris_cells.hpp"

namespace vsa {
namespace {

constexpr int FS_HALO = 4;                    // circle radius 3 + 1 (the NMS reads the strengths of the 8 neighbours)
constexpr int FS_USE = 64 - 2 * FS_HALO;      // owned columns per strip
constexpr int FS_ROWS = 54;                   // owned rows per chunk
constexpr int FS_STRIDE = 68;                 // LDS row stride (bytes): 64 lanes + the 0..3 bytes lane 0 lies past an aligned dword
constexpr int FS_R = FS_ROWS + 2 * FS_HALO;   // staged rows per chunk
constexpr int FS_LDS_ROWS = FS_R + 8;         // the last unrolled group of row pairs may read (never use) rows past the staged ones
constexpr int FS_NLD = (FS_R * (FS_STRIDE / 4) + 63) / 64;
constexpr int FS_TILE_BYTES = (FS_LDS_ROWS * FS_STRIDE + 15) & ~15;
// preferred owned rows per wave, as for the Harris pass (harris_cells.hpp): short waves when the pass is launched by itself,
// taller bands (less halo work per owned row) inside the PyrLK launch
constexpr int FS_BAND = 18, FS_BAND_FUSED = 36;

// Two image rows travel in one register, as the two 16-bit halves of a dword (pixel values and strengths fit easily), and every
// minimum, maximum and difference is a packed instruction (v_pk_min_i16 ...): measured on this GPU (tools/minmax_rate.hip) a
// 32-bit integer or float min / max issues at half rate whether it has two inputs or three, the packed 16-bit forms at the same
// half rate for two values, so a pair of rows costs what one row costs in 32-bit arithmetic -- in cycles and in instructions.
typedef short fs_pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pk_min(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(fs_pair, a), __builtin_bit_cast(fs_pair, b))); }
__device__ __forceinline__ int pk_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(fs_pair, a), __builtin_bit_cast(fs_pair, b))); }
__device__ __forceinline__ int pk_sub(int a, int b) { return __builtin_bit_cast(int, __builtin_bit_cast(fs_pair, a) - __builtin_bit_cast(fs_pair, b)); }
__device__ __forceinline__ int pk_add(int a, int b) { return __builtin_bit_cast(int, __builtin_bit_cast(fs_pair, a) + __builtin_bit_cast(fs_pair, b)); }

// Row pairs.  E_m = (row 2m, row 2m+1) is packed from the tile; O_m = (row 2m+1, row 2m+2) is cut out of two consecutive E's
// (v_alignbit).  A pair of centre rows (y, y+1) needs, for dy = -3 .. 3, the pair that starts at row y + dy: E and O alternate.
// Kept of each: the pair itself (c) and its copies shifted by 1, 2, 3 lanes to either side, as far as the circle uses them.
struct FastRing {
    int Ec[4], El1[4], Er1[4], El2[4], Er2[4], El3[4], Er3[4];
    int Oc[4], Ol2[4], Or2[4], Ol3[4], Or3[4];
};

// FAST-N strength of two pixels at once (the 16-bit halves of every operand) from the 16 circle pixels v[0..15] (clockwise from 12
// o'clock) and the centre p: the largest threshold for which the segment test still passes = max(max over arcs of min(v) - p,
// p - min over arcs of max(v)) - 1, -1 when no arc is uniformly brighter or darker.  The minima (maxima) of all 16 arcs of N
// consecutive pixels come from running minima over the circle's two halves (van Herk / Gil-Werman): S[i] = min(v[i] .. end of
// i's half), P[j] = min(start of j's half .. v[j]); an arc starting at i is the rest of i's half, then a prefix of the other half
// (and, for N > 9 and a late start, all of it plus a short prefix of the first half again).  57 two-input operations per polarity
// for N = 9 where 3-pixel groups (min3 of min3) need 79.
template <int N, bool WANT_MIN>
__device__ __forceinline__ int fast_arcs(const int (&v)[16], int p)
{
    auto op = [](int x, int y) { return WANT_MIN ? pk_min(x, y) : pk_max(x, y); };
    auto opposite = [](int x, int y) { return WANT_MIN ? pk_max(x, y) : pk_min(x, y); };
    int S[16], P[16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        S[8 * h + 7] = v[8 * h + 7];
#pragma unroll
        for (int i = 6; i >= 0; --i) S[8 * h + i] = op(v[8 * h + i], S[8 * h + i + 1]);
        P[8 * h] = v[8 * h];
#pragma unroll
        for (int j = 1; j < 7; ++j) P[8 * h + j] = op(P[8 * h + j - 1], v[8 * h + j]);
        P[8 * h + 7] = S[8 * h];      // the whole half, either way round
    }
    int best = p;                     // max over arcs of the arc minimum, but never below p (min: never above p)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int h = i >> 3, rest = N - (8 - (i & 7));      // pixels of the arc beyond i's half: 1 .. N - 1
        int w = S[i];
        if (rest <= 8) w = op(w, P[8 * (1 - h) + rest - 1]);
        else w = op(op(w, S[8 * (1 - h)]), P[8 * h + rest - 9]);
        best = opposite(best, w);
    }
    return best;
}

template <int N>
__device__ __forceinline__ int fast_strength(const int (&v)[16], int p)
{
    static_assert(N >= 9 && N <= 12, "FAST-9 .. FAST-12");
    const int best_lo = fast_arcs<N, true>(v, p);      // >= p
    // the two polarities one after the other, and one row pair after the other (fast_rows): interleaved by the scheduler the
    // temporaries of several trees are live at once and the kernel needs far more registers
    __builtin_amdgcn_sched_barrier(0);
    const int best_hi = fast_arcs<N, false>(v, p);     // <= p
    return pk_add(pk_max(pk_sub(best_lo, p), pk_sub(p, best_hi)), (int)0xffffffffu);      // - 1 in both halves
}

struct FastTile {
    const uint8_t *col;        // LDS: this lane's column of the pixel tile (row stride FS_STRIDE)
    int gx;                    // image column of this lane
    int gy_tile0;              // image row of tile row 0
    int th;                    // owned rows
    bool col_valid, col_owned;
    int cell_x, cell_row0;
    int thr;                   // FAST threshold: a corner needs strength >= thr
    uint64_t *keys;
};

// Tile rows 2m and 2m+1 enter the ring as E_m (slot PH = m % 4) with its six shifted copies, and with E_(m-1) they give O_(m-1).
template <int PH>
__device__ __forceinline__ void fast_load_rows(FastRing &g, const FastTile &t, int m)
{
    constexpr int PV = (PH + 3) % 4;
    const int e = (int)t.col[(2 * m) * FS_STRIDE] | ((int)t.col[(2 * m + 1) * FS_STRIDE] << 16);
    g.Ec[PH] = e;
    g.El1[PH] = left_i(e);         g.Er1[PH] = right_i(e);
    g.El2[PH] = left_i(g.El1[PH]); g.Er2[PH] = right_i(g.Er1[PH]);
    g.El3[PH] = left_i(g.El2[PH]); g.Er3[PH] = right_i(g.Er2[PH]);
    g.Oc[PV] = (int)__builtin_amdgcn_alignbit((unsigned)g.Ec[PH], (unsigned)g.Ec[PV], 16);
    g.Ol2[PV] = (int)__builtin_amdgcn_alignbit((unsigned)g.El2[PH], (unsigned)g.El2[PV], 16);
    g.Or2[PV] = (int)__builtin_amdgcn_alignbit((unsigned)g.Er2[PH], (unsigned)g.Er2[PV], 16);
    g.Ol3[PV] = (int)__builtin_amdgcn_alignbit((unsigned)g.El3[PH], (unsigned)g.El3[PV], 16);
    g.Or3[PV] = (int)__builtin_amdgcn_alignbit((unsigned)g.Er3[PH], (unsigned)g.Er3[PV], 16);
}

// Strengths of tile rows 2m-3 and 2m-2 (the pair O_(m-2)) once E_m is in the ring at slot PH, as the halves of one dword.
template <int N, int PH>
__device__ __forceinline__ int fast_pair_strength(const FastRing &g)
{
    constexpr int A = (PH + 1) % 4, B = (PH + 2) % 4, C = (PH + 3) % 4, D = PH;   // slots of E/O_(m-3), _(m-2), _(m-1), E_m
    // circle pixels clockwise from 12 o'clock: (dx, dy) = (0,-3) (1,-3) (2,-2) (3,-1) (3,0) (3,1) (2,2) (1,3) (0,3) (-1,3) (-2,2) (-3,1)
    // (-3,0) (-3,-1) (-2,-2) (-1,-3); the lane of column x+1 is the "right" neighbour.  The pair starting dy rows from the centre
    // pair: dy = -3: E[A], -2: O[A], -1: E[B], 0: O[B], 1: E[C], 2: O[C], 3: E[D]
    const int v[16] = { g.Ec[A], g.Er1[A], g.Or2[A], g.Er3[B], g.Or3[B], g.Er3[C], g.Or2[C], g.Er1[D],
                        g.Ec[D], g.El1[D], g.Ol2[C], g.El3[C], g.Ol3[B], g.El3[B], g.Ol2[A], g.El1[A] };
    return fast_strength<N>(v, g.Oc[B]);
}

__device__ __forceinline__ void fast_emit(const CornerArgs &a, const FastTile &t, int n /* owned row */, int s)
{
    const int gy = t.gy_tile0 + FS_HALO + n;
    if (a.mask && a.mask[(size_t)gy * a.mask_pitch + t.gx] == 0) return;
    const int cell = ((t.cell_row0 + n) / a.cs) * a.nbx + t.cell_x;
    const uint32_t idx = ((uint32_t)gy << 16) | (uint32_t)t.gx;
    const uint64_t key = ((uint64_t)orderable((float)s) << 32) | (uint64_t)(~idx);
    atomicMax(reinterpret_cast<unsigned long long *>(&t.keys[cell]), (unsigned long long)key);
}

// What the 3 x 3 non-maximum suppression keeps of the last strength rows: of the centre row its strengths and their left / right
// neighbours, of the row above only the maximum over each pixel's three columns.  A pixel survives when no neighbour that comes
// before it in raster order is >= and no later one is > (plateau-safe, as in the oracle): the three of the row above and the
// three of the row below enter only through their maxima, so a row costs two wave shifts where nine compares would need six.
struct FastNms {
    int c, cl, cr, hm_c, hm_up;
};

// the strengths s0 of tile row r0 have arrived: decide tile row r0 - 1 (owned row n = r0 - 1 - FS_HALO)
__device__ __forceinline__ void fast_nms_row(const CornerArgs &a, const FastTile &t, int r0, int s0, FastNms &m)
{
    const int l0 = left_i(s0), r0s = right_i(s0);
    const int hm_dn = max(max(l0, s0), r0s);
    const int n = r0 - 1 - FS_HALO;
    const bool cand = t.col_owned & (n >= 0) & (n < t.th) & (m.c >= t.thr) & (m.c >= 0) &
                      !(m.hm_up >= m.c) & !(m.cl >= m.c) & !(m.cr > m.c) & !(hm_dn > m.c);
    if (cand) fast_emit(a, t, n, m.c);
    m.hm_up = m.hm_c;
    m.c = s0; m.cl = l0; m.cr = r0s; m.hm_c = hm_dn;
}

// Steady state, one pair of tile rows: rows 2m and 2m+1 enter the ring, the strengths of rows 2m-3 and 2m-2 follow, and with them
// the NMS of rows 2m-4 and 2m-3 is decided.
template <int N, int PH>
__device__ __forceinline__ void fast_rows(FastRing &g, const CornerArgs &a, const FastTile &t, int m, FastNms &nm)
{
    fast_load_rows<PH>(g, t, m);
    const int s = fast_pair_strength<N, PH>(g);
    const int ra = 2 * m - 3, gya = t.gy_tile0 + ra;
    int sa = __builtin_amdgcn_sbfe(s, 0, 16), sb = s >> 16;      // sign-extended halves
    sa = (t.col_valid && gya >= 3 && gya < a.h - 3) ? sa : -1;      // -1 where the circle leaves the image
    sb = (t.col_valid && gya + 1 >= 3 && gya + 1 < a.h - 3) ? sb : -1;
    fast_nms_row(a, t, ra, sa, nm);
    fast_nms_row(a, t, ra + 1, sb, nm);
    __builtin_amdgcn_sched_barrier(0);
}

// One wave = workgroup `bid` of `nblocks`: the rectangle of cells it owns, streamed chunk by chunk.  smem: FS_LDS_ROWS * FS_STRIDE
// (rounded up to 16) bytes of pixel tile followed by one 64-bit key per owned cell (fast_cells_setup()).
template <int N>
__device__ __forceinline__ void fast_cells_wave(const CornerArgs &a, uint32_t bid, uint32_t nblocks, uint8_t *smem)
{
    uint8_t *tile = smem;
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem + FS_TILE_BYTES);
    const int ncl = a.nbx * a.nby;
    const int lane = threadIdx.x;

    const uint32_t gxn = (uint32_t)(a.ncx + a.nbx - 1) / (uint32_t)a.nbx;
    const uint32_t item = xcd_contiguous_item(bid, nblocks);      // each XCD owns a contiguous band of cell blocks
    const int bx = (int)(item % gxn), by = (int)(item / gxn);
    const int rx0 = bx * a.nbx * a.cs, ry0 = by * a.nby * a.cs;
    const int rw = min(a.nbx * a.cs, a.w - rx0), rh = min(a.nby * a.cs, a.h - ry0);
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.img) | a.pitch) & 3u) == 0;

    for (int i = lane; i < ncl; i += 64) keys[i] = 0;

    for (int ty = 0; ty < rh; ty += FS_ROWS) {
        for (int tx = 0; tx < rw; tx += FS_USE) {
            const int tw = min(FS_USE, rw - tx), th = min(FS_ROWS, rh - ty);
            const int gx0 = rx0 + tx, gy0 = ry0 + ty;
            const int xorg = gx0 - FS_HALO;                       // image column of lane 0
            const int R = th + 2 * FS_HALO;                       // tile rows: image rows gy0-4 .. gy0+th+3
            int off = 0;
            __syncthreads();                                      // previous tile fully consumed (orders the key init too)

            // stage the tile: rows replicate-clamped; columns outside the image hold arbitrary in-bounds bytes (every strength
            // whose circle touches them is invalid by definition)
            if (aligned) {
                const int xa = xorg & ~3, xmax = (int)a.pitch - 4;
                off = xorg - xa;
                constexpr int ND = FS_STRIDE / 4;
                uint32_t v[FS_NLD];
                // which (row, dword) a lane's i-th load is never changes, but hoisted out of the tile loops those 2 x 17 values
                // would be live across the streamed pass: derived here from a copy of the lane number the compiler cannot see through
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int i = 0; i < FS_NLD; ++i) {
                    const int idx = ln + 64 * i, r = idx / ND, d = idx - r * ND;
                    const int gy = min(max(gy0 - FS_HALO + r, 0), a.h - 1);
                    const int X = min(max(xa + 4 * d, 0), xmax);
                    v[i] = 0;
                    if (64 * i < R * ND)   // wave-uniform
                        v[i] = *reinterpret_cast<const uint32_t *>(a.img + (uint32_t)gy * a.pitch + (uint32_t)X);
                }
#pragma unroll
                for (int i = 0; i < FS_NLD; ++i)
                    if (64 * i < R * ND) reinterpret_cast<uint32_t *>(tile)[ln + 64 * i] = v[i];
            } else {
                const int gxc = min(max(xorg + lane, 0), a.w - 1);
                for (int r = 0; r < R; ++r) {
                    const int gy = min(max(gy0 - FS_HALO + r, 0), a.h - 1);
                    tile[r * FS_STRIDE + lane] = a.img[(size_t)gy * a.pitch + gxc];
                }
            }
            __syncthreads();

            FastTile t;
            t.col = tile + off + lane;
            t.gx = xorg + lane;
            t.gy_tile0 = gy0 - FS_HALO;
            t.th = th;
            t.col_owned = lane >= FS_HALO && lane < FS_HALO + tw;
            t.col_valid = t.gx >= 3 && t.gx < a.w - 3;
            t.cell_x = t.col_owned ? (tx + lane - FS_HALO) / a.cs : 0;
            t.cell_row0 = ty;
            t.thr = a.fast_t;
            t.keys = keys;
            FastRing g;
            fast_load_rows<0>(g, t, 0);     // warm-up: three row pairs fill the ring (O_(-1), cut from an unset E, is never used)
            fast_load_rows<1>(g, t, 1);
            fast_load_rows<2>(g, t, 2);
            FastNms nm{ -1, -1, -1, -1, -1 };
            // four row pairs per iteration, so that every ring slot is a compile-time constant; the last group may run on rows
            // past the staged ones, whose strengths belong to no owned row.  Pair m decides owned rows 2m - 8 and 2m - 7.
            for (int m = 3; 2 * m - 8 < th; m += 4) {
                fast_rows<N, 3>(g, a, t, m, nm);
                fast_rows<N, 0>(g, a, t, m + 1, nm);
                fast_rows<N, 1>(g, a, t, m + 2, nm);
                fast_rows<N, 2>(g, a, t, m + 3, nm);
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < ncl; i += 64) {
        const int cx = bx * a.nbx + i % a.nbx, cy = by * a.nby + i / a.nbx;
        if (cx < a.ncx && cy < a.ncy) a.cell_best[(size_t)cy * a.ncx + cx] = keys[i];
    }
}

// grid and LDS of the FAST cell pass for an image (host side)
inline void fast_cells_setup(const CornerTrackParams &p, CornerArgs &a, uint32_t &blocks, size_t &lds, int band = FS_BAND)
{
    harris_cells_setup(p, a, blocks, lds);      // the fields common to both detectors
    cell_block(p.cell_size, FS_USE, band, a.nbx, a.nby);
    blocks = (uint32_t)((a.ncx + a.nbx - 1) / a.nbx) * (uint32_t)((a.ncy + a.nby - 1) / a.nby);
    lds = (size_t)FS_TILE_BYTES + (((size_t)a.nbx * a.nby * 8 + 15) & ~(size_t)15);
}

}  // namespace
}  // namespace vsa
