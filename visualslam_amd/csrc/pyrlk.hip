// pyrlk.hip -- sparse pyramidal Lucas-Kanade optical flow, one wavefront per feature.
//
// Reference call sites replaced: vxOpticalFlowPyrLKNode / nvxcuOpticalFlowPyrLK
// (demos/feature_tracker/feature_tracker.cpp:402-412, demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:297-301,
// samples/object_tracker_nvxcu/object_tracker_keypoint_nvxcu.cpp:1791-1828).  Semantics and the exact
// floating-point operation order: oracle/vxoracle.c (lk_track_one) -- results are bit-identical.
//
// Design (MI355X, gfx950).  The kernel is bound by VALU issue, so everything is arranged to spend few
// vector instructions:
//   * a 64-lane wave owns one feature for all pyramid levels, coarse to fine; window pixels are
//     dealt to lanes as (row, segment of `seglen` consecutive pixels), so a lane streams along LDS
//     rows and every column it touches serves two neighbouring pixels (bilinear sampling is
//     y-first: one vertical lerp per column, one horizontal lerp per pixel);
//   * tiles are fetched as ALIGNED DWORDS (4 pixels per lane per load, all loads of a tile issued
//     back to back into registers), converted once and written to LDS as one ds_write_b128 each;
//     which (row, dword) a lane fetches never changes, so that index math is done once per wave.
//     Tiles that poke out of the image, and pyramids whose base or pitch is not 4-byte aligned,
//     take per-byte clamped loads into the same registers (replicate border);
//   * latency hiding inside the wave: the search tile's loads are in flight while the template is
//     resampled, and the next level's template loads are in flight while this level iterates;
//   * the template is resampled at the feature's sub-pixel offset once per value (a lane resamples its own
//     window row and exchanges rows with its neighbours through LDS), then differentiated (Scharr/32,
//     separable, constants folded); I, Ix, Iy stay in REGISTERS for all iterations;
//   * pair math is scalar f32 (the packed-f32 VALU forms issue slower than two plain ones on gfx950);
//   * sums are combined in the butterfly order the oracle mirrors, two or three sums per butterfly (lanes of
//     different lane % 4 carry different sums; DPP quad permutes and row rotations, then the gfx950 row swaps
//     v_permlane16_swap / v_permlane32_swap across rows); the 2x2 normal equations are solved wave-uniformly.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "kernels.hpp"
#include "harris_cells.hpp"
#include "fast_cells.hpp"

namespace vsa {
namespace {

#ifndef VSA_LK_VGPR_HALF
#define VSA_LK_VGPR_HALF 52    // amdgpu_num_vgpr argument of the 21 x 21 kernels: half of the register cap (see k_pyrlk_104)
#endif
constexpr int LK_MARGIN = 3;  // search tile margin M (pixels) on each side of the window (a position that walks further restages the tile)

// Staged-tile geometry.  A tile row of `px` pixels starting anywhere in a dword needs (px + 6) / 4 aligned dwords.  That count is
// rounded up to a power of two when it costs no extra load per lane: which (row, dword) a lane's i-th slot is then comes from a
// shift and a mask of the lane number, with constant strides between slots (21 x 21: 8 dwords for both tiles, 3 + 4 loads per lane
// and level), instead of a division by 7 or 9 for each of the eight slots at every issue and commit.
constexpr int ceil_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
constexpr int tile_slots_per_lane(int rows, int nd) { return (rows * nd + 63) / 64; }
constexpr int tile_dwords(int rows, int px)
{
    const int nd = (px + 6) >> 2, p2 = ceil_pow2(nd);
    return (p2 <= 64 && tile_slots_per_lane(rows, p2) <= tile_slots_per_lane(rows, nd)) ? p2 : nd;
}
constexpr int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
template <int WMAX>
struct LKTiles {
    static constexpr int PROWS = WMAX + 3, JROWS = WMAX + 1 + 2 * LK_MARGIN;                  // template / search tile rows (= pixels per row)
    static constexpr int NDP = tile_dwords(PROWS, PROWS), NDJ = tile_dwords(JROWS, JROWS);    // dwords per tile row
    static constexpr int NLP = tile_slots_per_lane(PROWS, NDP), NLJ = tile_slots_per_lane(JROWS, NDJ);
    static constexpr int PS4 = NDP | 1, JS4 = NDJ | 1;                                        // LDS row strides (float4), odd
    static constexpr int PS = 4 * PS4, JS = 4 * JS4;                                          // ... in floats
    static constexpr int PFLOATS = ((NLP * 64 + NDP - 1) / NDP) * PS, JFLOATS = ((NLJ * 64 + NDJ - 1) / NDJ) * JS;   // tile regions
};

struct LKArgs {
    PyramidView prev, curr;
    int item;
    const void *prev_pts;
    const uint32_t *prev_count;
    const void *estimates;
    void *out;
    uint32_t *out_count;
    uint32_t *out_count_host;
    uint32_t capacity;
    int use_eps;
    uint32_t max_it;
    float eps2;
    int win, half, segs, seglen, lanes;
    int calc_error;
};

struct KeypointF {
    float x, y, strength, scale, orientation;
    int32_t tracking_status;
    float error;
};

#ifdef VSA_LK_STAMPS   // development instrumentation: per-wave start/end cycle stamps and placement
__device__ uint64_t g_lk_stamps[8 * 8192];
#endif

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Wave sums.  The summation order is the pairwise butterfly (offsets 1,2,4,8,16,32) the oracle mirrors (ORC_SUM_WAVE64):
// ((l0+l1)+(l2+l3)) per quad, (q0+q1)+(q2+q3) per row of 16, (r0+r1)+(r2+r3) per wave; float addition commutes exactly, so
// which partner supplies which operand is immaterial.  DPP and cross-lane instructions issue at half the rate of plain
// f32 arithmetic on gfx950 (tools/valu_kinds.hip), so several sums share one butterfly: after the first step each value
// only needs half of the lanes, after the second a quarter.  Lanes with (lane & 1) == 0 carry the first value, odd lanes
// the second (wave_sum3: lanes with lane & 2 the third); the remaining steps pair lanes of equal lane % 4 -- row
// rotations by 4 and 8, whose result is in canonical order in lanes 12..15 of each row, then the gfx950 row swaps
// (v_permlane16_swap / v_permlane32_swap: lane i of row r meets lane i of row r^1, then of row r^2).  tools/wave_sum2_probe.hip
// checks the scheme bit for bit against separate butterflies on the GPU.  All 64 lanes must be active.
typedef unsigned lk_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float wave_sum_rows(float z)
{
    z += dpp_mov<0x124, 0xf>(z);   // row_ror:4: quad q meets quad q-1 -> (q3+q2) in lanes 12..15, (q1+q0) in lanes 4..7
    z += dpp_mov<0x128, 0xf>(z);   // row_ror:8: lanes 12..15 now hold (q3+q2)+(q1+q0)
    lk_v2u s = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    z = __uint_as_float(s.x) + __uint_as_float(s.y);   // rows 0,1: r0+r1; rows 2,3: r2+r3
    s = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    return __uint_as_float(s.x) + __uint_as_float(s.y);   // (r0+r1)+(r2+r3) in every row
}
__device__ __forceinline__ float lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ void wave_sum2(float a, float b, bool lane_bit0, float &sa, float &sb)
{
    a += dpp_mov<0xB1, 0xf>(a);    // quad_perm [1,0,3,2]
    b += dpp_mov<0xB1, 0xf>(b);
    float z = lane_bit0 ? b : a;
    z += dpp_mov<0x4E, 0xf>(z);    // quad_perm [2,3,0,1]: partner of the same parity
    z = wave_sum_rows(z);
    sa = lane_value(z, 62);
    sb = lane_value(z, 63);
}
__device__ __forceinline__ void wave_sum3(float a, float b, float c, bool lane_bit0, bool lane_bit1, float &sa, float &sb, float &sc)
{
    a += dpp_mov<0xB1, 0xf>(a);
    b += dpp_mov<0xB1, 0xf>(b);
    c += dpp_mov<0xB1, 0xf>(c);
    float z = lane_bit0 ? b : a;
    z += dpp_mov<0x4E, 0xf>(z);
    c += dpp_mov<0x4E, 0xf>(c);
    z = lane_bit1 ? c : z;         // lanes 0,1 of a quad: a, b; lanes 2,3: c
    z = wave_sum_rows(z);
    sa = lane_value(z, 60);
    sb = lane_value(z, 61);
    sc = lane_value(z, 63);
}
// one value alone (calcError)
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1, 0xf>(v);
    v += dpp_mov<0x4E, 0xf>(v);
    return lane_value(wave_sum_rows(v), 63);
}

// ---- tile staging ----------------------------------------------------------------------------
// A tile with origin (x0,y0) is fetched as rows of ND aligned dwords starting at column
// xa = floor4(x0).  Slot s = lane + 64*i is (row s / ND, dword s % ND) and lands in LDS as four floats at float4
// index row*LS4 + dword: tile element (r,c) lives at r*4*LS4 + (x0-xa) + c.  LS4 >= ND is the LDS row stride in
// float4 units, chosen ODD so that the rows a wave reads together start in different banks (a stride of 8 float4 =
// 32 floats put every other row on the same banks: 43% of PyrLK's LDS cycles were bank conflicts).  Slots past
// the last tile row fetch (clamped) rows below it; they only fill LDS that nobody interprets.
template <int NL>
struct TileRegs {
    uint32_t v[NL];
};

// Which (row, dword) slot i of a lane is never changes, but holding the three indices of all eight slots in registers
// costs 24 VGPRs that are live across the iteration loop.  They are recomputed where a tile is issued or committed
// (a division by a constant and two multiply-adds per slot, outside the iteration loop) from an opaque copy of the lane
// number, so that the compiler cannot hoist them back out of the level loop: the registers go to the iteration loop's
// cache of search rows instead, and the kernel stays at 4 waves per SIMD with room for other kernels' waves.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// 1 / 2^l, exactly (what 1.0f / (float)(1u << l) rounds to, without the division's dozen instructions)
__device__ __forceinline__ float pow2_recip(int l) { return __uint_as_float((uint32_t)(127 - l) << 23); }

// fast: the tile's dwords lie inside the row allocation and inside the image columns, and the level is
// 4-byte aligned -> one dword load per slot.  Otherwise four clamped byte loads (replicate border).
template <int NL, int ND>
__device__ __forceinline__ void tile_issue(TileRegs<NL> &t, int lane, const uint8_t *img, uint32_t pitch,
                                           int w, int h, int x0, int y0, bool fast)
{
    const int xa = x0 & ~3;
    constexpr bool POW2 = (ND & (ND - 1)) == 0;
    lane = opaque(lane);   // rows and offsets of the slots are re-derived here (an add each), not held across the level loop
    if (fast) {   // wave-uniform
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = lane + 64 * i;
            const int row = POW2 ? (lane >> ilog2(ND)) + (64 / ND) * i : idx / ND;
            const int col4 = POW2 ? 4 * (lane & (ND - 1)) : 4 * (idx - row * ND);
            const int gy = min(max(y0 + row, 0), h - 1);
            t.v[i] = *reinterpret_cast<const uint32_t *>(img + ((uint32_t)gy * pitch + (uint32_t)(xa + col4)));
        }
    } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = lane + 64 * i;
            const int row = POW2 ? (lane >> ilog2(ND)) + (64 / ND) * i : idx / ND;
            const int col4 = POW2 ? 4 * (lane & (ND - 1)) : 4 * (idx - row * ND);
            const int gy = min(max(y0 + row, 0), h - 1);
            const uint8_t *rowp = img + (uint32_t)gy * pitch;
            const int X = xa + col4;
            const uint32_t b0 = rowp[min(max(X, 0), w - 1)], b1 = rowp[min(max(X + 1, 0), w - 1)];
            const uint32_t b2 = rowp[min(max(X + 2, 0), w - 1)], b3 = rowp[min(max(X + 3, 0), w - 1)];
            t.v[i] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
    }
}

template <int NL, int ND, int LS4>
__device__ __forceinline__ void tile_commit(const TileRegs<NL> &t, int lane, float *dst)
{
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    constexpr bool POW2 = (ND & (ND - 1)) == 0;
    lane = opaque(lane);   // rows and offsets of the slots are re-derived here (an add each), not held across the level loop
    const int slot0 = (lane >> ilog2(ND)) * LS4 + (lane & (ND - 1));   // POW2: slot i lies (64 / ND) rows further down
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = lane + 64 * i, row = idx / ND;
        const uint32_t v = t.v[i];
        d4[POW2 ? slot0 + (64 / ND) * LS4 * i : row * LS4 + (idx - row * ND)] = make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
    }
}

__device__ __forceinline__ bool tile_inside(int x0, int n, int nd, int w, uint32_t pitch)
{
    return x0 >= 0 && x0 + n <= w && (x0 & ~3) + 4 * nd <= (int)pitch;
}

// NSEG: max pixels per lane, WMAX: largest window of this instantiation (fixes LDS strides and slot
// counts), ALIGNED: every level's base and pitch are multiples of 4, FULL: every active lane's segment has exactly
// NSEG pixels (the window splits evenly, e.g. 21 = 3 x 7) -- idle lanes then read a zeroed strip of LDS instead of
// shadowing lane 0, and no lane needs the per-pixel "past my segment" selects
template <int NSEG, int WMAX, bool ALIGNED, bool FULL>
__device__ __forceinline__ void pyrlk_wave(const LKArgs &a, const uint32_t bid, float *smem)
{
    using G = LKTiles<WMAX>;
    constexpr int NDP = G::NDP, NDJ = G::NDJ, NLP = G::NLP, NLJ = G::NLJ;
    constexpr int PS4 = G::PS4, JS4 = G::JS4, PS = G::PS, JS = G::JS, PFLOATS = G::PFLOATS, JFLOATS = G::JFLOATS;
    constexpr int TS = (WMAX + NSEG + 2) | 1;                                              // row stride of the resampled template tile
    constexpr int TBASE = PFLOATS + 64;                                                    // it follows the template tile and its slack; the
                                                                                           // search tile, committed after the pass, may overlap it
    constexpr int TEND = TBASE + (WMAX + 3) * TS;
    constexpr int TILE_FLOATS = (JFLOATS + 64 > TEND) ? JFLOATS + 64 : TEND;
    constexpr int ZSTRIP = 3 * TS + PS + 32;                                               // FULL: a strip that stays zero, what idle lanes read
    constexpr int LDS_FLOATS = TILE_FLOATS + (FULL ? ZSTRIP : 0);

    const int lane = threadIdx.x;
    const bool lane_bit0 = lane & 1, lane_bit1 = lane & 2;
#ifdef VSA_LK_STAMPS
    const uint64_t st_t0 = __builtin_readcyclecounter();
    const uint64_t st_w0 = wall_clock64();
    uint32_t st_iters = 0, st_first_head = 0;
    uint64_t st_ph[4] = {0, 0, 0, 0};   // cycles in: level head + P commit, template pass, solve + J commit, iterations
#define LK_PHASE(i) do { const uint64_t now_ = __builtin_readcyclecounter(); st_ph[i] += now_ - st_mark; st_mark = now_; } while (0)
    uint64_t st_mark = st_t0;
#else
#define LK_PHASE(i) do { } while (0)
#endif
    const uint32_t n = min(*a.prev_count, a.capacity);
    if (bid == 0 && lane == 0) {
        *a.out_count = n;
        if (a.out_count_host) *a.out_count_host = n;
    }
    if (bid >= n) return;
    // features arrive in detector-cell raster order: consecutive features are image neighbours, so give
    // each XCD a contiguous run of them (their tiles overlap in that XCD's L2)
    const uint32_t pi = xcd_contiguous_item(bid, n);

    float px, py, ex, ey;
    bool alive = true;
    if (a.item == kPoint2f) {
        const float2 p = reinterpret_cast<const float2 *>(a.prev_pts)[pi];
        const float2 e = reinterpret_cast<const float2 *>(a.estimates)[pi];
        px = p.x; py = p.y; ex = e.x; ey = e.y;
    } else {
        const KeypointF *p = reinterpret_cast<const KeypointF *>(a.prev_pts) + pi;
        const KeypointF *e = reinterpret_cast<const KeypointF *>(a.estimates) + pi;
        px = p->x; py = p->y; ex = e->x; ey = e->y;
        alive = p->tracking_status != 0;
    }

    const int w0 = (int)a.prev.width[0], h0 = (int)a.prev.height[0];
    bool ok = alive && px >= 0.0f && py >= 0.0f && px <= (float)(w0 - 1) && py <= (float)(h0 - 1);

    const int win = a.win, half = a.half;
    const int PW = win + 3, JW = win + 1 + 2 * LK_MARGIN;
    float *P = smem;                          // template tile; dead once I/Ix/Iy are in registers
    float *J = smem;                          // search tile (aliases P)
    float *T = smem + TBASE;                  // resampled template rows (written while P is being read; dead before J is committed)

    // lane -> (window row, segment).  Idle lanes shadow lane 0 with cnt = 0: every lane runs the same
    // branch-free code on valid LDS data and contributes exact zeros.
    int row = lane / a.segs;
    int x0 = (lane - row * a.segs) * a.seglen;
    int cnt = min(a.seglen, win - x0);
    const bool idle = lane >= a.lanes || cnt < 0;
    if (idle) { row = 0; x0 = 0; cnt = 0; }
    // FULL: an idle lane resamples and differentiates zeros (its template reads and its T row live in the zero strip),
    // so its I, Ix, Iy are exact zeros without any select; its search-tile reads stay on real pixels (finite x 0)
    // Where a lane's segment starts in each tile.  Derived where they are used, once per level, from a copy of the row number
    // the compiler cannot see through -- held across the level loop they are three of the 104 registers the kernel is capped at.
    auto lane_base_p = [&]() { return (FULL && idle) ? TILE_FLOATS - PS : opaque(row) * PS + x0; };   // reads start at TILE_FLOATS + (p_x0 & 3)
    auto lane_base_j = [&]() { return opaque(row) * JS + x0; };
    auto lane_base_t = [&]() { return (FULL && idle) ? TILE_FLOATS + PS + NSEG + 8 - TBASE : opaque(row) * TS + x0; };   // T rows behind the strip's template part
    static_assert(PS + NSEG + 8 + 2 * TS + NSEG + 2 <= ZSTRIP, "zero strip holds an idle lane's template and T accesses");

    // the top level's template tile first: nothing hides its loads but the set-up below
    const int top = (int)a.prev.levels - 1;
    const float s_top = pow2_recip(top);
    TileRegs<NLP> tp;
    TileRegs<NLJ> tj;
    int p_x0 = 0;
    if (ok) {
        p_x0 = (int)floorf(px * s_top) - half - 1;
        const int p_y0 = (int)floorf(py * s_top) - half - 1;
        const int tw = (int)a.prev.width[top];
        tile_issue<NLP, NDP>(tp, lane, a.prev.ptr[top], a.prev.pitch[top], tw, (int)a.prev.height[top], p_x0, p_y0,
                   ALIGNED && tile_inside(p_x0, PW, NDP, tw, a.prev.pitch[top]));
    }

    // LDS only ever holds converted pixels, resampled template values or the zeros written here (the slack a lane may
    // over-read past its segment at the end of either tile region, T entries nobody computes), so idle pixels
    // multiply finite values by exact zeros
    for (int i = PFLOATS + lane; i < LDS_FLOATS; i += 64) smem[i] = 0.0f;

    float gx = (ex - px) * s_top, gy = (ey - py) * s_top;
    float vx = 0.0f, vy = 0.0f;
    // wave-uniform and needed once per level: held in a scalar register, not in one of the 104 vector registers
    // (an explicit move: the compiler drops a readfirstlane builtin of a value it already knows to be uniform)
    float inv_area;
    {
        const float v = 1.0f / (float)(2 * win * win);
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(inv_area) : "v"(v));
    }
    const float fw = (float)win;

    float Iw[NSEG], Ix[NSEG], Iy[NSEG];
    float err = 0.0f;

    for (int L = top; L >= 0 && ok; --L) {
        const int lw = (int)a.prev.width[L], lh = (int)a.prev.height[L];
        const uint8_t *jp = a.curr.ptr[L];
        const uint32_t jpitch = a.curr.pitch[L];
        const float s = pow2_recip(L);
        const float ux = px * s, uy = py * s;
        const float ax = ux - floorf(ux), ay = uy - floorf(uy);
        vx = 0.0f; vy = 0.0f;

        // Search tile around the prediction: its loads fly while the template is resampled.  The compiler's wait counts are
        // conservative across this loop's branches: it waits for ALL outstanding loads before it commits a tile.  So the
        // template tile's registers are claimed here, before the search tile's loads are issued (their loads were issued a
        // level ago and have long landed: this wait is free, and the commit below then needs none), and the next level's
        // template loads are issued only after the search tile is committed.  Before, each commit found younger loads in
        // flight and exposed their whole memory latency at every level (phase stamps: 2.2 k + 1.3 k of a level's 9.3 k cycles).
#pragma unroll
        for (int i = 0; i < NLP; ++i) asm volatile("" : "+v"(tp.v[i]));
        int jx0 = (int)floorf(ux + gx) - half - LK_MARGIN;
        int jy0 = (int)floorf(uy + gy) - half - LK_MARGIN;
        tile_issue<NLJ, NDJ>(tj, lane, jp, jpitch, lw, lh, jx0, jy0, ALIGNED && tile_inside(jx0, JW, NDJ, lw, jpitch));

        __syncthreads();   // previous level's search tile is dead
        tile_commit<NLP, NDP, PS4>(tp, lane, P);
        __syncthreads();
        LK_PHASE(0);
#ifdef VSA_LK_STAMPS
        if (L == top) st_first_head = (uint32_t)(st_mark - st_t0);   // wave start .. top level's template tile in LDS
#endif

        // Template pass.  T(R, C), R, C = 0 .. win+1, is the template resampled at the feature's sub-pixel offset on the window
        // plus a one-pixel ring: v(R, c) = lerp_y(P(R, c), P(R+1, c)), T(R, C) = lerp_x(v(R, C), v(R, C+1)).  A lane needs three
        // rows of it (R = row, row+1, row+2) at columns x0 .. x0+NSEG+1, but every T value is needed by up to six lanes, so
        //   phase 1: a lane resamples only its centre row R = row+1 (kept in registers, and stored to the T tile in LDS);
        //            lanes 0 .. 2(win+2)-1 add one value each of the two rows no lane has as its centre (R = 0, win+1);
        //   phase 2: it reads the rows above and below from the T tile and differentiates (Scharr/32, separable, 3/32 and 10/32
        //            folded in): sm, df = column terms; pixel k: Ix = sm[k+2] - sm[k], Iy from df[k..k+2], I = T(row+1, k+1).
        // Neighbouring segments store the columns they share with identical bits; columns past the window hold finite values
        // nobody uses (pixels k >= cnt are zeroed below).
        float sxx = 0.0f, sxy = 0.0f, syy = 0.0f;
        {
            const float *q = P + lane_base_p() + (p_x0 & 3) + PS;   // P(row + 1 + rr, x0 + c) = q[rr*PS + c]
            // (the T tile's offset goes into the base register: folded into the instructions' 8-bit dword offsets it does not fit,
            // and every ds_read2 then gets an address add of its own)
            const float *Tl = smem + opaque(TBASE + lane_base_t());
            float tc[NSEG + 2];
            {
                float vp = fmaf(ay, q[PS] - q[0], q[0]);
#pragma unroll
                for (int c = 1; c < NSEG + 3; ++c) {
                    const float v = fmaf(ay, q[PS + c] - q[c], q[c]);
                    tc[c - 1] = fmaf(ax, v - vp, vp);
                    vp = v;
                }
            }
            float *tw = const_cast<float *>(Tl) + TS;          // T(row + 1, x0 + j)
#pragma unroll
            for (int j = 0; j < NSEG + 2; ++j) tw[j] = tc[j];
            for (int e = opaque(lane); e < 2 * (win + 2); e += 64) {   // rows R = 0 and R = win+1
                const int far = e >= win + 2;
                const int C = e - (far ? win + 2 : 0), R = far ? win + 1 : 0;
                const float *pe = P + (p_x0 & 3) + R * PS + C;
                const float va = fmaf(ay, pe[PS] - pe[0], pe[0]), vb = fmaf(ay, pe[PS + 1] - pe[1], pe[1]);
                T[R * TS + C] = fmaf(ax, vb - va, va);
            }
            __syncthreads();
            const float *tu = Tl, *td = Tl + 2 * TS;                    // T(row, x0 + j), T(row + 2, x0 + j)
            float sm_p = 0.0f, sm_pp = 0.0f, df_p = 0.0f, df_pp = 0.0f;
#pragma unroll
            for (int j = 0; j < NSEG + 2; ++j) {
                const float up = tu[j], dn = td[j];
                const float sm = fmaf(0.3125f, tc[j], 0.09375f * (up + dn));
                const float df = dn - up;
                if (j >= 2) {
                    const int k = j - 2;
                    float ix = sm - sm_pp;
                    float iy = fmaf(0.3125f, df_p, 0.09375f * (df_pp + df));
                    if (!FULL && k >= cnt) { ix = 0.0f; iy = 0.0f; }   // selects, not a branch
                    Iw[k] = tc[k + 1];
                    Ix[k] = ix;
                    Iy[k] = iy;
                    sxx = fmaf(ix, ix, sxx);
                    sxy = fmaf(ix, iy, sxy);
                    syy = fmaf(iy, iy, syy);
                }
                sm_pp = sm_p; sm_p = sm; df_pp = df_p; df_p = df;
            }
        }
        float Gxx, Gxy, Gyy;
        wave_sum3(sxx, sxy, syy, lane_bit0, lane_bit1, Gxx, Gxy, Gyy);
        LK_PHASE(1);

        // next level's template loads fly while this level iterates (issued once the search tile is committed, see above)
        auto issue_next_template = [&]() {
            if (L > 0) {
                const float sn = pow2_recip(L - 1);
                p_x0 = (int)floorf(px * sn) - half - 1;
                const int p_y0 = (int)floorf(py * sn) - half - 1;
                const int nw = (int)a.prev.width[L - 1];
                tile_issue<NLP, NDP>(tp, lane, a.prev.ptr[L - 1], a.prev.pitch[L - 1], nw, (int)a.prev.height[L - 1], p_x0, p_y0,
                           ALIGNED && tile_inside(p_x0, PW, NDP, nw, a.prev.pitch[L - 1]));
            }
        };

        const float det = Gxx * Gyy - Gxy * Gxy;
        const float dxy = Gxx - Gyy;
        const float disc = sqrtf(dxy * dxy + 4.0f * (Gxy * Gxy));
        const float min_eig = ((Gyy + Gxx) - disc) * inv_area;
        if (!(min_eig >= 1e-4f) || !(det >= FLT_EPSILON)) {
            if (L == 0) ok = false;
            gx = 2.0f * gx; gy = 2.0f * gy;
            issue_next_template();
            continue;
        }
        const float inv_det = 1.0f / det;
        const float flw = (float)lw, flh = (float)lh;

        __syncthreads();   // every lane is done with P
        tile_commit<NLJ, NDJ, JS4>(tj, lane, J);
        __syncthreads();
        issue_next_template();
        LK_PHASE(2);

        float ja[NSEG + 1], jd[NSEG + 1];                     // search row jy + row at columns jx + x0 .. and row jy + row + 1 MINUS it
        int cached_jx = INT_MIN, cached_jy = INT_MIN;         // ... for this integer position (none yet at this level)
        for (uint32_t it = 0; it < a.max_it; ++it) {
            const float cx = (ux + gx) + vx, cy = (uy + gy) + vy;
            if (!(cx > -fw && cy > -fw && cx < flw + fw && cy < flh + fw)) { ok = false; break; }
            const float jfx = floorf(cx), jfy = floorf(cy);
            const float bx = cx - jfx, by = cy - jfy;
            // convert in the vector ALU, then ONE move to a scalar register per coordinate (left alone the compiler moves
            // the float over first and converts it there, a second v_readfirstlane each)
            int jxi = (int)jfx, jyi = (int)jfy;
            asm volatile("" : "+v"(jxi), "+v"(jyi));
            const int jx = __builtin_amdgcn_readfirstlane(jxi) - half, jy = __builtin_amdgcn_readfirstlane(jyi) - half;
            int ox = jx - jx0, oy = jy - jy0;
            if (ox < 0 || oy < 0 || ox > 2 * LK_MARGIN || oy > 2 * LK_MARGIN) {
                // walked out of the staged tile: restage around the current position (rare)
                jx0 = jx - LK_MARGIN; jy0 = jy - LK_MARGIN; ox = LK_MARGIN; oy = LK_MARGIN;
                tile_issue<NLJ, NDJ>(tj, lane, jp, jpitch, lw, lh, jx0, jy0, ALIGNED && tile_inside(jx0, JW, NDJ, lw, jpitch));
                __syncthreads();
                tile_commit<NLJ, NDJ, JS4>(tj, lane, J);
                __syncthreads();
            }
#ifdef VSA_LK_STAMPS
            ++st_iters;
#endif
            // The two search-tile rows a lane samples depend only on the INTEGER part of the position.  After a level's first
            // iteration the update is usually a fraction of a pixel, so the rows are kept in registers and re-read only when
            // the integer position moves: the iteration loop is bound by the LDS pipe (16 dword reads per lane with 2-way
            // bank conflicts; profiles/r02_kernel_experiments.txt), and most iterations now read nothing.  What is kept of the
            // lower row is its difference from the upper one (exact: both are integers below 256), the only form the vertical
            // lerp uses: eight subtractions less in every iteration that reads nothing.
            if (jx != cached_jx || jy != cached_jy) {   // wave-uniform
                const float *jt = J + lane_base_j() + (oy * JS + (jx0 & 3) + ox);
#pragma unroll
                for (int k = 0; k < NSEG + 1; ++k) { ja[k] = jt[k]; jd[k] = jt[JS + k] - ja[k]; }
                cached_jx = jx; cached_jy = jy;
            }
            float b1 = 0.0f, b2 = 0.0f;
            {
                float cp = fmaf(by, jd[0], ja[0]);
#pragma unroll
                for (int k = 0; k < NSEG; ++k) {   // pixels past cnt have Ix = Iy = 0 and add exact zeros
                    const float cc = fmaf(by, jd[k + 1], ja[k + 1]);
                    const float jv = fmaf(bx, cc - cp, cp);
                    const float d = Iw[k] - jv;
                    b1 = fmaf(d, Ix[k], b1);
                    b2 = fmaf(d, Iy[k], b2);
                    cp = cc;
                }
            }
            wave_sum2(b1, b2, lane_bit0, b1, b2);
            const float etx = (Gyy * b1 - Gxy * b2) * inv_det;
            const float ety = (Gxx * b2 - Gxy * b1) * inv_det;
            vx += etx; vy += ety;
            if (a.use_eps && fmaf(etx, etx, ety * ety) < a.eps2) break;
        }
        LK_PHASE(3);
        if (L > 0) { gx = 2.0f * (gx + vx); gy = 2.0f * (gy + vy); }
        else if (a.calc_error && ok) {
            // calcError: mean |I - J| over the window at the final position of level 0 (the position an iteration
            // would sample next); the template is still in registers, the search tile in LDS
            const float cx = (ux + gx) + vx, cy = (uy + gy) + vy;
            if (cx > -fw && cy > -fw && cx < flw + fw && cy < flh + fw) {
                const float jfx = floorf(cx), jfy = floorf(cy);
                const float bx = cx - jfx, by = cy - jfy;
                const int jx = __builtin_amdgcn_readfirstlane((int)jfx) - half, jy = __builtin_amdgcn_readfirstlane((int)jfy) - half;
                int ox = jx - jx0, oy = jy - jy0;
                if (ox < 0 || oy < 0 || ox > 2 * LK_MARGIN || oy > 2 * LK_MARGIN) {
                    jx0 = jx - LK_MARGIN; jy0 = jy - LK_MARGIN; ox = LK_MARGIN; oy = LK_MARGIN;
                    tile_issue<NLJ, NDJ>(tj, lane, jp, jpitch, lw, lh, jx0, jy0, ALIGNED && tile_inside(jx0, JW, NDJ, lw, jpitch));
                    __syncthreads();
                    tile_commit<NLJ, NDJ, JS4>(tj, lane, J);
                    __syncthreads();
                }
                const float *jt = J + lane_base_j() + (oy * JS + (jx0 & 3) + ox);
                float cp = fmaf(by, jt[JS] - jt[0], jt[0]);
                float e = 0.0f;
#pragma unroll
                for (int k = 0; k < NSEG; ++k) {
                    const float cc = fmaf(by, jt[JS + k + 1] - jt[k + 1], jt[k + 1]);
                    const float jv = fmaf(bx, cc - cp, cp);
                    e = e + (k < cnt ? fabsf(Iw[k] - jv) : 0.0f);
                    cp = cc;
                }
                int area = win * win;
                asm volatile("" : "+s"(area));   // keeps the reciprocal here: hoisted to the kernel's head it holds a register throughout
                err = wave_sum(e) * (1.0f / (float)area);
            }
        }
    }

    float rx = -1.0f, ry = -1.0f;
    if (ok) {
        rx = px + (gx + vx); ry = py + (gy + vy);
        ok = rx >= 0.0f && ry >= 0.0f && rx <= (float)(w0 - 1) && ry <= (float)(h0 - 1);
    }
    if (lane == 0) {
        if (a.item == kPoint2f) {
            reinterpret_cast<float2 *>(a.out)[pi] = ok ? make_float2(rx, ry) : make_float2(-1.0f, -1.0f);
        } else {
            // field by field: a struct copy is lowered through a private-memory temporary, which leaves the kernel
            // with a 16-byte scratch allocation per lane although no instruction ever touches it
            const KeypointF *src = reinterpret_cast<const KeypointF *>(a.prev_pts) + pi;
            KeypointF *dst = reinterpret_cast<KeypointF *>(a.out) + pi;
            const float strength = src->strength, scale = src->scale, orientation = src->orientation;
            // a lost item keeps its estimate: read again here (through a pointer the compiler cannot match with the first
            // read) rather than held in two registers from the kernel's head to its last instructions
            const KeypointF *est = reinterpret_cast<const KeypointF *>(a.estimates);
            asm volatile("" : "+s"(est));
            const float lost_x = est[pi].x, lost_y = est[pi].y;
            dst->x = ok ? rx : lost_x;
            dst->y = ok ? ry : lost_y;
            dst->strength = strength;
            dst->scale = scale;
            dst->orientation = orientation;
            dst->tracking_status = ok ? 255 : 0;
            dst->error = ok ? err : 0.0f;
        }
    }
#ifdef VSA_LK_STAMPS
    if (lane == 0 && bid < 8192) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        uint64_t *o = g_lk_stamps + 8 * bid;
        o[0] = st_t0; o[1] = __builtin_readcyclecounter(); o[2] = ((uint64_t)xcc << 32) | hw;
        o[3] = ((wall_clock64() - st_w0) << 32) | (st_iters & 0xffu) | (min(st_first_head, 0xffffffu) << 8);
        o[4] = st_ph[0]; o[5] = st_ph[1]; o[6] = st_ph[2]; o[7] = st_ph[3];
    }
#endif
}

template <int NSEG, int WMAX, bool ALIGNED, bool FULL>
__global__ __launch_bounds__(64) void k_pyrlk(LKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    pyrlk_wave<NSEG, WMAX, ALIGNED, FULL>(a, blockIdx.x, smem);
}
// The 21 x 21 bucket (the feature tracker's default window) with at most 104 registers: four of its waves then leave 96 of a
// SIMD's 512 registers to other camera streams' detector (92) and merge (48) waves, which otherwise wait for PyrLK waves to
// drain (-9 % on the 32-stream bench at 109 registers).  On gfx950 the attribute counts the unified VGPR + AGPR file in
// halves: 52 caps the kernel at 104.
// ... and with no fewer: at 96 registers or less a fifth wave fits on a SIMD, and a GPU full of PyrLK waves keeps the other
// camera streams' short kernels waiting (-9 % on the bench at 92 registers, -15 % at 96 with spills).  Naming v103 in an
// empty asm statement makes the allocation 104 whatever the compiler needs.
#define VSA_LK_PIN_104() asm volatile("" ::: "v103")
template <int NSEG, int WMAX, bool ALIGNED, bool FULL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(VSA_LK_VGPR_HALF))) void k_pyrlk_104(LKArgs a)
{
    VSA_LK_PIN_104();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    pyrlk_wave<NSEG, WMAX, ALIGNED, FULL>(a, blockIdx.x, smem);
}

// PyrLK + the detector's cell pass in one launch (launch_pyrlk_with_cells): workgroups [0, a.capacity) each track a feature,
// the rest each own a rectangle of detector cells (harris_cells.hpp, fast_cells.hpp).  PyrLK's waves come first -- they are the
// long ones, and a launch of them alone ends with compute units idling behind its slowest waves; the cell-pass waves fill that tail.
// DET: 0 = Harris, 9 .. 12 = FAST-N.
template <int NSEG, int WMAX, bool FULL, int DET>
__device__ __forceinline__ void pyrlk_cells_wave(const LKArgs &a, const CornerArgs &c)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (blockIdx.x < a.capacity) pyrlk_wave<NSEG, WMAX, true, FULL>(a, blockIdx.x, smem);
    else {
        // the cell-pass waves start in the launch's tail, beside PyrLK's slowest waves: with issue priority over them the
        // launch alone ends 2.4 us earlier (40.5 against 42.9 us); the other order (cell pass first) changes nothing
        __builtin_amdgcn_s_setprio(3);
        if constexpr (DET == 0) harris_cells_wave(c, blockIdx.x - a.capacity, gridDim.x - a.capacity, reinterpret_cast<uint8_t *>(smem));
        else fast_cells_wave<DET>(c, blockIdx.x - a.capacity, gridDim.x - a.capacity, reinterpret_cast<uint8_t *>(smem));
    }
}
template <int NSEG, int WMAX, bool FULL, int DET>
__global__ __launch_bounds__(64) void k_pyrlk_cells(LKArgs a, CornerArgs c)
{
    pyrlk_cells_wave<NSEG, WMAX, FULL, DET>(a, c);
}
template <int NSEG, int WMAX, bool FULL, int DET>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(VSA_LK_VGPR_HALF))) void k_pyrlk_cells_104(LKArgs a, CornerArgs c)
{
    VSA_LK_PIN_104();
    pyrlk_cells_wave<NSEG, WMAX, FULL, DET>(a, c);
}

template <int NSEG, int WMAX, bool FULL, int DET>
void launch_cells_kernel(const LKArgs &a, const CornerArgs &c, dim3 grid, size_t lds, hipStream_t stream)
{
    if constexpr (NSEG == 7) hipLaunchKernelGGL((k_pyrlk_cells_104<NSEG, WMAX, FULL, DET>), grid, dim3(64), lds, stream, a, c);
    else hipLaunchKernelGGL((k_pyrlk_cells<NSEG, WMAX, FULL, DET>), grid, dim3(64), lds, stream, a, c);
}
template <int NSEG, int WMAX, bool FULL>
void launch_cells_detector(const LKArgs &a, const CornerArgs &c, int det, dim3 grid, size_t lds, hipStream_t stream)
{
    switch (det) {
    case 0: launch_cells_kernel<NSEG, WMAX, FULL, 0>(a, c, grid, lds, stream); break;
    case 9: launch_cells_kernel<NSEG, WMAX, FULL, 9>(a, c, grid, lds, stream); break;
    case 10: launch_cells_kernel<NSEG, WMAX, FULL, 10>(a, c, grid, lds, stream); break;
    case 11: launch_cells_kernel<NSEG, WMAX, FULL, 11>(a, c, grid, lds, stream); break;
    default: launch_cells_kernel<NSEG, WMAX, FULL, 12>(a, c, grid, lds, stream); break;
    }
}

template <int NSEG, int WMAX>
hipError_t launch_bucket(const LKArgs &a, bool aligned, uint32_t capacity, hipStream_t stream, const CornerArgs *cells = nullptr,
                         uint32_t cell_blocks = 0, size_t cell_lds = 0, int det = 0)
{
    using G = LKTiles<WMAX>;
    constexpr int PS = G::PS, PFLOATS = G::PFLOATS, JFLOATS = G::JFLOATS;
    constexpr int TS = (WMAX + NSEG + 2) | 1;
    constexpr int TEND = PFLOATS + 64 + (WMAX + 3) * TS;   // template tile + zeroed slack + T tile; the search tile overlaps T
    constexpr int TILE_FLOATS = (JFLOATS + 64 > TEND) ? JFLOATS + 64 : TEND;
    // every active lane has a whole segment of NSEG pixels: the specialisation without per-pixel selects (+ a zero strip in LDS)
    const bool full = a.seglen == NSEG && a.segs * a.seglen == a.win;
    const size_t lds = (size_t)(TILE_FLOATS + (full ? 3 * TS + PS + 32 : 0)) * sizeof(float);
    const dim3 block(64);
    if (cells) {   // aligned pyramids only (pyrlk_cells_fusable)
        const dim3 grid(capacity + cell_blocks);
        const size_t both = lds > cell_lds ? lds : cell_lds;
        if (full) launch_cells_detector<NSEG, WMAX, true>(a, *cells, det, grid, both, stream);
        else launch_cells_detector<NSEG, WMAX, false>(a, *cells, det, grid, both, stream);
        return hipGetLastError();
    }
    const dim3 grid(capacity);
    if constexpr (NSEG == 7) {
        if (aligned && full) hipLaunchKernelGGL((k_pyrlk_104<NSEG, WMAX, true, true>), grid, block, lds, stream, a);
        else if (aligned) hipLaunchKernelGGL((k_pyrlk_104<NSEG, WMAX, true, false>), grid, block, lds, stream, a);
        else if (full) hipLaunchKernelGGL((k_pyrlk<NSEG, WMAX, false, true>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((k_pyrlk<NSEG, WMAX, false, false>), grid, block, lds, stream, a);
    } else {
        if (aligned && full) hipLaunchKernelGGL((k_pyrlk<NSEG, WMAX, true, true>), grid, block, lds, stream, a);
        else if (aligned) hipLaunchKernelGGL((k_pyrlk<NSEG, WMAX, true, false>), grid, block, lds, stream, a);
        else if (full) hipLaunchKernelGGL((k_pyrlk<NSEG, WMAX, false, true>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((k_pyrlk<NSEG, WMAX, false, false>), grid, block, lds, stream, a);
    }
    return hipGetLastError();
}

bool pyramids_aligned(const PyrLKParams &p)
{
    bool aligned = true;
    for (uint32_t l = 0; l < p.prev.levels; ++l)
        aligned = aligned && ((uintptr_t)p.prev.ptr[l] % 4u == 0) && (p.prev.pitch[l] % 4u == 0) &&
                  ((uintptr_t)p.curr.ptr[l] % 4u == 0) && (p.curr.pitch[l] % 4u == 0);
    return aligned;
}

hipError_t launch_lk(const PyrLKParams &p, hipStream_t stream, const CornerArgs *cells, uint32_t cell_blocks, size_t cell_lds, int det)
{
    const bool aligned = pyramids_aligned(p);
    LKArgs a{};
    a.prev = p.prev; a.curr = p.curr; a.item = p.item;
    a.prev_pts = p.prev_pts; a.prev_count = p.prev_count; a.estimates = p.estimates;
    a.out = p.out; a.out_count = p.out_count; a.out_count_host = p.out_count_host; a.capacity = p.capacity;
    a.use_eps = p.use_epsilon; a.max_it = p.max_iterations; a.eps2 = p.epsilon * p.epsilon;
    a.calc_error = p.calc_error ? 1 : 0;
    const int win = (int)p.window;
    a.win = win; a.half = win / 2;
    // lane map shared with the oracle (lk_map): segs x win lanes of seglen pixels
    int segs = 64 / win;
    if (segs > win) segs = win;
    if (segs < 1) segs = 1;
    a.seglen = (win + segs - 1) / segs;
    a.segs = (win + a.seglen - 1) / a.seglen;
    a.lanes = a.segs * win;

    if (win <= 8) return launch_bucket<1, 8>(a, aligned, p.capacity, stream, cells, cell_blocks, cell_lds, det);   // one pixel per lane
    if (win <= 10) return launch_bucket<2, 10>(a, aligned, p.capacity, stream, cells, cell_blocks, cell_lds, det);
    if (win <= 16) return launch_bucket<4, 16>(a, aligned, p.capacity, stream, cells, cell_blocks, cell_lds, det);
    if (win <= 21) return launch_bucket<7, 21>(a, aligned, p.capacity, stream, cells, cell_blocks, cell_lds, det);
    return launch_bucket<16, 32>(a, aligned, p.capacity, stream, cells, cell_blocks, cell_lds, det);
}

}  // namespace

hipError_t launch_pyrlk(const PyrLKParams &p, hipStream_t stream)
{
    if (p.capacity == 0) return hipSuccess;
    return launch_lk(p, stream, nullptr, 0, 0, 0);
}

bool pyrlk_cells_fusable(const PyrLKParams &lk, const CornerTrackParams &ct)
{
    const bool det_ok = ct.detector == kHarris || (ct.detector == kFast && ct.fast_n >= 9u && ct.fast_n <= 12u);
    return det_ok && ct.cell_size != 0u && ct.image.width <= 65535u && ct.image.height <= 65535u &&
           lk.capacity != 0u && pyramids_aligned(lk);
}

hipError_t launch_pyrlk_with_cells(const PyrLKParams &lk, const CornerTrackParams &ct, hipStream_t stream)
{
    if (!pyrlk_cells_fusable(lk, ct)) return hipErrorInvalidValue;
    CornerArgs c{};
    uint32_t blocks = 0;
    size_t lds = 0;
    if (ct.detector == kHarris) harris_cells_setup(ct, c, blocks, lds, HS_BAND_FUSED);
    else fast_cells_setup(ct, c, blocks, lds, FS_BAND_FUSED);
    return launch_lk(lk, stream, &c, blocks, lds, ct.detector == kHarris ? 0 : (int)ct.fast_n);
}

}  // namespace vsa

#ifdef VSA_LK_STAMPS
extern "C" int visualslam_amd_debug_lk_stamps(uint64_t *host, size_t n_u64)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vsa::g_lk_stamps), n_u64 * sizeof(uint64_t));
}
#endif
