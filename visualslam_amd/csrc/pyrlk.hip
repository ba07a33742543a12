// pyrlk.hip -- sparse pyramidal Lucas-Kanade optical flow, one wavefront per feature (fast path).
//
// Reference call sites replaced: vxOpticalFlowPyrLKNode / nvxcuOpticalFlowPyrLK
// (demos/feature_tracker/feature_tracker.cpp:402-412, demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:297-301,
// samples/object_tracker_nvxcu/object_tracker_keypoint_nvxcu.cpp:1791-1828).  Semantics and the exact
// floating-point operation order: oracle/vxoracle.c (lk_track_one) -- results are bit-identical.
//
// Design (MI355X, gfx950):
//   * a 64-lane wave owns one feature for all pyramid levels, coarse to fine; window pixels are
//     dealt to lanes as (row, segment of `seglen` consecutive pixels), so a lane streams along two
//     LDS rows and every column it reads serves two neighbouring pixels;
//   * tiles are fetched as ALIGNED DWORDS (4 pixels per lane per load, all loads of a tile issued
//     back to back into registers), converted once and written to LDS as one ds_write_b128 each;
//     tiles that poke out of the image are patched per byte with the row's edge pixels (replicate);
//   * latency hiding inside the wave: the search tile's loads are in flight while the template is
//     resampled, and the next level's template loads are in flight while this level iterates;
//   * Scharr gradients are never stored: each lane derives them for its segment from four LDS rows
//     (exact small integers, so the separable evaluation order is free), then resamples I, Ix, Iy
//     bilinearly into REGISTERS where they stay for all iterations;
//   * packed f32 math (v_pk_fma_f32) carries the (top,bottom) row pair and the (Ix,Iy) pair;
//   * sums are combined by DPP row reductions + 4 readlanes, in the butterfly order the oracle mirrors;
//   * the 2x2 normal equations are solved in registers, wave-uniformly.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "kernels.hpp"

namespace vsa {
namespace {

constexpr int LK_MARGIN = 4;  // search tile margin M (pixels) on each side of the window

// A pair of floats handled as two SCALAR f32 operations: on gfx950 the packed-f32 VALU forms (v_pk_fma_f32,
// v_pk_add_f32, v_pk_mul_f32) issue far slower than two plain v_fma_f32/v_add_f32 (MI355X_MICROARCH.md,
// constants table), so the pair is kept only as notation and the file is built with -fno-slp-vectorize.
struct v2f {
    float x, y;
};
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return { a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return { a.x * b.x, a.y * b.y }; }

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
    int win, half, segs, seglen, lanes, lds_floats;
};

struct KeypointF {
    float x, y, strength, scale, orientation;
    int32_t tracking_status;
    float error;
};

#ifdef VSA_LK_STAMPS   // development instrumentation: per-wave start/end cycle stamps and placement
__device__ uint64_t g_lk_stamps[4 * 8192];
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_bcast(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// Pairwise butterfly (offsets 1,2,4,8,16,32); every lane ends with the same bits.  Within a row of
// 16 the partner choice of steps 4 and 8 (mirrors) is immaterial because partial sums are already
// uniform inside each group; rows are combined as (r0+r1)+(r2+r3).  Needs all 64 lanes active.
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    const float r0 = lane_bcast(v, 0), r1 = lane_bcast(v, 16), r2 = lane_bcast(v, 32), r3 = lane_bcast(v, 48);
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return { fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y) }; }
__device__ __forceinline__ v2f splat(float x) { return { x, x }; }

// ---- dword tile staging ----------------------------------------------------------------------
// A tile of `n` rows x `n` columns with origin (x0,y0) is fetched as n x nd aligned dwords starting
// at column xa = floor4(x0); LDS image: row r, dword d -> floats [4*(r*nd+d) .. +3], so tile
// element (r,c) lives at r*4*nd + (x0-xa) + c.
template <int NL>
struct TileLoad {
    uint32_t v[NL];
    uint32_t edge[NL];   // row[0] | row[w-1] << 8, only when the tile is not inside the image
};

template <int NL>
__device__ __forceinline__ void tile_issue(TileLoad<NL> &t, const uint8_t *img, uint32_t pitch, int w, int h,
                                           int x0, int y0, int n, int nd, bool inside, int lane)
{
    const int xa = x0 & ~3;
    const float inv_nd = 1.0f / (float)nd;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        // lanes past the end repeat the last dword (same LDS slot, same data): no divergence
        const int idx = min(lane + 64 * i, n * nd - 1);
        const int r = (int)(((float)idx + 0.5f) * inv_nd);
        const int d = idx - r * nd;
        const int gy = min(max(y0 + r, 0), h - 1);
        const uint8_t *rowp = img + (uint32_t)gy * pitch;
        const int X0 = xa + 4 * d;
        t.edge[i] = 0;
        if (inside) {   // wave-uniform
            t.v[i] = *reinterpret_cast<const uint32_t *>(rowp + X0);
        } else {
            const int Xl = min(max(X0, 0), (w - 1) & ~3);
            t.v[i] = *reinterpret_cast<const uint32_t *>(rowp + Xl);
            t.edge[i] = (uint32_t)rowp[0] | ((uint32_t)rowp[w - 1] << 8);
        }
    }
}

template <int NL>
__device__ __forceinline__ void tile_commit(const TileLoad<NL> &t, float *dst, int w, int x0, int n, int nd,
                                            bool inside, int lane)
{
    const int xa = x0 & ~3;
    const float inv_nd = 1.0f / (float)nd;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = min(lane + 64 * i, n * nd - 1);
        const uint32_t v = t.v[i];
        uint32_t b0 = v & 0xffu, b1 = (v >> 8) & 0xffu, b2 = (v >> 16) & 0xffu, b3 = v >> 24;
        if (!inside) {   // wave-uniform
            const int r = (int)(((float)idx + 0.5f) * inv_nd);
            const int X0 = xa + 4 * (idx - r * nd);
            const uint32_t l = t.edge[i] & 0xffu, rr = t.edge[i] >> 8;
            b0 = (X0 + 0 < 0) ? l : ((X0 + 0 >= w) ? rr : b0);
            b1 = (X0 + 1 < 0) ? l : ((X0 + 1 >= w) ? rr : b1);
            b2 = (X0 + 2 < 0) ? l : ((X0 + 2 >= w) ? rr : b2);
            b3 = (X0 + 3 < 0) ? l : ((X0 + 3 >= w) ? rr : b3);
        }
        *reinterpret_cast<float4 *>(dst + 4 * idx) = make_float4((float)b0, (float)b1, (float)b2, (float)b3);
    }
}

__device__ __forceinline__ bool tile_inside(int x0, int n, int nd, int w, uint32_t pitch)
{
    return x0 >= 0 && x0 + n <= w && (x0 & ~3) + 4 * nd <= (int)pitch;
}

// NSEG: max pixels per lane, WMAX: max window of this instantiation (bounds the unrolled staging)
template <int NSEG, int WMAX>
__global__ __launch_bounds__(64) void k_pyrlk(LKArgs a)
{
    constexpr int NLP = ((WMAX + 3) * ((WMAX + 9) / 4) + 63) / 64;
    constexpr int NLJ = ((WMAX + 1 + 2 * LK_MARGIN) * ((WMAX + 7 + 2 * LK_MARGIN) / 4) + 63) / 64;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
#ifdef VSA_LK_STAMPS
    const uint64_t st_t0 = __builtin_readcyclecounter();
    const uint64_t st_w0 = wall_clock64();
    uint32_t st_iters = 0;
#endif
    const uint32_t n = min(*a.prev_count, a.capacity);
    if (blockIdx.x == 0 && lane == 0) {
        *a.out_count = n;
        if (a.out_count_host) *a.out_count_host = n;
    }
    if (blockIdx.x >= n) return;
    // features arrive in detector-cell raster order: consecutive features are image neighbours, so give
    // each XCD a contiguous run of them (their tiles overlap in that XCD's L2)
    const uint32_t pi = xcd_contiguous_item(blockIdx.x, n);

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
    const int NDP = (PW + 6) >> 2, NDJ = (JW + 6) >> 2;
    const int PS = 4 * NDP, JS = 4 * NDJ;     // LDS row strides (floats)
    float *P = smem;                          // template tile; dead once I/Ix/Iy are in registers
    float *J = smem;                          // search tile (aliases P)

    // lane -> (window row, segment).  Idle lanes shadow lane 0 with cnt = 0: every lane runs the same
    // branch-free code on valid LDS data and contributes exact zeros.
    int row = lane / a.segs;
    int x0 = (lane - row * a.segs) * a.seglen;
    int cnt = min(a.seglen, win - x0);
    if (lane >= a.lanes || cnt < 0) { row = 0; x0 = 0; cnt = 0; }

    // LDS only ever holds converted pixels or these zeros, so over-reads past a segment are finite
    for (int i = lane; i < a.lds_floats; i += 64) smem[i] = 0.0f;

    const int top = (int)a.prev.levels - 1;
    const float s_top = 1.0f / (float)(1u << top);
    float gx = (ex - px) * s_top, gy = (ey - py) * s_top;
    float vx = 0.0f, vy = 0.0f;
    const float inv_area = 1.0f / (float)(2 * win * win);
    const float fw = (float)win;

    float Iw[NSEG];
    v2f Ig[NSEG];   // (Ix, Iy)

    TileLoad<NLP> tp;
    TileLoad<NLJ> tj;
    int p_x0 = 0;
    bool p_in = false;
    if (ok) {   // template tile of the top level
        p_x0 = (int)floorf(px * s_top) - half - 1;
        const int p_y0 = (int)floorf(py * s_top) - half - 1;
        p_in = tile_inside(p_x0, PW, NDP, (int)a.prev.width[top], a.prev.pitch[top]);
        tile_issue(tp, a.prev.ptr[top], a.prev.pitch[top], (int)a.prev.width[top], (int)a.prev.height[top],
                   p_x0, p_y0, PW, NDP, p_in, lane);
    }

    for (int L = top; L >= 0 && ok; --L) {
        const int lw = (int)a.prev.width[L], lh = (int)a.prev.height[L];
        const uint8_t *jp = a.curr.ptr[L];
        const uint32_t jpitch = a.curr.pitch[L];
        const float s = 1.0f / (float)(1u << L);
        const float ux = px * s, uy = py * s;
        const float ax = ux - floorf(ux), ay = uy - floorf(uy);
        vx = 0.0f; vy = 0.0f;

        // search tile around the prediction: its loads fly while the template is resampled
        int jx0 = (int)floorf(ux + gx) - half - LK_MARGIN;
        int jy0 = (int)floorf(uy + gy) - half - LK_MARGIN;
        bool j_in = tile_inside(jx0, JW, NDJ, lw, jpitch);
        tile_issue(tj, jp, jpitch, lw, lh, jx0, jy0, JW, NDJ, j_in, lane);

        __syncthreads();   // previous level's search tile is dead
        tile_commit(tp, P, lw, p_x0, PW, NDP, p_in, lane);
        __syncthreads();

        // template pass: Scharr on the fly from four LDS rows (exact integers), bilinear resample
        // of I, Ix, Iy into registers.  Streaming over the lane's P columns c = 0 .. cnt+2:
        // grid column g = c-2 (centre P column c-1) completes at c, window pixel k = c-3 right after.
        float sxx = 0.0f, sxy = 0.0f, syy = 0.0f;
        {
            const float *q = P + row * PS + (p_x0 & 3) + x0;   // P(row + rr, x0 + c) = q[rr*PS + c]
            v2f sm0 = splat(0.0f), sm1 = splat(0.0f), df0 = splat(0.0f), df1 = splat(0.0f), ctr1 = splat(0.0f);
            v2f Ip = splat(0.0f), GXp = splat(0.0f), GYp = splat(0.0f);
#pragma unroll
            for (int c = 0; c < NSEG + 3; ++c) {
                const float p0 = q[c], p1 = q[PS + c], p2 = q[2 * PS + c], p3 = q[3 * PS + c];
                const v2f lo = { p0, p1 }, mid = { p1, p2 }, hi = { p2, p3 };      // (top, bottom) grid rows
                const v2f sm = pk_fma(splat(3.0f), hi, pk_fma(splat(10.0f), mid, splat(3.0f) * lo));
                const v2f df = hi - lo;
                if (c >= 2) {
                    const v2f Ic = ctr1;
                    const v2f GXc = sm - sm0;
                    const v2f GYc = pk_fma(splat(3.0f), df, pk_fma(splat(10.0f), df1, splat(3.0f) * df0));
                    if (c >= 3) {
                        const int k = c - 3;
                        const v2f hi_ = pk_fma(splat(ax), Ic - Ip, Ip);          // (top, bot) horizontal lerps
                        const v2f hx = pk_fma(splat(ax), GXc - GXp, GXp);
                        const v2f hy = pk_fma(splat(ax), GYc - GYp, GYp);
                        Iw[k] = fmaf(ay, hi_.y - hi_.x, hi_.x);
                        const v2f tops = { hx.x, hy.x }, bots = { hx.y, hy.y };
                        v2f g = pk_fma(splat(ay), bots - tops, tops) * splat(0.03125f);
                        if (k >= cnt) g = splat(0.0f);                           // select, not a branch
                        Ig[k] = g;
                        sxx = fmaf(g.x, g.x, sxx);
                        sxy = fmaf(g.x, g.y, sxy);
                        syy = fmaf(g.y, g.y, syy);
                    }
                    Ip = Ic; GXp = GXc; GYp = GYc;
                }
                sm0 = sm1; sm1 = sm; df0 = df1; df1 = df; ctr1 = mid;
            }
        }
        const float Gxx = wave_sum(sxx), Gxy = wave_sum(sxy), Gyy = wave_sum(syy);

        // next level's template loads fly while this level iterates
        if (L > 0) {
            const float sn = 1.0f / (float)(1u << (L - 1));
            p_x0 = (int)floorf(px * sn) - half - 1;
            const int p_y0 = (int)floorf(py * sn) - half - 1;
            const int nw = (int)a.prev.width[L - 1];
            p_in = tile_inside(p_x0, PW, NDP, nw, a.prev.pitch[L - 1]);
            tile_issue(tp, a.prev.ptr[L - 1], a.prev.pitch[L - 1], nw, (int)a.prev.height[L - 1],
                       p_x0, p_y0, PW, NDP, p_in, lane);
        }

        const float det = Gxx * Gyy - Gxy * Gxy;
        const float dxy = Gxx - Gyy;
        const float disc = sqrtf(dxy * dxy + 4.0f * (Gxy * Gxy));
        const float min_eig = ((Gyy + Gxx) - disc) * inv_area;
        if (!(min_eig >= 1e-4f) || !(det >= FLT_EPSILON)) {
            if (L == 0) ok = false;
            gx = 2.0f * gx; gy = 2.0f * gy;
            continue;
        }
        const float inv_det = 1.0f / det;
        const float flw = (float)lw, flh = (float)lh;

        __syncthreads();   // every lane is done with P
        tile_commit(tj, J, lw, jx0, JW, NDJ, j_in, lane);
        __syncthreads();

        for (uint32_t it = 0; it < a.max_it; ++it) {
            const float cx = (ux + gx) + vx, cy = (uy + gy) + vy;
            if (!(cx > -fw && cy > -fw && cx < flw + fw && cy < flh + fw)) { ok = false; break; }
            const float jfx = floorf(cx), jfy = floorf(cy);
            const float bx = cx - jfx, by = cy - jfy;
            const int jx = (int)jfx - half, jy = (int)jfy - half;
            int ox = jx - jx0, oy = jy - jy0;
            if (ox < 0 || oy < 0 || ox > 2 * LK_MARGIN || oy > 2 * LK_MARGIN) {
                // walked out of the staged tile: restage around the current position (rare)
                jx0 = jx - LK_MARGIN; jy0 = jy - LK_MARGIN; ox = LK_MARGIN; oy = LK_MARGIN;
                j_in = tile_inside(jx0, JW, NDJ, lw, jpitch);
                tile_issue(tj, jp, jpitch, lw, lh, jx0, jy0, JW, NDJ, j_in, lane);
                __syncthreads();
                tile_commit(tj, J, lw, jx0, JW, NDJ, j_in, lane);
                __syncthreads();
            }
#ifdef VSA_LK_STAMPS
            ++st_iters;
#endif
            v2f acc = splat(0.0f);
            {
                const float *jt = J + (oy + row) * JS + (jx0 & 3) + ox + x0;
                v2f prev = { jt[0], jt[JS] };
#pragma unroll
                for (int k = 0; k < NSEG; ++k) {   // pixels past cnt have Ig = 0 and add exact zeros
                    const v2f cur = { jt[k + 1], jt[JS + k + 1] };
                    const v2f h = pk_fma(splat(bx), cur - prev, prev);
                    const float jv = fmaf(by, h.y - h.x, h.x);
                    const float d = Iw[k] - jv;
                    acc = pk_fma(splat(d), Ig[k], acc);
                    prev = cur;
                }
            }
            const float b1 = wave_sum(acc.x), b2 = wave_sum(acc.y);
            const float etx = (Gyy * b1 - Gxy * b2) * inv_det;
            const float ety = (Gxx * b2 - Gxy * b1) * inv_det;
            vx += etx; vy += ety;
            if (a.use_eps && fmaf(etx, etx, ety * ety) < a.eps2) break;
        }
        if (L > 0) { gx = 2.0f * (gx + vx); gy = 2.0f * (gy + vy); }
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
            KeypointF o = reinterpret_cast<const KeypointF *>(a.prev_pts)[pi];
            o.error = 0.0f;
            if (ok) { o.x = rx; o.y = ry; o.tracking_status = 255; }
            else    { o.x = ex; o.y = ey; o.tracking_status = 0; }
            reinterpret_cast<KeypointF *>(a.out)[pi] = o;
        }
    }
#ifdef VSA_LK_STAMPS
    if (lane == 0 && blockIdx.x < 8192) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        uint64_t *o = g_lk_stamps + 4 * blockIdx.x;
        o[0] = st_t0; o[1] = __builtin_readcyclecounter(); o[2] = ((uint64_t)xcc << 32) | hw;
        o[3] = ((wall_clock64() - st_w0) << 32) | st_iters;
    }
#endif
}

}  // namespace

hipError_t launch_pyrlk(const PyrLKParams &p, hipStream_t stream)
{
    if (p.capacity == 0) return hipSuccess;
    for (uint32_t l = 0; l < p.prev.levels; ++l) {
        const bool aligned = ((uintptr_t)p.prev.ptr[l] % 4u == 0) && (p.prev.pitch[l] % 4u == 0) &&
                             ((uintptr_t)p.curr.ptr[l] % 4u == 0) && (p.curr.pitch[l] % 4u == 0);
        if (!aligned) return launch_pyrlk_generic(p, stream);
    }
    LKArgs a{};
    a.prev = p.prev; a.curr = p.curr; a.item = p.item;
    a.prev_pts = p.prev_pts; a.prev_count = p.prev_count; a.estimates = p.estimates;
    a.out = p.out; a.out_count = p.out_count; a.out_count_host = p.out_count_host; a.capacity = p.capacity;
    a.use_eps = p.use_epsilon; a.max_it = p.max_iterations; a.eps2 = p.epsilon * p.epsilon;
    const int win = (int)p.window;
    a.win = win; a.half = win / 2;
    int segs = 64 / win;
    if (segs > win) segs = win;
    if (segs < 1) segs = 1;
    a.seglen = (win + segs - 1) / segs;
    a.segs = (win + a.seglen - 1) / a.seglen;
    a.lanes = a.segs * win;

    const int PW = win + 3, JW = win + 1 + 2 * LK_MARGIN;
    const size_t fl = (size_t)PW * 4 * ((PW + 6) >> 2);
    const size_t jl = (size_t)JW * 4 * ((JW + 6) >> 2);
    a.lds_floats = (int)(fl > jl ? fl : jl) + 32;   // slack: a lane may read NSEG floats past its segment
    const size_t lds = (size_t)a.lds_floats * sizeof(float);
    const dim3 grid(p.capacity), block(64);
    if (win <= 10)      hipLaunchKernelGGL((k_pyrlk<2, 10>),  grid, block, lds, stream, a);
    else if (win <= 16) hipLaunchKernelGGL((k_pyrlk<4, 16>),  grid, block, lds, stream, a);
    else if (win <= 21) hipLaunchKernelGGL((k_pyrlk<7, 21>),  grid, block, lds, stream, a);
    else                hipLaunchKernelGGL((k_pyrlk<16, 32>), grid, block, lds, stream, a);
    return hipGetLastError();
}

}  // namespace vsa

#ifdef VSA_LK_STAMPS
extern "C" int visualslam_amd_debug_lk_stamps(uint64_t *host, size_t n_u64)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vsa::g_lk_stamps), n_u64 * sizeof(uint64_t));
}
#endif
