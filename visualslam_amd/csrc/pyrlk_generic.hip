// pyrlk_generic.hip -- any-alignment fallback of the PyrLK kernel (byte gathers, shuffle reductions).
// Used only when a pyramid level's base address or pitch is not 4-byte aligned (ROIs, odd pitches);
// the fast path is pyrlk.hip.  Same arithmetic, same results.
//
// sparse pyramidal Lucas-Kanade optical flow, one wavefront per feature.
//
// Reference call sites replaced: vxOpticalFlowPyrLKNode / nvxcuOpticalFlowPyrLK
// (demos/feature_tracker/feature_tracker.cpp:402-412, demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:297-301,
// samples/object_tracker_nvxcu/object_tracker_keypoint_nvxcu.cpp:1791-1828).  Semantics and the exact
// floating-point operation order: oracle/vxoracle.c (lk_track_one) -- results are bit-identical.
//
// Design (MI355X): a 64-lane wave owns one feature for all pyramid levels, coarse to fine.
//   * window pixels are dealt to lanes as (row, segment): lane l handles `seglen` consecutive
//     pixels of one window row, so the two bilinear source rows are read from LDS once per lane
//     and each column is shared by two neighbouring pixels;
//   * per level the (win+3)^2 template patch is staged in LDS, Scharr gradients are computed on
//     the integer grid into LDS, and the bilinear-resampled template I, Ix, Iy stay in REGISTERS
//     for all iterations;
//   * the search image is staged once per level as a (win+1+2M)^2 LDS tile around the predicted
//     position; an iteration that walks out of the tile restages it (wave-uniform branch);
//   * the 2x2 normal equations live in registers; sums are combined by a cross-lane butterfly
//     whose order the oracle mirrors.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "kernels.hpp"

namespace vsa {
namespace {

constexpr int LK_MARGIN = 4;  // search tile margin M (pixels) on each side of the window

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
};

struct KeypointF {
    float x, y, strength, scale, orientation;
    int32_t tracking_status;
    float error;
};

__device__ __forceinline__ float wave_sum(float v)
{
    // pairwise butterfly, offsets 1,2,4,..,32: every lane ends with the same bits
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float bilerp(float a, float b, float c, float d, float fx, float fy)
{
    const float top = fmaf(fx, b - a, a);
    const float bot = fmaf(fx, d - c, c);
    return fmaf(fy, bot - top, top);
}

// stage an n x n tile of a U8 level into LDS as float, replicate-clamped
__device__ __forceinline__ void stage_tile(float *dst, int n, const uint8_t *img, uint32_t pitch, int w, int h,
                                           int x0, int y0, int lane)
{
    const float inv = 1.0f / (float)n;
    for (int i = lane; i < n * n; i += 64) {
        const int r = (int)(((float)i + 0.5f) * inv);
        const int c = i - r * n;
        const int gy = min(max(y0 + r, 0), h - 1);
        const int gx = min(max(x0 + c, 0), w - 1);
        dst[i] = (float)img[(size_t)gy * pitch + gx];
    }
}

template <int NSEG>
__global__ __launch_bounds__(64) void k_pyrlk_generic(LKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const uint32_t pi = blockIdx.x;
    const uint32_t n = min(*a.prev_count, a.capacity);
    if (pi == 0 && lane == 0) {
        *a.out_count = n;
        if (a.out_count_host) *a.out_count_host = n;
    }
    if (pi >= n) return;

    float px, py, ex, ey;
    KeypointF kp{};
    bool alive = true;
    if (a.item == kPoint2f) {
        const float2 p = reinterpret_cast<const float2 *>(a.prev_pts)[pi];
        const float2 e = reinterpret_cast<const float2 *>(a.estimates)[pi];
        px = p.x; py = p.y; ex = e.x; ey = e.y;
    } else {
        kp = reinterpret_cast<const KeypointF *>(a.prev_pts)[pi];
        const KeypointF *e = reinterpret_cast<const KeypointF *>(a.estimates) + pi;
        px = kp.x; py = kp.y; ex = e->x; ey = e->y;
        alive = kp.tracking_status != 0;
    }

    const int w0 = (int)a.prev.width[0], h0 = (int)a.prev.height[0];
    bool ok = alive && px >= 0.0f && py >= 0.0f && px <= (float)(w0 - 1) && py <= (float)(h0 - 1);

    const int win = a.win, half = a.half;
    const int PW = win + 3, GWD = win + 1, JW = win + 1 + 2 * LK_MARGIN;
    float *P = smem;                 // (win+3)^2 template pixels (1-pixel margin for Scharr)
    float *GX = P + PW * PW;         // (win+1)^2 Scharr x on the integer grid
    float *GY = GX + GWD * GWD;
    float *J = smem;                 // search tile; aliases P/GX/GY (dead once I/Ix/Iy are in registers)

    const int row = lane / a.segs;
    const int x0 = (lane - row * a.segs) * a.seglen;
    int cnt = (lane < a.lanes) ? min(a.seglen, win - x0) : 0;
    if (cnt < 0) cnt = 0;

    const int top = (int)a.prev.levels - 1;
    const float s_top = 1.0f / (float)(1u << top);
    float gx = (ex - px) * s_top, gy = (ey - py) * s_top;
    float vx = 0.0f, vy = 0.0f;
    const float inv_area = 1.0f / (float)(2 * win * win);
    const float fw = (float)win;

    float Iw[NSEG], Ix[NSEG], Iy[NSEG];

    for (int L = top; L >= 0 && ok; --L) {
        const uint8_t *ip = a.prev.ptr[L];
        const uint8_t *jp = a.curr.ptr[L];
        const uint32_t ipitch = a.prev.pitch[L], jpitch = a.curr.pitch[L];
        const int lw = (int)a.prev.width[L], lh = (int)a.prev.height[L];
        const float s = 1.0f / (float)(1u << L);
        const float ux = px * s, uy = py * s;
        const float fx0 = floorf(ux), fy0 = floorf(uy);
        const float ax = ux - fx0, ay = uy - fy0;
        const int ix = (int)fx0 - half, iy = (int)fy0 - half;
        vx = 0.0f; vy = 0.0f;

        __syncthreads();  // previous level's search tile is dead
        stage_tile(P, PW, ip, ipitch, lw, lh, ix - 1, iy - 1, lane);
        __syncthreads();
        {
            const float inv = 1.0f / (float)GWD;
            for (int i = lane; i < GWD * GWD; i += 64) {
                const int r = (int)(((float)i + 0.5f) * inv);
                const int c = i - r * GWD;
                const float *q = P + (r + 1) * PW + (c + 1);
                const float p00 = q[-PW - 1], p01 = q[-PW], p02 = q[-PW + 1];
                const float p10 = q[-1], p12 = q[1];
                const float p20 = q[PW - 1], p21 = q[PW], p22 = q[PW + 1];
                GX[i] = (3.0f * (p02 - p00) + 10.0f * (p12 - p10)) + 3.0f * (p22 - p20);
                GY[i] = (3.0f * (p20 - p00) + 10.0f * (p21 - p01)) + 3.0f * (p22 - p02);
            }
        }
        __syncthreads();

        float sxx = 0.0f, sxy = 0.0f, syy = 0.0f;
        {
            const float *pt = P + (row + 1) * PW + (x0 + 1), *pb = pt + PW;
            const float *xt = GX + row * GWD + x0, *xb = xt + GWD;
            const float *yt = GY + row * GWD + x0, *yb = yt + GWD;
            float a0 = 0, c0 = 0, xa = 0, xc = 0, ya = 0, yc = 0;
            if (cnt > 0) { a0 = pt[0]; c0 = pb[0]; xa = xt[0]; xc = xb[0]; ya = yt[0]; yc = yb[0]; }
#pragma unroll
            for (int k = 0; k < NSEG; ++k) {
                Iw[k] = 0.0f; Ix[k] = 0.0f; Iy[k] = 0.0f;
                if (k < cnt) {
                    const float b0 = pt[k + 1], d0 = pb[k + 1];
                    const float xb_ = xt[k + 1], xd = xb[k + 1];
                    const float yb_ = yt[k + 1], yd = yb[k + 1];
                    Iw[k] = bilerp(a0, b0, c0, d0, ax, ay);
                    Ix[k] = bilerp(xa, xb_, xc, xd, ax, ay) * 0.03125f;
                    Iy[k] = bilerp(ya, yb_, yc, yd, ax, ay) * 0.03125f;
                    sxx = fmaf(Ix[k], Ix[k], sxx);
                    sxy = fmaf(Ix[k], Iy[k], sxy);
                    syy = fmaf(Iy[k], Iy[k], syy);
                    a0 = b0; c0 = d0; xa = xb_; xc = xd; ya = yb_; yc = yd;
                }
            }
        }
        const float Gxx = wave_sum(sxx), Gxy = wave_sum(sxy), Gyy = wave_sum(syy);

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

        // search tile around the prediction
        int jx0 = (int)floorf(ux + gx) - half - LK_MARGIN;
        int jy0 = (int)floorf(uy + gy) - half - LK_MARGIN;
        bool staged = false;

        for (uint32_t it = 0; it < a.max_it; ++it) {
            const float cx = (ux + gx) + vx, cy = (uy + gy) + vy;
            if (!(cx > -fw && cy > -fw && cx < flw + fw && cy < flh + fw)) { ok = false; break; }
            const float jfx = floorf(cx), jfy = floorf(cy);
            const float bx = cx - jfx, by = cy - jfy;
            const int jx = (int)jfx - half, jy = (int)jfy - half;
            int ox = jx - jx0, oy = jy - jy0;
            if (!staged || ox < 0 || oy < 0 || ox > 2 * LK_MARGIN || oy > 2 * LK_MARGIN) {
                if (staged) { jx0 = jx - LK_MARGIN; jy0 = jy - LK_MARGIN; ox = LK_MARGIN; oy = LK_MARGIN; }
                __syncthreads();
                stage_tile(J, JW, jp, jpitch, lw, lh, jx0, jy0, lane);
                __syncthreads();
                staged = true;
            }
            float s1 = 0.0f, s2 = 0.0f;
            {
                const float *jt = J + (oy + row) * JW + (ox + x0), *jb = jt + JW;
                float a0 = 0, c0 = 0;
                if (cnt > 0) { a0 = jt[0]; c0 = jb[0]; }
#pragma unroll
                for (int k = 0; k < NSEG; ++k) {
                    if (k < cnt) {
                        const float b0 = jt[k + 1], d0 = jb[k + 1];
                        const float jv = bilerp(a0, b0, c0, d0, bx, by);
                        const float d = Iw[k] - jv;
                        s1 = fmaf(d, Ix[k], s1);
                        s2 = fmaf(d, Iy[k], s2);
                        a0 = b0; c0 = d0;
                    }
                }
            }
            const float b1 = wave_sum(s1), b2 = wave_sum(s2);
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
            KeypointF o = kp;
            o.error = 0.0f;
            if (ok) { o.x = rx; o.y = ry; o.tracking_status = 255; }
            else    { o.x = ex; o.y = ey; o.tracking_status = 0; }
            reinterpret_cast<KeypointF *>(a.out)[pi] = o;
        }
    }
}

}  // namespace

hipError_t launch_pyrlk_generic(const PyrLKParams &p, hipStream_t stream)
{
    if (p.capacity == 0) return hipSuccess;
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

    const int PW = win + 3, GWD = win + 1, JW = win + 1 + 2 * LK_MARGIN;
    const size_t fl = (size_t)PW * PW + 2 * (size_t)GWD * GWD;
    const size_t jl = (size_t)JW * JW;
    const size_t lds = (fl > jl ? fl : jl) * sizeof(float);
    const dim3 grid(p.capacity), block(64);
    if (a.seglen <= 2)       hipLaunchKernelGGL(k_pyrlk_generic<2>,  grid, block, lds, stream, a);
    else if (a.seglen <= 4)  hipLaunchKernelGGL(k_pyrlk_generic<4>,  grid, block, lds, stream, a);
    else if (a.seglen <= 7)  hipLaunchKernelGGL(k_pyrlk_generic<7>,  grid, block, lds, stream, a);
    else                     hipLaunchKernelGGL(k_pyrlk_generic<16>, grid, block, lds, stream, a);
    return hipGetLastError();
}

}  // namespace vsa
