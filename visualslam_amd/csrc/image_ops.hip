// image_ops.hip -- streaming U8 kernels: RGBX->U8 colour convert and the HALF-scale Gaussian
// pyramid step.  Both are pure HBM streams: every input byte is read
// once from global memory, halos are served from LDS.
//
// Reference call sites replaced: vxColorConvertNode / nvxcuColorConvert
// (demos/feature_tracker/feature_tracker.cpp:386, demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:279-282)
// and vxGaussianPyramidNode / nvxcuGaussianPyramid (feature_tracker.cpp:393, feature_tracker_nvxcu.cpp:284-285).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace vsa {
namespace {

// (54 R + 183 G + 19 B + 128) >> 8 as one byte-wise dot product (v_dot4_u32_u8: bytes R, G, B, X against 54, 183, 19, 0,
// accumulated onto the rounding constant) and a shift -- two instructions a pixel instead of three field extractions
// and three multiply-adds
__device__ __forceinline__ uint32_t luma709_x256(uint32_t rgbx) { return __builtin_amdgcn_udot4(rgbx, 0x0013B736u, 128u, false); }
__device__ __forceinline__ uint32_t luma709(uint32_t rgbx) { return luma709_x256(rgbx) >> 8; }
// four pixels -> four luma bytes: byte 1 of each 16-bit sum is the luma, two byte permutes gather them
__device__ __forceinline__ uint32_t luma709_x4(uint4 v)
{
    const uint32_t lo = __builtin_amdgcn_perm(luma709_x256(v.y), luma709_x256(v.x), 0x0c0c0501u);
    const uint32_t hi = __builtin_amdgcn_perm(luma709_x256(v.w), luma709_x256(v.z), 0x05010c0cu);
    return lo | hi;
}

// 4 pixels per lane: one 16-byte load, one 4-byte store.  A wave moves 1 KiB in / 256 B out per
// instruction pair.  Requires 16-B aligned source rows and 4-B aligned destination rows.
__global__ __launch_bounds__(256) void k_rgbx_to_u8_vec(const uint8_t *__restrict__ src, uint32_t sp,
                                                        uint8_t *__restrict__ dst, uint32_t dp,
                                                        uint32_t w, uint32_t h)
{
    VSA_SHORT_KERNEL_PRIORITY();
    const uint32_t x = (blockIdx.x * 64u + threadIdx.x) * 4u;
    const uint32_t y = blockIdx.y * 4u + threadIdx.y;
    if (y >= h || x >= w) return;
    const uint8_t *s = src + (size_t)y * sp + (size_t)x * 4u;
    uint8_t *d = dst + (size_t)y * dp + x;
    if (x + 4u <= w) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s);
        *reinterpret_cast<uint32_t *>(d) = luma709_x4(v);
    } else {
        for (uint32_t i = 0; x + i < w; ++i)
            d[i] = (uint8_t)luma709(reinterpret_cast<const uint32_t *>(s)[i]);
    }
}

// fallback for rows that are not 16-B / 4-B aligned (ROIs, odd pitches)
__global__ __launch_bounds__(256) void k_rgbx_to_u8_scalar(const uint8_t *__restrict__ src, uint32_t sp,
                                                           uint8_t *__restrict__ dst, uint32_t dp,
                                                           uint32_t w, uint32_t h)
{
    const uint32_t x = blockIdx.x * 64u + threadIdx.x;
    const uint32_t y = blockIdx.y * 4u + threadIdx.y;
    if (y >= h || x >= w) return;
    const uint8_t *s = src + (size_t)y * sp + (size_t)x * 4u;
    dst[(size_t)y * dp + x] = (uint8_t)((54u * s[0] + 183u * s[1] + 19u * s[2] + 128u) >> 8);
}

// ---- Gaussian pyramid, up to four HALF steps fused in one launch -----------------------------
// One workgroup owns a TxT tile of the group's top level and builds everything under it in LDS:
// the source region (side 2^NS*T + 3*(2^NS-1)) is fetched once as aligned dwords, then each step
// runs the separable [1 4 6 4 1] filter LDS->LDS (horizontal pass kept at full precision in u16,
// one rounding after the vertical pass) and stores the pixels the tile owns.  Regions carry
// replicate semantics: the entry for an out-of-range coordinate holds the value at the clamped
// coordinate, so every step is the oracle's clamp-at-each-level arithmetic.
// For 1080p / 5 levels this is ONE launch of 510 workgroups instead of four dependent launches,
// and level 0 is read from HBM once.
// 8 waves per workgroup.  16 waves are a little faster when the kernel has the GPU to itself (9.3 against ~10 us at 1080p) but a
// 16-wave, 27 KB workgroup finds a compute unit with room less often while other camera streams' PyrLK waves hold wave slots and
// LDS: with 16 streams the whole bench runs 4 % faster with 8 waves (tools/exp_lib_ab.sh: 1024 threads 26.0 k, 512 27.1 k, 256
// 25.9 k frames/s)
constexpr int PYR_THREADS = 512;
constexpr int pyr_region(int t, int up) { return up == 0 ? t : 2 * pyr_region(t, up - 1) + 3; }

struct PyrFusedArgs {
    const uint8_t *src;
    uint32_t src_pitch;
    int sw, sh;
    uint8_t *dst[4];
    uint32_t dst_pitch[4];
    int dw[4], dh[4];
    int aligned;    // src base and pitch are multiples of 4
    // RGBX variants only: the colour frame level 0 is made from (16-byte aligned base and pitch, rows readable up to the
    // next multiple of 4 pixels); `src` is then level 0 as an OUTPUT (4-byte aligned), written by the tiles that own it
    const uint8_t *rgbx;
    uint32_t rgbx_pitch;
};

constexpr int pyr_pow2_at_least(int n) { return n <= 1 ? 1 : 2 * pyr_pow2_at_least((n + 1) / 2); }

// LDS layout of a step: the source tile has a row stride SS that is a multiple of 4 (so the byte offset of a column
// inside its dword is the same in every row), the destination tile a stride SD = pyr_stride(RD), the intermediate
// (horizontal sums, u16, at most 16 * 255) a stride TD = RD rounded up to even so that two columns share a dword.
// Rows past a tile's last column hold values nobody uses (finite garbage is fine: all arithmetic is integer and the
// packed halves are independent).
constexpr int pyr_stride(int rd) { return (rd + 3) / 4 * 4; }
constexpr int pyr_tstride(int rd) { return (rd + 1) / 2 * 2; }

typedef unsigned short pyr_u16x2 __attribute__((ext_vector_type(2)));

// One HALF step, LDS -> LDS (+ the owned pixels to global memory).
//   horizontal: a thread keeps one destination column (the replicate clamp, the source dword and the byte offset
//     inside it are computed once); per row TWO aligned dwords (one ds_read2_b32) hold the five taps, v_alignbyte
//     brings the first four into one register and v_dot4_u32_u8 weighs them 1 4 6 4 with the fifth as the accumulator:
//     1 LDS read and 3 vector instructions per value (it was 5 byte reads and 4 multiply-adds);
//   vertical: a thread keeps a PAIR of destination columns; five dword reads of the u16 sums and packed 16-bit
//     arithmetic (the total, 16 * 16 * 255 + 128, still fits 16 bits) give two pixels.
// Every value is the exact integer the oracle computes: (sum of the 25 weighted taps + 128) >> 8.
template <int RS, int SS, int RD>
__device__ __forceinline__ void pyr_step(const uint8_t *src, int soff, int sox, int soy,
                                         uint8_t *dstl, int dox, int doy, int dw, int dh,
                                         uint16_t *tmp, uint8_t *gdst, uint32_t gpitch,
                                         int own_x0, int own_y0, int own_n, int tid)
{
    static_assert(SS % 4 == 0, "source rows start on dword boundaries");
    constexpr int SD = pyr_stride(RD), TD = pyr_tstride(RD);
    {   // horizontal: every source-region row, every destination column
        constexpr int CW = pyr_pow2_at_least(RD) < PYR_THREADS ? pyr_pow2_at_least(RD) : PYR_THREADS;
        constexpr int RSTEP = PYR_THREADS / CW;
        const int c = tid & (CW - 1), r0 = tid / CW;
        const int xc = min(max(dox + c, 0), dw - 1);
        const int A = soff + (2 * xc - 2 - sox);                 // byte offset of the first tap inside a row
        const uint32_t shift = (uint32_t)A & 3u;
        const uint32_t *p = reinterpret_cast<const uint32_t *>(src + r0 * SS + (A & ~3));
        uint16_t *t = tmp + r0 * TD + c;
        if (c < RD)
            for (int r = r0; r < RS; r += RSTEP, p += RSTEP * (SS / 4), t += RSTEP * TD) {
                const uint32_t lo = p[0], hi = p[1];
                const uint32_t taps = __builtin_amdgcn_alignbyte(hi, lo, shift);          // taps 0..3
                const uint32_t last = __builtin_amdgcn_ubfe(hi, 8u * shift, 8u);          // tap 4
                *t = (uint16_t)__builtin_amdgcn_udot4(taps, 0x04060401u, last, false);
            }
    }
    __syncthreads();
    {   // vertical: every destination row, pairs of destination columns
        constexpr int NP = TD / 2;
        constexpr int CW = pyr_pow2_at_least(NP) < PYR_THREADS ? pyr_pow2_at_least(NP) : PYR_THREADS;
        constexpr int RSTEP = PYR_THREADS / CW;
        const int cp = tid & (CW - 1), r0 = tid / CW;
        const int X0 = dox + 2 * cp;
        const bool own0 = X0 >= own_x0 && X0 < own_x0 + own_n && X0 < dw && 2 * cp < RD;
        const bool own1 = X0 + 1 >= own_x0 && X0 + 1 < own_x0 + own_n && X0 + 1 < dw && 2 * cp + 1 < RD;
        const int y_end = min(own_y0 + own_n, dh);
        const pyr_u16x2 *tw = reinterpret_cast<const pyr_u16x2 *>(tmp) + cp;
        if (cp < NP)
            for (int y = r0; y < RD; y += RSTEP) {
                const int Y = doy + y;
                const int yc = min(max(Y, 0), dh - 1);
                const pyr_u16x2 *t = tw + (2 * yc - 2 - soy) * NP;
                pyr_u16x2 sum = (t[0] + t[4 * NP]) + (t[NP] + t[3 * NP]) * (unsigned short)4 + t[2 * NP] * (unsigned short)6;
                sum = (sum + (unsigned short)128) >> (unsigned short)8;
                const uint32_t v0 = sum.x, v1 = sum.y;
                *reinterpret_cast<uint16_t *>(dstl + y * SD + 2 * cp) = (uint16_t)(v0 | (v1 << 8));
                if (Y >= own_y0 && Y < y_end) {
                    uint8_t *g = gdst + (size_t)Y * gpitch + X0;
                    if (own0) g[0] = (uint8_t)v0;
                    if (own1) g[1] = (uint8_t)v1;
                }
            }
    }
    __syncthreads();
}

// RGBX: the launch also IS the colour convert -- the source region is fetched from the RGBX frame (16 bytes = 4 pixels
// per load), turned into luma on the way into LDS, and every tile stores the part of level 0 it owns.  One launch less
// per frame: on this workload the four hardware queues run their streams' kernels back to back with ~4.5 us between two
// kernels, so a launch costs ~4 % of the frame rate whatever it computes (profiles/r02_kernel_experiments.txt), against
// which the luma of the halos being computed by several tiles is cheap.
template <int NS, int T, bool RGBX>
__global__ __launch_bounds__(PYR_THREADS) void k_pyramid_fused(PyrFusedArgs a)
{
    VSA_SHORT_KERNEL_PRIORITY();
    constexpr int R0 = pyr_region(T, NS);
    constexpr int R1 = pyr_region(T, NS - 1);
    constexpr int R2 = NS >= 2 ? pyr_region(T, NS - 2) : 1;
    constexpr int R3 = NS >= 3 ? pyr_region(T, NS - 3) : 1;
    constexpr int R4 = NS >= 4 ? pyr_region(T, NS - 4) : 1;
    constexpr int S0 = (R0 + 3 + 3) / 4 * 4;          // aligned-dword row stride of the source region
    constexpr int ND = S0 / 4;

    // + 8: the two-dword reads of the last row's last columns may run a few bytes past the tile
    __shared__ __attribute__((aligned(16))) uint8_t l0[R0 * S0 + 8];
    __shared__ __attribute__((aligned(16))) uint8_t l1[R1 * pyr_stride(R1) + 8];
    __shared__ __attribute__((aligned(16))) uint8_t l2[R2 * pyr_stride(R2) + 8];
    __shared__ __attribute__((aligned(16))) uint8_t l3[R3 * pyr_stride(R3) + 8];
    __shared__ __attribute__((aligned(16))) uint8_t l4[R4 * pyr_stride(R4) + 8];
    __shared__ __attribute__((aligned(16))) uint16_t tmp[R0 * pyr_tstride(R1)];

    const int tid = threadIdx.x;
    int ox[5], oy[5];
    // each XCD builds a contiguous band of tiles: the source halos neighbours share stay in one L2
    const uint32_t item = xcd_contiguous_item(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    ox[NS] = (int)(item % gridDim.x) * T;
    oy[NS] = (int)(item / gridDim.x) * T;
#pragma unroll
    for (int g = NS; g > 0; --g) { ox[g - 1] = 2 * ox[g] - 2; oy[g - 1] = 2 * oy[g] - 2; }

    // source region
    const int xa = ox[0] & ~3, off = ox[0] - xa;
    if (RGBX || a.aligned) {
        uint32_t *l0w = reinterpret_cast<uint32_t *>(l0);
        const int xmax = (a.sw - 1) & ~3;
        constexpr int NLD = (R0 * ND + PYR_THREADS - 1) / PYR_THREADS;
        if (RGBX) {
            uint4 v[NLD];
#pragma unroll
            for (int k = 0; k < NLD; ++k) {   // all loads in flight before the first LDS write
                const int i = min(tid + k * PYR_THREADS, R0 * ND - 1);
                const int r = i / ND, d = i % ND;
                const int gy = min(max(oy[0] + r, 0), a.sh - 1);
                const int X = min(max(xa + 4 * d, 0), xmax);
                v[k] = *reinterpret_cast<const uint4 *>(a.rgbx + (size_t)gy * a.rgbx_pitch + 4 * (size_t)X);
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) l0w[min(tid + k * PYR_THREADS, R0 * ND - 1)] = luma709_x4(v[k]);
            __syncthreads();
            // level 0 itself: the T << NS square this tile owns, as dwords (its first column is a multiple of 4 both in
            // the image and in the region: ox[0] = own_x0 - (2^(NS+1) - 2), xa = ox[0] - 2)
            constexpr int OWN = T << NS, OD = OWN / 4;
            const int own_x0 = ox[NS] << NS, own_y0 = oy[NS] << NS;
            const int dcol = (own_x0 - xa) / 4, rrow = own_y0 - oy[0];
            uint8_t *lvl0 = const_cast<uint8_t *>(a.src);
            for (int i = tid; i < OWN * OD; i += PYR_THREADS) {
                const int r = i / OD, d = i % OD;
                const int Y = own_y0 + r, X = own_x0 + 4 * d;
                if (Y >= a.sh || X >= a.sw) continue;
                const uint32_t px = l0w[(rrow + r) * ND + dcol + d];
                uint8_t *g = lvl0 + (size_t)Y * a.src_pitch + X;
                if (X + 4 <= a.sw) *reinterpret_cast<uint32_t *>(g) = px;
                else for (int b = 0; X + b < a.sw; ++b) g[b] = (uint8_t)(px >> (8 * b));
            }
        } else {
            uint32_t v[NLD];
#pragma unroll
            for (int k = 0; k < NLD; ++k) {   // all loads in flight before the first LDS write
                const int i = min(tid + k * PYR_THREADS, R0 * ND - 1);
                const int r = i / ND, d = i % ND;
                const int gy = min(max(oy[0] + r, 0), a.sh - 1);
                const int X = min(max(xa + 4 * d, 0), xmax);
                v[k] = *reinterpret_cast<const uint32_t *>(a.src + (size_t)gy * a.src_pitch + X);
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) l0w[min(tid + k * PYR_THREADS, R0 * ND - 1)] = v[k];
            __syncthreads();
        }
        if (ox[0] < 0 || ox[0] + R0 > a.sw) {   // replicate the columns outside the image
            for (int i = tid; i < R0 * R0; i += PYR_THREADS) {
                const int r = i / R0, c = i % R0;
                const int X = ox[0] + c;
                if (X < 0 || X >= a.sw) l0[r * S0 + off + c] = l0[r * S0 + (min(max(X, 0), a.sw - 1) - xa)];
            }
            __syncthreads();
        }
    } else {
        for (int i = tid; i < R0 * R0; i += PYR_THREADS) {
            const int r = i / R0, c = i % R0;
            const int gy = min(max(oy[0] + r, 0), a.sh - 1);
            const int gx = min(max(ox[0] + c, 0), a.sw - 1);
            l0[r * S0 + off + c] = a.src[(size_t)gy * a.src_pitch + gx];
        }
        __syncthreads();
    }

    pyr_step<R0, S0, R1>(l0, off, ox[0], oy[0], l1, ox[1], oy[1], a.dw[0], a.dh[0], tmp, a.dst[0], a.dst_pitch[0],
                         ox[NS] << (NS - 1), oy[NS] << (NS - 1), T << (NS - 1), tid);
    if (NS >= 2)
        pyr_step<R1, pyr_stride(R1), R2>(l1, 0, ox[1], oy[1], l2, ox[2], oy[2], a.dw[1], a.dh[1], tmp, a.dst[1], a.dst_pitch[1],
                             ox[NS] << (NS - 2), oy[NS] << (NS - 2), T << (NS - 2), tid);
    if (NS >= 3)
        pyr_step<R2, pyr_stride(R2), R3>(l2, 0, ox[2], oy[2], l3, ox[3], oy[3], a.dw[2], a.dh[2], tmp, a.dst[2], a.dst_pitch[2],
                             ox[NS] << (NS - 3), oy[NS] << (NS - 3), T << (NS - 3), tid);
    if (NS >= 4)
        pyr_step<R3, pyr_stride(R3), R4>(l3, 0, ox[3], oy[3], l4, ox[4], oy[4], a.dw[3], a.dh[3], tmp, a.dst[3], a.dst_pitch[3],
                             ox[NS], oy[NS], T, tid);
}


// ---- perspective warp ------------------------------------------------------------------------------
// output(x,y) = input(X/Z, Y/Z), (X,Y,Z) = M (x,y,1).  One thread per output pixel, 64x4 tiles dealt to the XCDs in
// contiguous bands of rows (near-identity warps then re-read their taps from one L2).  Arithmetic and its order:
// oracle/vxoracle_homography.c (warp perspective) -- results are bit-identical.
struct WarpArgs {
    const uint8_t *src;
    uint32_t src_pitch;
    int sw, sh;
    uint8_t *dst;
    uint32_t dst_pitch;
    int dw, dh;
    float m[9];
    int bilinear;
    uint32_t border;      // constant border pixel (U8: low byte)
    uint32_t tiles_x, tiles;
};

template <int CH>
__device__ __forceinline__ uint32_t warp_fetch(const uint8_t *row, int x)
{
    return CH == 4 ? reinterpret_cast<const uint32_t *>(row)[x] : (uint32_t)row[x];
}

__device__ __forceinline__ uint32_t warp_lerp_channel(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, int shift, float fx, float fy)
{
    const float a = (float)((p00 >> shift) & 0xffu), b = (float)((p01 >> shift) & 0xffu);
    const float c = (float)((p10 >> shift) & 0xffu), d = (float)((p11 >> shift) & 0xffu);
    const float top = fmaf(fx, b - a, a), bot = fmaf(fx, d - c, c);
    const float v = fmaf(fy, bot - top, top);
    return ((uint32_t)(int)(v + 0.5f)) << shift;
}

template <int CH>
__global__ __launch_bounds__(256) void k_warp_perspective(WarpArgs a)
{
    const uint32_t item = xcd_contiguous_item(blockIdx.x, a.tiles);
    const int x = (int)((item % a.tiles_x) * 64u + threadIdx.x), y = (int)((item / a.tiles_x) * 4u + threadIdx.y);
    if (x >= a.dw || y >= a.dh) return;
    const float xf = (float)x, yf = (float)y;
    const float X = fmaf(a.m[0], xf, fmaf(a.m[1], yf, a.m[2]));
    const float Y = fmaf(a.m[3], xf, fmaf(a.m[4], yf, a.m[5]));
    const float Z = fmaf(a.m[6], xf, fmaf(a.m[7], yf, a.m[8]));
    const float xs = X / Z, ys = Y / Z;
    uint32_t out = a.border;
    if (a.bilinear) {
        if (xs >= 0.0f && ys >= 0.0f && xs <= (float)(a.sw - 1) && ys <= (float)(a.sh - 1)) {   // false for NaN
            const float x0f = floorf(xs), y0f = floorf(ys);
            const float fx = xs - x0f, fy = ys - y0f;
            const int x0 = (int)x0f, y0 = (int)y0f;
            const int x1 = min(x0 + 1, a.sw - 1), y1 = min(y0 + 1, a.sh - 1);
            const uint8_t *r0 = a.src + (size_t)y0 * a.src_pitch, *r1 = a.src + (size_t)y1 * a.src_pitch;
            const uint32_t p00 = warp_fetch<CH>(r0, x0), p01 = warp_fetch<CH>(r0, x1);
            const uint32_t p10 = warp_fetch<CH>(r1, x0), p11 = warp_fetch<CH>(r1, x1);
            out = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) out |= warp_lerp_channel(p00, p01, p10, p11, 8 * c, fx, fy);
        }
    } else {
        const float xr = floorf(xs + 0.5f), yr = floorf(ys + 0.5f);
        if (xr >= 0.0f && yr >= 0.0f && xr <= (float)(a.sw - 1) && yr <= (float)(a.sh - 1))
            out = warp_fetch<CH>(a.src + (size_t)(int)yr * a.src_pitch, (int)xr);
    }
    uint8_t *o = a.dst + (size_t)y * a.dst_pitch;
    if (CH == 4) reinterpret_cast<uint32_t *>(o)[x] = out;
    else o[x] = (uint8_t)out;
}

}  // namespace

hipError_t launch_warp_perspective(const WarpPerspectiveParams &p, hipStream_t stream)
{
    if (p.dst_width == 0 || p.dst_height == 0) return hipSuccess;
    if (p.src.width == 0 || p.src.height == 0 || (p.channels != 1 && p.channels != 4)) return hipErrorInvalidValue;
    if (p.channels == 4 && (((uintptr_t)p.src.ptr | (uintptr_t)p.dst | p.src.pitch | p.dst_pitch) & 3u)) return hipErrorInvalidValue;
    WarpArgs a{};
    a.src = p.src.ptr; a.src_pitch = p.src.pitch; a.sw = (int)p.src.width; a.sh = (int)p.src.height;
    a.dst = p.dst; a.dst_pitch = p.dst_pitch; a.dw = (int)p.dst_width; a.dh = (int)p.dst_height;
    for (int i = 0; i < 9; ++i) a.m[i] = p.m[i];
    a.bilinear = p.bilinear ? 1 : 0;
    a.border = p.border;
    a.tiles_x = (p.dst_width + 63u) / 64u;
    a.tiles = a.tiles_x * ((p.dst_height + 3u) / 4u);
    const dim3 block(64, 4);
    if (p.channels == 4) hipLaunchKernelGGL(k_warp_perspective<4>, dim3(a.tiles), block, 0, stream, a);
    else hipLaunchKernelGGL(k_warp_perspective<1>, dim3(a.tiles), block, 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rgbx_to_u8(const ImageView &src, uint8_t *dst, uint32_t dst_pitch, hipStream_t stream)
{
    if (src.width == 0 || src.height == 0) return hipSuccess;
    const bool vec = ((uintptr_t)src.ptr % 16u == 0) && (src.pitch % 16u == 0) &&
                     ((uintptr_t)dst % 4u == 0) && (dst_pitch % 4u == 0);
    const dim3 block(64, 4);
    if (vec) {
        const dim3 grid((src.width + 255u) / 256u, (src.height + 3u) / 4u);
        hipLaunchKernelGGL(k_rgbx_to_u8_vec, grid, block, 0, stream, src.ptr, src.pitch, dst, dst_pitch,
                           src.width, src.height);
    } else {
        const dim3 grid((src.width + 63u) / 64u, (src.height + 3u) / 4u);
        hipLaunchKernelGGL(k_rgbx_to_u8_scalar, grid, block, 0, stream, src.ptr, src.pitch, dst, dst_pitch,
                           src.width, src.height);
    }
    return hipGetLastError();
}

namespace {

// levels 1.. of the pyramid, up to four steps per launch; `rgbx` != nullptr: the first launch also makes level 0 from it
hipError_t launch_pyramid_levels(const PyramidView &pyr, const ImageView *rgbx, hipStream_t stream)
{
    uint32_t lvl = 0;
    while (lvl + 1 < pyr.levels) {
        // More than four steps left: the launch on the LARGER level takes the fewer steps -- a tile's source region grows with the
        // steps under it (67, 73, 85, 109 pixels a side for 64 owned), and that halo is paid on every pixel of the launch's source
        // level.  Five steps as 2 + 3 read 1.4 times the larger level, as 4 + 1 they read 2.9 times (4K / 7 levels: +1.2 %).
        const uint32_t left = pyr.levels - 1 - lvl;
        uint32_t ns = left <= 4u ? left : (left <= 8u ? (left - 4u > 2u ? left - 4u : 2u) : 4u);
        PyrFusedArgs a{};
        a.src = pyr.ptr[lvl]; a.src_pitch = pyr.pitch[lvl];
        a.sw = (int)pyr.width[lvl]; a.sh = (int)pyr.height[lvl];
        a.aligned = ((uintptr_t)a.src % 4u == 0) && (a.src_pitch % 4u == 0);
        const bool from_rgbx = rgbx != nullptr && lvl == 0;
        // The launch that reads the colour frame builds ONE level: a tile's source region is then 67 x 67 pixels for the
        // 64 x 64 it owns, against 109 x 109 with four levels under it -- the luma of the halo is computed by every tile
        // that needs it, and the frame is 4 bytes a pixel.  Measured on the 32-stream bench (frames/s): 1 level 32.6 k,
        // 2 levels 32.0 k, 3 levels 31.3 k, 4 levels 31.1 k, colour convert as its own launch 31.4 k.
        if (from_rgbx) ns = 1;
        if (from_rgbx) { a.rgbx = rgbx->ptr; a.rgbx_pitch = rgbx->pitch; }
        for (uint32_t g = 0; g < ns; ++g) {
            a.dst[g] = const_cast<uint8_t *>(pyr.ptr[lvl + 1 + g]);
            a.dst_pitch[g] = pyr.pitch[lvl + 1 + g];
            a.dw[g] = (int)pyr.width[lvl + 1 + g];
            a.dh[g] = (int)pyr.height[lvl + 1 + g];
        }
        const int tw = a.dw[ns - 1], th = a.dh[ns - 1];
#define VSA_PYR_LAUNCH(NS_, T_)                                                                                                       \
    do {                                                                                                                              \
        const dim3 grid((tw + (T_) - 1) / (T_), (th + (T_) - 1) / (T_));                                                              \
        if (from_rgbx) hipLaunchKernelGGL((k_pyramid_fused<NS_, T_, true>), grid, dim3(PYR_THREADS), 0, stream, a);                   \
        else hipLaunchKernelGGL((k_pyramid_fused<NS_, T_, false>), grid, dim3(PYR_THREADS), 0, stream, a);                            \
    } while (0)
        // Frames up to 640 x 480 (BASELINE configs[0]): a camera stream's launches are a chain of latencies on a GPU with room to
        // spare, and half-size top tiles shorten a workgroup's chain of steps and barriers while the extra halo work finds idle
        // compute units: 83.3 k against 80.6 k frames/s.  At 1080p the GPU is busy and the same choice for the launches with
        // fewer workgroups than compute units costs 1.5 % (configs[4]: 3 %), profiles/r03_kernel_experiments.txt.
        const bool small_frame = (uint64_t)pyr.width[0] * pyr.height[0] <= 640u * 480u;
        switch (ns) {
        case 1: if (small_frame) VSA_PYR_LAUNCH(1, 16); else VSA_PYR_LAUNCH(1, 32); break;
        case 2: VSA_PYR_LAUNCH(2, 16); break;
        case 3: VSA_PYR_LAUNCH(3, 8); break;
        default: if (small_frame) VSA_PYR_LAUNCH(4, 2); else VSA_PYR_LAUNCH(4, 4); break;
        }
#undef VSA_PYR_LAUNCH
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        lvl += ns;
    }
    return hipSuccess;
}

}  // namespace

hipError_t launch_gaussian_pyramid(const PyramidView &pyr, hipStream_t stream) { return launch_pyramid_levels(pyr, nullptr, stream); }

bool color_pyramid_fusable(const ImageView &rgbx, const PyramidView &pyr)
{
    if (pyr.levels < 2 || rgbx.width != pyr.width[0] || rgbx.height != pyr.height[0] || rgbx.width == 0 || rgbx.height == 0) return false;
    const uint32_t padded = (rgbx.width + 3u) & ~3u;      // the loads take whole groups of 4 pixels
    return ((uintptr_t)rgbx.ptr % 16u == 0) && (rgbx.pitch % 16u == 0) && (rgbx.pitch >= 4u * padded) &&
           ((uintptr_t)pyr.ptr[0] % 4u == 0) && (pyr.pitch[0] % 4u == 0) && (pyr.pitch[0] >= padded);
}

hipError_t launch_color_pyramid(const ImageView &rgbx, const PyramidView &pyr, hipStream_t stream)
{
    if (!color_pyramid_fusable(rgbx, pyr)) return hipErrorInvalidValue;
    return launch_pyramid_levels(pyr, &rgbx, stream);
}

}  // namespace vsa
