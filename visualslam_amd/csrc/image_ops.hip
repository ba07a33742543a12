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

__device__ __forceinline__ uint32_t luma709(uint32_t rgbx)
{
    const uint32_t r = rgbx & 0xffu, g = (rgbx >> 8) & 0xffu, b = (rgbx >> 16) & 0xffu;
    return (54u * r + 183u * g + 19u * b + 128u) >> 8;
}

// 4 pixels per lane: one 16-byte load, one 4-byte store.  A wave moves 1 KiB in / 256 B out per
// instruction pair.  Requires 16-B aligned source rows and 4-B aligned destination rows.
__global__ __launch_bounds__(256) void k_rgbx_to_u8_vec(const uint8_t *__restrict__ src, uint32_t sp,
                                                        uint8_t *__restrict__ dst, uint32_t dp,
                                                        uint32_t w, uint32_t h)
{
    const uint32_t x = (blockIdx.x * 64u + threadIdx.x) * 4u;
    const uint32_t y = blockIdx.y * 4u + threadIdx.y;
    if (y >= h || x >= w) return;
    const uint8_t *s = src + (size_t)y * sp + (size_t)x * 4u;
    uint8_t *d = dst + (size_t)y * dp + x;
    if (x + 4u <= w) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s);
        const uint32_t o = luma709(v.x) | (luma709(v.y) << 8) | (luma709(v.z) << 16) | (luma709(v.w) << 24);
        *reinterpret_cast<uint32_t *>(d) = o;
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

// ---- Gaussian HALF step ---------------------------------------------------------------------
// One workgroup produces a 64x16 output tile from a (2*64+3)x(2*16+3) input tile staged in LDS:
// separable [1 4 6 4 1] with the horizontal pass kept at full precision (<= 4080, u16), a single
// rounding after the vertical pass.  Each lane of the vertical pass owns 4 adjacent outputs so
// the store is one dword.
constexpr int PT_W = 64, PT_H = 16;
constexpr int PIN_W = 2 * PT_W + 3, PIN_H = 2 * PT_H + 3, PIN_STRIDE = 132;

__global__ __launch_bounds__(256) void k_gaussian_half(const uint8_t *__restrict__ src, uint32_t sp,
                                                       int sw, int sh,
                                                       uint8_t *__restrict__ dst, uint32_t dp,
                                                       int dw, int dh, int vec_store)
{
    __shared__ uint8_t tin[PIN_H][PIN_STRIDE];
    __shared__ __attribute__((aligned(8))) uint16_t th[PIN_H][PT_W];

    const int tid = threadIdx.x;
    const int ox = blockIdx.x * PT_W, oy = blockIdx.y * PT_H;
    const int sx0 = 2 * ox - 2, sy0 = 2 * oy - 2;

    for (int i = tid; i < PIN_H * PIN_STRIDE; i += 256) {
        const int r = i / PIN_STRIDE, c = i % PIN_STRIDE;
        if (c < PIN_W) {
            const int gy = min(max(sy0 + r, 0), sh - 1);
            const int gx = min(max(sx0 + c, 0), sw - 1);
            tin[r][c] = src[(size_t)gy * sp + gx];
        }
    }
    __syncthreads();
    for (int i = tid; i < PIN_H * PT_W; i += 256) {
        const int r = i / PT_W, x = i % PT_W;
        const uint8_t *p = &tin[r][2 * x];
        th[r][x] = (uint16_t)(p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4]);
    }
    __syncthreads();

    const int x4 = (tid & 15) * 4, y = tid >> 4;
    const int gx = ox + x4, gy = oy + y;
    if (gy >= dh || gx >= dw) return;
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = th[2 * y][x4 + i] + 4u * th[2 * y + 1][x4 + i] + 6u * th[2 * y + 2][x4 + i] +
                           4u * th[2 * y + 3][x4 + i] + th[2 * y + 4][x4 + i];
        o[i] = (s + 128u) >> 8;
    }
    uint8_t *d = dst + (size_t)gy * dp + gx;
    if (vec_store && gx + 4 <= dw) {
        *reinterpret_cast<uint32_t *>(d) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
    } else {
        for (int i = 0; i < 4 && gx + i < dw; ++i) d[i] = (uint8_t)o[i];
    }
}

}  // namespace

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

hipError_t launch_gaussian_half(const ImageView &src, uint8_t *dst, uint32_t dst_pitch,
                                uint32_t dst_width, uint32_t dst_height, hipStream_t stream)
{
    if (dst_width == 0 || dst_height == 0) return hipSuccess;
    const int vec = ((uintptr_t)dst % 4u == 0) && (dst_pitch % 4u == 0);
    const dim3 grid((dst_width + PT_W - 1) / PT_W, (dst_height + PT_H - 1) / PT_H);
    hipLaunchKernelGGL(k_gaussian_half, grid, dim3(256), 0, stream, src.ptr, src.pitch, (int)src.width,
                       (int)src.height, dst, dst_pitch, (int)dst_width, (int)dst_height, vec);
    return hipGetLastError();
}

}  // namespace vsa
