// kernels.hpp -- internal launch interface between the HIP kernels (*.hip) and the C-ABI
// layers (nvxcu_api.cpp, the OpenVX-subset runtime).  Nothing here is exported.
//
// All launchers are asynchronous on `stream` and allocate nothing.
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime_api.h>

// Kernels of one camera stream run one after the other, and streams that share a hardware queue wait for each other, so a
// short, latency-bound kernel (colour convert, pyramid, detector, merge) that is slowed by another stream's PyrLK waves holds
// up its whole queue.  Their waves therefore ask for issue priority over PyrLK's (s_setprio: 0 = default ... 3 = highest).
#define VSA_SHORT_KERNEL_PRIORITY() __builtin_amdgcn_s_setprio(3)

namespace vsa {

constexpr int kMaxPyrLevels = 10;
constexpr int kLkMinWin = 3;
constexpr int kLkMaxWin = 32;

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  This maps workgroup id `bid` of a
// grid of `n` to a work item so that every XCD owns one CONTIGUOUS range of items (a bijection on [0,n)):
// neighbouring items -- spatially neighbouring tiles / features -- then share an L2 instead of being
// fetched once per XCD.
constexpr uint32_t kNumXcd = 8;
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t xcd_contiguous_item(uint32_t bid, uint32_t n)
{
    const uint32_t q = n / kNumXcd, r = n % kNumXcd;
    const uint32_t x = bid % kNumXcd, k = bid / kNumXcd;
    return x * q + (x < r ? x : r) + k;
}
#endif

enum ItemKind : int { kPoint2f = 0, kKeypointf = 1 };
enum Detector : int { kHarris = 0, kFast = 1 };

struct ImageView {
    const uint8_t *ptr;
    uint32_t pitch, width, height;
};

struct PyramidView {
    const uint8_t *ptr[kMaxPyrLevels];
    uint32_t pitch[kMaxPyrLevels];
    uint32_t width[kMaxPyrLevels];
    uint32_t height[kMaxPyrLevels];
    uint32_t levels;
};

// RGBX -> U8 luma, (54 R + 183 G + 19 B + 128) >> 8
hipError_t launch_rgbx_to_u8(const ImageView &src, uint8_t *dst, uint32_t dst_pitch, hipStream_t stream);

// levels 1..n-1 from level 0: dst(x,y) = (sum_{i,j} w_i w_j src(clamp(2x+i-2), clamp(2y+j-2)) + 128) >> 8,
// w = [1 4 6 4 1]; up to four steps fused per launch
hipError_t launch_gaussian_pyramid(const PyramidView &pyr, hipStream_t stream);

// RGBX -> U8 into level 0 AND levels 1..n-1 in the pyramid's first launch (one launch less than launch_rgbx_to_u8 +
// launch_gaussian_pyramid, identical bytes).  Needs >= 2 levels, a 16-byte aligned frame whose rows are readable up to
// the next multiple of 4 pixels, and a 4-byte aligned level 0: color_pyramid_fusable() says whether that holds.
bool color_pyramid_fusable(const ImageView &rgbx, const PyramidView &pyr);
hipError_t launch_color_pyramid(const ImageView &rgbx, const PyramidView &pyr, hipStream_t stream);

// output(x,y) = input(X/Z, Y/Z) with (X,Y,Z) = m (x,y,1), m row-major; U8 (channels 1) or RGBX (channels 4, 4-byte
// aligned); bilinear or nearest neighbour; samples outside the input take `border`
struct WarpPerspectiveParams {
    ImageView src;
    uint8_t *dst;
    uint32_t dst_pitch, dst_width, dst_height;
    uint32_t channels;
    float m[9];
    bool bilinear;
    uint32_t border;
};
hipError_t launch_warp_perspective(const WarpPerspectiveParams &p, hipStream_t stream);

struct CornerTrackParams {
    Detector detector;
    ImageView image;
    const uint8_t *mask;          // may be null
    uint32_t mask_pitch;
    ItemKind item;
    const void *tracked;          // may be null
    const uint32_t *tracked_count;
    uint32_t tracked_capacity;
    void *out;
    uint32_t *out_count;
    uint32_t out_capacity;
    size_t *num_corners;          // may be null (device pointer)
    uint32_t *out_count_host = nullptr;     // optional mirrors in device-visible pinned host memory: the
    uint64_t *num_corners_host = nullptr;   // merge kernel stores the counts there too (no read-back copy)
    float harris_k;
    float threshold;              // Harris threshold, or (float)FAST threshold
    uint32_t fast_n;
    uint32_t cell_size;
    void *scratch;                // corner_track_scratch_bytes() bytes, 8-byte aligned
};

size_t corner_track_scratch_bytes(uint32_t width, uint32_t height, uint32_t cell_size);
// cell pass (one key per detector cell into p.scratch; reads the image and the mask only) + merge pass (tracked points and
// keys -> output array); launch_corner_track = both, one after the other
hipError_t launch_corner_cells(const CornerTrackParams &p, hipStream_t stream);
hipError_t launch_corner_merge(const CornerTrackParams &p, hipStream_t stream);
hipError_t launch_corner_track(const CornerTrackParams &p, hipStream_t stream);

struct PyrLKParams {
    PyramidView prev, curr;
    ItemKind item;
    const void *prev_pts;
    const uint32_t *prev_count;
    const void *estimates;        // may alias out
    void *out;
    uint32_t *out_count;          // may alias prev_count's sibling; written with the item count
    uint32_t *out_count_host = nullptr;   // optional mirror in device-visible pinned host memory
    uint32_t capacity;
    int use_epsilon;
    uint32_t max_iterations;
    float epsilon;
    uint32_t window;
    bool calc_error = false;      // KEYPOINTF: error = mean |I - J| over the level-0 window at the final position
};

// one kernel (pyrlk.hip): aligned dword staging when every level is 4-byte aligned, per-byte clamped loads
// otherwise; both produce identical results
hipError_t launch_pyrlk(const PyrLKParams &p, hipStream_t stream);

// PyrLK and the CELL pass of the detector node (Harris or FAST) that follows it in ONE launch (one wave per feature, then one
// wave per rectangle of cells, filling the compute units the last PyrLK waves leave idle): the cell pass reads only the
// image, so only launch_corner_merge(ct) has to follow.  Results are those of launch_pyrlk + launch_corner_cells.
// pyrlk_cells_fusable(): 4-byte aligned pyramids (and FAST-9 .. 12).
bool pyrlk_cells_fusable(const PyrLKParams &lk, const CornerTrackParams &ct);
hipError_t launch_pyrlk_with_cells(const PyrLKParams &lk, const CornerTrackParams &ct, hipStream_t stream);

// nvxFindHomography (RANSAC) on POINT2F correspondences (homography.hip; semantics: oracle/vxoracle_homography.c)
struct HomographyParams {
    const void *src, *dst;                // POINT2F arrays
    const uint32_t *src_count, *dst_count;
    uint32_t src_capacity, dst_capacity;
    float threshold;
    uint32_t hypotheses;                  // homography_hypotheses(confidence, outlier_ratio, max_estimate_iters)
    uint32_t max_refine_iters;
    float *matrix;                        // 9 floats, element [3*c + r] = H[r][c]
    float *matrix_host = nullptr;         // optional mirror in device-visible pinned host memory
    uint8_t *mask;                        // may be null; one byte per correspondence, 255 = inlier
    uint32_t *mask_count;                 // item count of the mask array (device), may be null iff mask is
    uint32_t *mask_count_host = nullptr;
    uint32_t mask_capacity;
    void *scratch;                        // homography_scratch_bytes(hypotheses) bytes, 16-byte aligned; its first 16 bytes
                                          // (the ticket word) must be zero before the launch -- the kernel leaves them zero,
                                          // so a buffer used by nothing else needs clearing once
};
uint32_t homography_hypotheses(float confidence, float outlier_ratio, uint32_t max_estimate_iters);
size_t homography_scratch_bytes(uint32_t hypotheses);
hipError_t launch_find_homography(const HomographyParams &p, hipStream_t stream);
// FindHomography and the MERGE pass of a detector node (whose cell pass has already run, e.g. inside the PyrLK launch) in one launch:
// both read PyrLK's output, neither reads what the other writes.  Results are those of launch_find_homography + launch_corner_merge.
bool find_homography_merge_fusable(const HomographyParams &p, const CornerTrackParams &ct);
hipError_t launch_find_homography_with_merge(const HomographyParams &p, const CornerTrackParams &ct, hipStream_t stream);

}  // namespace vsa
