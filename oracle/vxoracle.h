/*
 * vxoracle.h -- CPU oracle for the feature-tracking front-end (TEST INFRASTRUCTURE ONLY).
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped library (libvisualslam_amd.so)
 * never links or calls anything declared here.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in NVIDIA's closed libvisionworks 1.6
 * (absent from /root/reference) and the reference holds no golden
 * vector or test for it.  This file restates (i) the behaviour the reference *callers* pin,
 * (ii) Khronos OpenVX 1.1 semantics, (iii) frozen policies [P] documented in DESIGN.md.
 * Each function cites the reference call site it stands in for (paths relative to
 * /root/reference/VisionWorks/VisionWorks-1.6-Samples/).
 */
#ifndef VXORACLE_H
#define VXORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layouts pinned by nvxio/src/OpenGLShaders.hpp:567-597 */
typedef struct { float x, y; } orc_point2f_t;
typedef struct {
    float x, y, strength, scale, orientation;
    int32_t tracking_status;
    float error;
} orc_keypointf_t;

enum { ORC_ITEM_POINT2F = 0, ORC_ITEM_KEYPOINTF = 1 };
enum { ORC_TERM_ITERATIONS = 0, ORC_TERM_EPSILON = 1, ORC_TERM_BOTH = 2 };
enum { ORC_SUM_WAVE64 = 0, /* 64 interleaved partial sums + pairwise tree (canonical)   */
       ORC_SUM_F64SEQ = 1  /* raster-order float64 accumulation (independent cross-check) */ };

typedef struct {
    const uint8_t *data;
    uint32_t pitch, width, height;
} orc_image_t;

/* demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:164-176 (float ceil sizing) */
void orc_pyramid_dims(uint32_t width, uint32_t height, uint32_t levels, float scale,
                      uint32_t *widths, uint32_t *heights);

/* vxColorConvertNode RGBX->U8: demos/feature_tracker/feature_tracker.cpp:322,386 */
void orc_color_convert_rgbx_u8(const uint8_t *src, uint32_t src_pitch,
                               uint8_t *dst, uint32_t dst_pitch,
                               uint32_t width, uint32_t height);

/* one HALF-scale step of vxGaussianPyramidNode: demos/feature_tracker/feature_tracker.cpp:323,393 */
void orc_gaussian_half(const uint8_t *src, uint32_t src_pitch, uint32_t sw, uint32_t sh,
                       uint8_t *dst, uint32_t dst_pitch, uint32_t dw, uint32_t dh);

/* Harris response image (float, -inf where undefined): stage (i) of nvxHarrisTrackNode,
 * demos/feature_tracker/feature_tracker.cpp:418-422 */
void orc_harris_response(const uint8_t *img, uint32_t pitch, uint32_t w, uint32_t h,
                         float k, float *resp);

/* FAST-N strength image (int32, -1 where not a corner for any t>=0): stage (i) of
 * nvxFastTrackNode, demos/feature_tracker/feature_tracker.cpp:426-430 */
void orc_fast_strength(const uint8_t *img, uint32_t pitch, uint32_t w, uint32_t h,
                       uint32_t type_n, int32_t *strength);

/* nvxHarrisTrackNode / nvxuHarrisTrack / nvxcuHarrisTrack.  tracked may be NULL
 * (feature_tracker.cpp:326-330), mask may be NULL.  Returns the number of items written
 * to out (<= capacity); *num_corners (optional) = count before truncation. */
uint32_t orc_harris_track(const uint8_t *img, uint32_t pitch, uint32_t w, uint32_t h,
                          const uint8_t *mask, uint32_t mask_pitch,
                          int item_type, const void *tracked, uint32_t n_tracked,
                          float k, float threshold, uint32_t cell_size,
                          void *out, uint32_t capacity, size_t *num_corners,
                          float *strength_out /* optional, capacity floats */);

/* nvxFastTrackNode / nvxuFastTrack / nvxcuFastTrack */
uint32_t orc_fast_track(const uint8_t *img, uint32_t pitch, uint32_t w, uint32_t h,
                        const uint8_t *mask, uint32_t mask_pitch,
                        int item_type, const void *tracked, uint32_t n_tracked,
                        uint32_t type_n, uint32_t threshold, uint32_t cell_size,
                        void *out, uint32_t capacity, size_t *num_corners,
                        float *strength_out);

/* vxOpticalFlowPyrLKNode / nvxcuOpticalFlowPyrLK: demos/feature_tracker/feature_tracker.cpp:402-412,
 * demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:297-301.  pts_inout holds the initial
 * estimates on entry (feature_tracker_nvxcu.cpp:288-295) and the tracked positions on return;
 * item count and order are preserved (object_tracker_keypoint_nvxcu.cpp:1862-1863). */
void orc_optical_flow_pyr_lk(const orc_image_t *prev_levels, const orc_image_t *curr_levels,
                             uint32_t num_levels,
                             int item_type, const void *prev_pts, void *pts_inout, uint32_t n,
                             int termination, float epsilon, uint32_t num_iterations,
                             uint32_t window, int sum_mode, int calc_error);

/* nvxFindHomographyNode (RANSAC): demos/video_stabilizer/stabilizer.cpp:264-273, SfM.cpp:1576-1578.
 * src/dst: n x {f32 x,y}; H_out: 9 floats, element [3*c + r] = H[r][c] (homography_filter_node.cpp:87-91);
 * mask: n bytes, 255 = inlier.  Returns the inlier count.  Policies: vxoracle_homography.c. */
uint32_t orc_find_homography(const float *src, const float *dst, uint32_t n, float threshold,
                             uint32_t max_estimate_iters, uint32_t max_refine_iters, float confidence,
                             float outlier_ratio, float *H_out, uint8_t *mask);
uint32_t orc_homography_hypotheses(float confidence, float outlier_ratio, uint32_t max_estimate_iters);

/* vxWarpPerspectiveNode: demos/video_stabilizer/stabilizer.cpp:292-295.  m: 9 floats, row-major M of
 * (x0,y0,z0) = M (x,y,1); channels 1 (U8) or 4 (RGBX); border: `channels` bytes.  Policies: vxoracle_homography.c. */
void orc_warp_perspective(const uint8_t *src, uint32_t src_pitch, uint32_t sw, uint32_t sh, uint32_t channels,
                          uint8_t *dst, uint32_t dst_pitch, uint32_t dw, uint32_t dh,
                          const float *m, int bilinear, const uint8_t *border);

#ifdef __cplusplus
}
#endif
#endif
