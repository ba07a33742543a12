/*
 * NVX/nvxcu.h -- "thin layer" C ABI of libvisualslam_amd: stateless primitives on caller-owned
 * device buffers, enqueued on a caller-owned HIP stream.
 *
 * This header is written from scratch for MI355X (HIP, gfx950).  It keeps the names, argument
 * order and struct shapes the VisionWorks samples use for this path, so a maintainer of those
 * samples swaps `cuda*` runtime calls for `hip*` and relinks; each declaration cites the
 * reference call site it replaces (paths relative to
 * /root/reference/VisionWorks/VisionWorks-1.6-Samples/).  The reference never ships the original
 * header (closed libvisionworks), so numeric enum values are ours; the only coupling the callers
 * rely on -- image formats are FOURCCs and nvxcu enums equal the vx enums
 * (nvxio/src/OVX/Private/TypesOVX.cpp:56,97) -- is kept.
 *
 * All primitives are asynchronous on exec_target->stream; the caller synchronises once per
 * frame (demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:320).
 */
#ifndef NVX_NVXCU_H
#define NVX_NVXCU_H

#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVXCU_DF_IMAGE(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))

/* nvxio/include/NVX/Utility.hpp:84-92 (NVXCU_SAFE_CALL compares against NVXCU_SUCCESS) */
typedef enum nvxcu_error_status_e {
    NVXCU_SUCCESS = 0,
    NVXCU_FAILURE = -1,
    NVXCU_ERROR_NOT_IMPLEMENTED = -2,
    NVXCU_ERROR_NOT_SUPPORTED = -3,
    NVXCU_ERROR_NO_MEMORY = -8,
    NVXCU_ERROR_INVALID_PARAMETERS = -10,
    NVXCU_ERROR_INVALID_FORMAT = -14,
    NVXCU_ERROR_INVALID_DIMENSION = -15,
    NVXCU_ERROR_INVALID_TYPE = -17
} nvxcu_error_status_e;

/* FOURCC values shared with vx_df_image (nvxio/src/NVX/Render/EventLogger.cpp:220-222) */
typedef enum nvxcu_df_image_e {
    NVXCU_DF_IMAGE_NONE = NVXCU_DF_IMAGE('V', 'I', 'R', 'T'),
    NVXCU_DF_IMAGE_U8   = NVXCU_DF_IMAGE('U', '0', '0', '8'),
    NVXCU_DF_IMAGE_RGB  = NVXCU_DF_IMAGE('R', 'G', 'B', '2'),
    NVXCU_DF_IMAGE_RGBX = NVXCU_DF_IMAGE('R', 'G', 'B', 'A'),
    NVXCU_DF_IMAGE_NV12 = NVXCU_DF_IMAGE('N', 'V', '1', '2'),
    NVXCU_DF_IMAGE_2F32 = NVXCU_DF_IMAGE('2', 'F', '3', '2')
} nvxcu_df_image_e;

/* numeric values shared with vx_type_e / nvx_type_e (TypesOVX.cpp:97) */
typedef enum nvxcu_array_item_type_e {
    NVXCU_TYPE_RECTANGLE     = 0x020,
    NVXCU_TYPE_KEYPOINT      = 0x021,
    NVXCU_TYPE_COORDINATES2D = 0x022,
    NVXCU_TYPE_COORDINATES3D = 0x023,
    NVXCU_TYPE_POINT2F       = 0x400,
    NVXCU_TYPE_POINT3F       = 0x401,
    NVXCU_TYPE_POINT4F       = 0x402,
    NVXCU_TYPE_KEYPOINTF     = 0x403
} nvxcu_array_item_type_e;

/* item layouts: nvxio/src/OpenGLShaders.hpp:567-597, nvxio/src/NVX/Private/Types.cpp:39-60 */
typedef struct nvxcu_point2f_t { float x, y; } nvxcu_point2f_t;
typedef struct nvxcu_point3f_t { float x, y, z; } nvxcu_point3f_t;
typedef struct nvxcu_point4f_t { float x, y, z, w; } nvxcu_point4f_t;
typedef struct nvxcu_keypointf_t {
    float x, y, strength, scale, orientation;
    int32_t tracking_status;            /* 255 = tracked (object_tracker_keypoint_nvxcu.cpp:1887), 0 = lost */
    float error;
} nvxcu_keypointf_t;
typedef struct nvxcu_keypoint_t {
    int32_t x, y;
    float strength, scale, orientation;
    int32_t tracking_status;
    float error;
} nvxcu_keypoint_t;
typedef struct nvxcu_rectangle_t { uint32_t start_x, start_y, end_x, end_y; } nvxcu_rectangle_t;
typedef struct nvxcu_coordinates2d_t { uint32_t x, y; } nvxcu_coordinates2d_t;
typedef struct nvxcu_coordinates3d_t { uint32_t x, y, z; } nvxcu_coordinates3d_t;

/* images: feature_tracker_nvxcu.cpp:135-141 */
typedef enum nvxcu_image_type_e { NVXCU_PITCH_LINEAR_IMAGE = 0 } nvxcu_image_type_e;
typedef struct nvxcu_image_t {
    nvxcu_image_type_e image_type;
    nvxcu_df_image_e format;
    uint32_t width, height;
} nvxcu_image_t;
#define NVXCU_MAX_PLANES 4
typedef struct nvxcu_image_plane_t { void *dev_ptr; uint32_t pitch_in_bytes; } nvxcu_image_plane_t;
typedef struct nvxcu_pitch_linear_image_t {
    nvxcu_image_t base;
    nvxcu_image_plane_t planes[NVXCU_MAX_PLANES];
} nvxcu_pitch_linear_image_t;

/* pyramids: feature_tracker_nvxcu.cpp:157-176 (level i is ceil(base * scale^i), scale is a float) */
typedef enum nvxcu_pyramid_type_e { NVXCU_PITCH_LINEAR_PYRAMID = 0 } nvxcu_pyramid_type_e;
#define NVXCU_SCALE_PYRAMID_HALF 0.5f
#define NVXCU_PYRAMID_MAX_LEVELS 10
typedef struct nvxcu_pyramid_t {
    nvxcu_pyramid_type_e pyramid_type;
    uint32_t num_levels;
    float scale;
} nvxcu_pyramid_t;
typedef struct nvxcu_pitch_linear_pyramid_t {
    nvxcu_pyramid_t base;
    nvxcu_pitch_linear_image_t *levels;
} nvxcu_pitch_linear_pyramid_t;

/* arrays: feature_tracker_nvxcu.cpp:111-116 (payload + device-resident u32 item count) */
typedef enum nvxcu_array_type_e { NVXCU_PLAIN_ARRAY = 0 } nvxcu_array_type_e;
typedef struct nvxcu_array_t {
    nvxcu_array_type_e array_type;
    nvxcu_array_item_type_e item_type;
    uint32_t capacity;
} nvxcu_array_t;
typedef struct nvxcu_plain_array_t {
    nvxcu_array_t base;
    void *dev_ptr;
    uint32_t *num_items_dev_ptr;
} nvxcu_plain_array_t;

/* execution target: feature_tracker_nvxcu.cpp:213-215 (a HIP stream instead of a CUDA stream) */
typedef enum nvxcu_exec_target_type_e { NVXCU_STREAM_EXEC_TARGET = 0 } nvxcu_exec_target_type_e;
typedef struct nvxcu_exec_target_t { nvxcu_exec_target_type_e exec_target_type; } nvxcu_exec_target_t;
typedef struct nvxcu_stream_exec_target_t {
    nvxcu_exec_target_t base;
    hipStream_t stream;
    hipDeviceProp_t dev_prop;
} nvxcu_stream_exec_target_t;

/* border: feature_tracker_nvxcu.cpp:219-220, object_tracker_keypoint_nvxcu.cpp:1639 */
typedef enum nvxcu_border_mode_e {
    NVXCU_BORDER_MODE_UNDEFINED = 0,
    NVXCU_BORDER_MODE_CONSTANT = 1,
    NVXCU_BORDER_MODE_REPLICATE = 2
} nvxcu_border_mode_e;
typedef struct nvxcu_border_t {
    nvxcu_border_mode_e mode;
    union { uint8_t U8; int16_t S16; uint32_t U32; float F32; } constant_value;
} nvxcu_border_t;

/* scratch protocol: feature_tracker_nvxcu.cpp:413-449 */
typedef struct nvxcu_tmp_buf_size_t { size_t dev_buf_size, host_buf_size; } nvxcu_tmp_buf_size_t;
typedef struct nvxcu_tmp_buf_t { void *dev_ptr, *host_ptr; } nvxcu_tmp_buf_t;

typedef enum nvxcu_color_space_e { NVXCU_COLOR_SPACE_NONE = 0, NVXCU_COLOR_SPACE_BT601_525 = 1,
                                   NVXCU_COLOR_SPACE_BT601_625 = 2, NVXCU_COLOR_SPACE_BT709 = 3,
                                   NVXCU_COLOR_SPACE_DEFAULT = NVXCU_COLOR_SPACE_BT709 } nvxcu_color_space_e;
typedef enum nvxcu_channel_range_e { NVXCU_CHANNEL_RANGE_FULL = 0, NVXCU_CHANNEL_RANGE_RESTRICTED = 1 } nvxcu_channel_range_e;
typedef enum nvxcu_term_criteria_e { NVXCU_TERM_CRITERIA_ITERATIONS = 0, NVXCU_TERM_CRITERIA_EPSILON = 1,
                                     NVXCU_TERM_CRITERIA_BOTH = 2 } nvxcu_term_criteria_e;

/* ---- primitives ------------------------------------------------------------------------- */

/* replaces nvxcuColorConvert -- feature_tracker_nvxcu.cpp:279-282, 461-464.
 * RGBX -> U8 (BT.709 luma, 8-bit fixed point) and U8 -> U8 copy. */
nvxcu_error_status_e nvxcuColorConvert(const nvxcu_image_t *input, const nvxcu_image_t *output,
                                       nvxcu_color_space_e color_space, nvxcu_channel_range_e channel_range,
                                       const nvxcu_exec_target_t *exec_target);

/* replaces nvxcuGaussianPyramid_GetBufSize -- feature_tracker_nvxcu.cpp:413-415 (returns {0,0}: no scratch) */
nvxcu_tmp_buf_size_t nvxcuGaussianPyramid_GetBufSize(uint32_t width, uint32_t height, uint32_t num_levels,
                                                     float scale, const nvxcu_border_t *border,
                                                     const hipDeviceProp_t *dev_prop);
/* replaces nvxcuGaussianPyramid -- feature_tracker_nvxcu.cpp:284-285, 466.
 * Level 0 is the input and is left untouched; levels 1..n-1 are written. */
nvxcu_error_status_e nvxcuGaussianPyramid(const nvxcu_pyramid_t *pyramid, const nvxcu_tmp_buf_t *tmp_buf,
                                          const nvxcu_border_t *border, const nvxcu_exec_target_t *exec_target);

/* replaces nvxcuHarrisTrack_GetBufSize -- feature_tracker_nvxcu.cpp:431-435 */
nvxcu_tmp_buf_size_t nvxcuHarrisTrack_GetBufSize(uint32_t width, uint32_t height, nvxcu_df_image_e format,
                                                 nvxcu_array_item_type_e item_type, uint32_t capacity,
                                                 uint32_t cell_size, const hipDeviceProp_t *dev_prop);
/* replaces nvxcuHarrisTrack -- feature_tracker_nvxcu.cpp:307-310, 470-473.
 * mask, tracked_points and num_corners_dev_ptr may be NULL. */
nvxcu_error_status_e nvxcuHarrisTrack(const nvxcu_image_t *input, const nvxcu_array_t *output,
                                      const nvxcu_image_t *mask, const nvxcu_array_t *tracked_points,
                                      float k, float threshold, uint32_t cell_size,
                                      size_t *num_corners_dev_ptr, const nvxcu_tmp_buf_t *tmp_buf,
                                      const nvxcu_exec_target_t *exec_target);

/* NO COUNTERPART IN THE REFERENCE.  nvxcuHarrisTrack for callers that cannot allocate or be handed a workspace:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments, results and
 * error codes as nvxcuHarrisTrack minus tmp_buf: the per-cell keys live in a device array inside the library
 * (8 bytes x NVXCU_HARRIS_TRACK_NOALLOC_MAX_CELLS = 512 KiB per GPU, held until the process ends), and the call
 * allocates nothing.  Limits: ceil(width / cell_size) * ceil(height / cell_size) <= NVXCU_HARRIS_TRACK_NOALLOC_MAX_CELLS,
 * else NVXCU_ERROR_NO_MEMORY (1080p and 4K frames at cell 18 use 6 420 and 25 680 cells).  The device that owns
 * exec_target's stream must be the calling thread's current device, else NVXCU_ERROR_INVALID_PARAMETERS (checked with
 * hipStreamGetDevice).  The first call on a GPU looks up the array's address there (hipGetSymbolAddress, which may load
 * the library's code object): make one call on each GPU outside stream capture before capturing calls into a HIP
 * graph; every later call is launches and one table read.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (two streams, two threads) race -- serialise them or use nvxcuHarrisTrack with separate tmp_bufs. */
#define NVXCU_HARRIS_TRACK_NOALLOC_MAX_CELLS 65536u
nvxcu_error_status_e nvxcuHarrisTrack_NoAlloc(const nvxcu_image_t *input, const nvxcu_array_t *output,
                                              const nvxcu_image_t *mask, const nvxcu_array_t *tracked_points,
                                              float k, float threshold, uint32_t cell_size,
                                              size_t *num_corners_dev_ptr, const nvxcu_exec_target_t *exec_target);

/* replaces nvxcuFastTrack_GetBufSize -- feature_tracker_nvxcu.cpp:436-440 */
nvxcu_tmp_buf_size_t nvxcuFastTrack_GetBufSize(uint32_t width, uint32_t height, nvxcu_df_image_e format,
                                               nvxcu_array_item_type_e item_type, uint32_t capacity,
                                               uint32_t cell_size, const hipDeviceProp_t *dev_prop);
/* replaces nvxcuFastTrack -- feature_tracker_nvxcu.cpp:314-317, 477-480 */
nvxcu_error_status_e nvxcuFastTrack(const nvxcu_image_t *input, const nvxcu_array_t *output,
                                    const nvxcu_image_t *mask, const nvxcu_array_t *tracked_points,
                                    uint32_t type, uint32_t threshold, uint32_t cell_size,
                                    size_t *num_corners_dev_ptr, const nvxcu_tmp_buf_t *tmp_buf,
                                    const nvxcu_exec_target_t *exec_target);

/* replaces nvxcuOpticalFlowPyrLK_GetBufSize -- samples/object_tracker_nvxcu/object_tracker_keypoint_nvxcu.cpp:1661-1667
 * (returns {0,0}: the kernel keeps its windows in LDS) */
nvxcu_tmp_buf_size_t nvxcuOpticalFlowPyrLK_GetBufSize(uint32_t width, uint32_t height, uint32_t num_levels,
                                                      float scale, nvxcu_array_item_type_e item_type,
                                                      uint32_t capacity, uint32_t window_dimension,
                                                      uint32_t calc_error, const nvxcu_border_t *border,
                                                      const hipDeviceProp_t *dev_prop);
/* replaces nvxcuOpticalFlowPyrLK -- feature_tracker_nvxcu.cpp:297-301, object_tracker_keypoint_nvxcu.cpp:1791-1828.
 * new_points holds the initial estimates on entry (count + payload, feature_tracker_nvxcu.cpp:288-295)
 * and the tracked positions on return; item count and order equal old_points'.  calc_error != 0 fills
 * nvxcu_keypointf_t::error of tracked items with the mean absolute template difference at the final position
 * (0 otherwise, and for lost items). */
nvxcu_error_status_e nvxcuOpticalFlowPyrLK(const nvxcu_pyramid_t *old_pyramid, const nvxcu_pyramid_t *new_pyramid,
                                           const nvxcu_array_t *old_points, const nvxcu_array_t *new_points,
                                           nvxcu_term_criteria_e termination, float epsilon,
                                           uint32_t num_iterations, uint32_t window_dimension,
                                           uint32_t calc_error, const nvxcu_tmp_buf_t *tmp_buf,
                                           const nvxcu_border_t *border, const nvxcu_exec_target_t *exec_target);

#ifdef __cplusplus
}
#endif
#endif /* NVX_NVXCU_H */
