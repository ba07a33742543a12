// nvxcu_api.cpp -- the thin C ABI (include/NVX/nvxcu.h): argument validation and launches.
// Stateless: every buffer is caller-owned, every launch goes to the caller's HIP stream, nothing
// is allocated or synchronised here (callers may capture these calls into a HIP graph).
#include <hip/hip_runtime.h>

#include <NVX/nvxcu.h>

#include <cmath>
#include <cstring>
#include <mutex>

#include "kernels.hpp"

// ---- the cell-key buffer of nvxcuHarrisTrack_NoAlloc -------------------------------------------------------------
// One 64-bit key per detector cell, in a file-scope device array instead of the caller's tmp_buf.  Every GPU the
// library runs on holds its own copy (512 KiB) at its own address until the process ends.  In host code the array's
// name is only a handle for hipGetSymbolAddress: the address kernels and copies take is the one looked up below,
// once per device, with that device current.
static __device__ __align__(256) unsigned long long scratch[NVXCU_HARRIS_TRACK_NOALLOC_MAX_CELLS];

static constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mu;
static void *g_scratch_ptr[kMaxDevices];      // device address of `scratch` on device i, looked up on first use

// The copy of `scratch` on the GPU that `stream` belongs to, which must be the calling thread's current device (launches go
// to the current device's code object; a pointer into another GPU's copy would fault or silently use peer access):
// *status = NVXCU_ERROR_INVALID_PARAMETERS on a mismatch, NVXCU_FAILURE when a HIP call fails.  The first call on a device
// looks the address up (hipGetSymbolAddress may load the code object: not something to do under stream capture, so the
// header asks for one call outside capture first); later calls only read the table.
static void *static_scratch(hipStream_t stream, nvxcu_error_status_e *status)
{
    int dev = -1;
    hipDevice_t sdev = -1;
    *status = NVXCU_FAILURE;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    if (hipStreamGetDevice(stream, &sdev) != hipSuccess) return nullptr;
    if ((int)sdev != dev) { *status = NVXCU_ERROR_INVALID_PARAMETERS; return nullptr; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return nullptr;
        g_scratch_ptr[dev] = ptr;
    }
    *status = NVXCU_SUCCESS;
    return g_scratch_ptr[dev];
}

namespace {

using namespace vsa;

inline const nvxcu_pitch_linear_image_t *as_pl(const nvxcu_image_t *img)
{
    if (!img || img->image_type != NVXCU_PITCH_LINEAR_IMAGE) return nullptr;
    return reinterpret_cast<const nvxcu_pitch_linear_image_t *>(img);
}

inline const nvxcu_plain_array_t *as_plain(const nvxcu_array_t *arr)
{
    if (!arr || arr->array_type != NVXCU_PLAIN_ARRAY) return nullptr;
    return reinterpret_cast<const nvxcu_plain_array_t *>(arr);
}

inline bool get_stream(const nvxcu_exec_target_t *t, hipStream_t *s)
{
    if (!t || t->exec_target_type != NVXCU_STREAM_EXEC_TARGET) return false;
    *s = reinterpret_cast<const nvxcu_stream_exec_target_t *>(t)->stream;
    return true;
}

inline nvxcu_error_status_e from_hip(hipError_t e) { return e == hipSuccess ? NVXCU_SUCCESS : NVXCU_FAILURE; }

inline bool item_kind(nvxcu_array_item_type_e t, ItemKind *k)
{
    if (t == NVXCU_TYPE_POINT2F) { *k = kPoint2f; return true; }
    if (t == NVXCU_TYPE_KEYPOINTF) { *k = kKeypointf; return true; }
    return false;
}

// level i must be ceil(base * 0.5^i) in float arithmetic (feature_tracker_nvxcu.cpp:164-176)
nvxcu_error_status_e pyramid_view(const nvxcu_pyramid_t *pyr, PyramidView *v)
{
    if (!pyr || pyr->pyramid_type != NVXCU_PITCH_LINEAR_PYRAMID) return NVXCU_ERROR_INVALID_PARAMETERS;
    const auto *p = reinterpret_cast<const nvxcu_pitch_linear_pyramid_t *>(pyr);
    if (pyr->num_levels < 1 || pyr->num_levels > (uint32_t)kMaxPyrLevels || !p->levels)
        return NVXCU_ERROR_INVALID_PARAMETERS;
    if (pyr->scale != NVXCU_SCALE_PYRAMID_HALF) return NVXCU_ERROR_NOT_SUPPORTED;
    const uint32_t w0 = p->levels[0].base.width, h0 = p->levels[0].base.height;
    float scale = 1.0f;
    uint32_t w = w0, h = h0;
    for (uint32_t i = 0; i < pyr->num_levels; ++i) {
        const nvxcu_pitch_linear_image_t &l = p->levels[i];
        if (l.base.image_type != NVXCU_PITCH_LINEAR_IMAGE || l.base.format != NVXCU_DF_IMAGE_U8)
            return NVXCU_ERROR_INVALID_FORMAT;
        if (l.base.width != w || l.base.height != h || w == 0 || h == 0) return NVXCU_ERROR_INVALID_DIMENSION;
        if (!l.planes[0].dev_ptr || l.planes[0].pitch_in_bytes < w) return NVXCU_ERROR_INVALID_PARAMETERS;
        v->ptr[i] = static_cast<const uint8_t *>(l.planes[0].dev_ptr);
        v->pitch[i] = l.planes[0].pitch_in_bytes;
        v->width[i] = w;
        v->height[i] = h;
        scale *= pyr->scale;
        w = (uint32_t)ceilf((float)w0 * scale);
        h = (uint32_t)ceilf((float)h0 * scale);
    }
    v->levels = pyr->num_levels;
    return NVXCU_SUCCESS;
}

nvxcu_error_status_e corner_track(Detector det, const nvxcu_image_t *input, const nvxcu_array_t *output,
                                  const nvxcu_image_t *mask, const nvxcu_array_t *tracked,
                                  float k, float threshold, uint32_t fast_n, uint32_t cell_size,
                                  size_t *num_corners, const nvxcu_tmp_buf_t *tmp,
                                  const nvxcu_exec_target_t *exec_target, bool static_keys = false)
{
    hipStream_t stream;
    const auto *in = as_pl(input);
    const auto *out = as_plain(output);
    if (!in || !out || !get_stream(exec_target, &stream)) return NVXCU_ERROR_INVALID_PARAMETERS;
    if (in->base.format != NVXCU_DF_IMAGE_U8) return NVXCU_ERROR_INVALID_FORMAT;
    if (in->base.width == 0 || in->base.height == 0 || cell_size == 0) return NVXCU_ERROR_INVALID_PARAMETERS;
    if ((uint64_t)in->base.width * in->base.height >= (1ull << 32)) return NVXCU_ERROR_INVALID_DIMENSION;
    if (!in->planes[0].dev_ptr || !out->dev_ptr || !out->num_items_dev_ptr) return NVXCU_ERROR_INVALID_PARAMETERS;

    CornerTrackParams p{};
    if (!item_kind(out->base.item_type, &p.item)) return NVXCU_ERROR_INVALID_TYPE;
    p.detector = det;
    p.image = { static_cast<const uint8_t *>(in->planes[0].dev_ptr), in->planes[0].pitch_in_bytes,
                in->base.width, in->base.height };
    if (mask) {
        const auto *m = as_pl(mask);
        if (!m || m->base.format != NVXCU_DF_IMAGE_U8) return NVXCU_ERROR_INVALID_FORMAT;
        if (m->base.width != in->base.width || m->base.height != in->base.height) return NVXCU_ERROR_INVALID_DIMENSION;
        if (!m->planes[0].dev_ptr) return NVXCU_ERROR_INVALID_PARAMETERS;
        p.mask = static_cast<const uint8_t *>(m->planes[0].dev_ptr);
        p.mask_pitch = m->planes[0].pitch_in_bytes;
    }
    if (tracked) {
        const auto *t = as_plain(tracked);
        if (!t || t->base.item_type != out->base.item_type) return NVXCU_ERROR_INVALID_TYPE;
        if (!t->dev_ptr || !t->num_items_dev_ptr || t->dev_ptr == out->dev_ptr) return NVXCU_ERROR_INVALID_PARAMETERS;
        p.tracked = t->dev_ptr;
        p.tracked_count = t->num_items_dev_ptr;
        p.tracked_capacity = t->base.capacity;
    }
    if (det == kFast && (fast_n < 9 || fast_n > 12 || threshold >= 255.0f)) return NVXCU_ERROR_INVALID_PARAMETERS;
    const size_t need = corner_track_scratch_bytes(in->base.width, in->base.height, cell_size);
    void *keys = nullptr;
    if (static_keys) {
        // the keys of every cell are rewritten by each call (k_harris_cells stores all of them), so the array needs
        // no clearing between calls although it keeps the previous call's contents
        if (need > sizeof(unsigned long long) * NVXCU_HARRIS_TRACK_NOALLOC_MAX_CELLS) return NVXCU_ERROR_NO_MEMORY;
        nvxcu_error_status_e st = NVXCU_FAILURE;
        keys = static_scratch(stream, &st);
        if (!keys) return st;
    } else {
        if (!tmp || !tmp->dev_ptr || ((uintptr_t)tmp->dev_ptr & 7u)) return need ? NVXCU_ERROR_INVALID_PARAMETERS : NVXCU_FAILURE;
        keys = tmp->dev_ptr;
    }
    p.out = out->dev_ptr;
    p.out_count = out->num_items_dev_ptr;
    p.out_capacity = out->base.capacity;
    p.num_corners = num_corners;
    p.harris_k = k;
    p.threshold = threshold;
    p.fast_n = fast_n;
    p.cell_size = cell_size;
    p.scratch = keys;
    return from_hip(launch_corner_track(p, stream));
}

}  // namespace

extern "C" {

nvxcu_error_status_e nvxcuColorConvert(const nvxcu_image_t *input, const nvxcu_image_t *output,
                                       nvxcu_color_space_e color_space, nvxcu_channel_range_e channel_range,
                                       const nvxcu_exec_target_t *exec_target)
{
    hipStream_t stream;
    const auto *in = as_pl(input);
    const auto *out = as_pl(output);
    if (!in || !out || !get_stream(exec_target, &stream)) return NVXCU_ERROR_INVALID_PARAMETERS;
    if (in->base.width != out->base.width || in->base.height != out->base.height) return NVXCU_ERROR_INVALID_DIMENSION;
    if (!in->planes[0].dev_ptr || !out->planes[0].dev_ptr) return NVXCU_ERROR_INVALID_PARAMETERS;
    if (out->base.format != NVXCU_DF_IMAGE_U8) return NVXCU_ERROR_INVALID_FORMAT;
    if (in->base.format == NVXCU_DF_IMAGE_RGBX) {
        if (color_space != NVXCU_COLOR_SPACE_BT709 && color_space != NVXCU_COLOR_SPACE_NONE) return NVXCU_ERROR_NOT_SUPPORTED;
        if (channel_range != NVXCU_CHANNEL_RANGE_FULL) return NVXCU_ERROR_NOT_SUPPORTED;
        if (in->planes[0].pitch_in_bytes < 4u * in->base.width || out->planes[0].pitch_in_bytes < out->base.width)
            return NVXCU_ERROR_INVALID_PARAMETERS;
        const ImageView src{ static_cast<const uint8_t *>(in->planes[0].dev_ptr), in->planes[0].pitch_in_bytes,
                             in->base.width, in->base.height };
        return from_hip(launch_rgbx_to_u8(src, static_cast<uint8_t *>(out->planes[0].dev_ptr),
                                          out->planes[0].pitch_in_bytes, stream));
    }
    if (in->base.format == NVXCU_DF_IMAGE_U8) {
        return from_hip(hipMemcpy2DAsync(out->planes[0].dev_ptr, out->planes[0].pitch_in_bytes,
                                         in->planes[0].dev_ptr, in->planes[0].pitch_in_bytes,
                                         in->base.width, in->base.height, hipMemcpyDeviceToDevice, stream));
    }
    return NVXCU_ERROR_INVALID_FORMAT;
}

nvxcu_tmp_buf_size_t nvxcuGaussianPyramid_GetBufSize(uint32_t, uint32_t, uint32_t, float, const nvxcu_border_t *,
                                                     const hipDeviceProp_t *)
{
    return nvxcu_tmp_buf_size_t{ 0, 0 };
}

nvxcu_error_status_e nvxcuGaussianPyramid(const nvxcu_pyramid_t *pyramid, const nvxcu_tmp_buf_t *,
                                          const nvxcu_border_t *border, const nvxcu_exec_target_t *exec_target)
{
    hipStream_t stream;
    if (!get_stream(exec_target, &stream)) return NVXCU_ERROR_INVALID_PARAMETERS;
    // UNDEFINED is refined to REPLICATE (whole level defined); CONSTANT is not offered
    if (border && border->mode == NVXCU_BORDER_MODE_CONSTANT) return NVXCU_ERROR_NOT_SUPPORTED;
    PyramidView v{};
    nvxcu_error_status_e st = pyramid_view(pyramid, &v);
    if (st != NVXCU_SUCCESS) return st;
    return from_hip(launch_gaussian_pyramid(v, stream));
}

nvxcu_tmp_buf_size_t nvxcuHarrisTrack_GetBufSize(uint32_t width, uint32_t height, nvxcu_df_image_e,
                                                 nvxcu_array_item_type_e, uint32_t, uint32_t cell_size,
                                                 const hipDeviceProp_t *)
{
    return nvxcu_tmp_buf_size_t{ corner_track_scratch_bytes(width, height, cell_size), 0 };
}

nvxcu_error_status_e nvxcuHarrisTrack(const nvxcu_image_t *input, const nvxcu_array_t *output,
                                      const nvxcu_image_t *mask, const nvxcu_array_t *tracked_points,
                                      float k, float threshold, uint32_t cell_size,
                                      size_t *num_corners_dev_ptr, const nvxcu_tmp_buf_t *tmp_buf,
                                      const nvxcu_exec_target_t *exec_target)
{
    return corner_track(kHarris, input, output, mask, tracked_points, k, threshold, 0, cell_size,
                        num_corners_dev_ptr, tmp_buf, exec_target);
}

nvxcu_error_status_e nvxcuHarrisTrack_NoAlloc(const nvxcu_image_t *input, const nvxcu_array_t *output,
                                              const nvxcu_image_t *mask, const nvxcu_array_t *tracked_points,
                                              float k, float threshold, uint32_t cell_size,
                                              size_t *num_corners_dev_ptr, const nvxcu_exec_target_t *exec_target)
{
    return corner_track(kHarris, input, output, mask, tracked_points, k, threshold, 0, cell_size,
                        num_corners_dev_ptr, nullptr, exec_target, /*static_keys=*/true);
}

nvxcu_tmp_buf_size_t nvxcuFastTrack_GetBufSize(uint32_t width, uint32_t height, nvxcu_df_image_e,
                                               nvxcu_array_item_type_e, uint32_t, uint32_t cell_size,
                                               const hipDeviceProp_t *)
{
    return nvxcu_tmp_buf_size_t{ corner_track_scratch_bytes(width, height, cell_size), 0 };
}

nvxcu_error_status_e nvxcuFastTrack(const nvxcu_image_t *input, const nvxcu_array_t *output,
                                    const nvxcu_image_t *mask, const nvxcu_array_t *tracked_points,
                                    uint32_t type, uint32_t threshold, uint32_t cell_size,
                                    size_t *num_corners_dev_ptr, const nvxcu_tmp_buf_t *tmp_buf,
                                    const nvxcu_exec_target_t *exec_target)
{
    return corner_track(kFast, input, output, mask, tracked_points, 0.0f, (float)threshold, type, cell_size,
                        num_corners_dev_ptr, tmp_buf, exec_target);
}

nvxcu_tmp_buf_size_t nvxcuOpticalFlowPyrLK_GetBufSize(uint32_t, uint32_t, uint32_t, float, nvxcu_array_item_type_e,
                                                      uint32_t, uint32_t, uint32_t, const nvxcu_border_t *,
                                                      const hipDeviceProp_t *)
{
    return nvxcu_tmp_buf_size_t{ 0, 0 };
}

nvxcu_error_status_e nvxcuOpticalFlowPyrLK(const nvxcu_pyramid_t *old_pyramid, const nvxcu_pyramid_t *new_pyramid,
                                           const nvxcu_array_t *old_points, const nvxcu_array_t *new_points,
                                           nvxcu_term_criteria_e termination, float epsilon,
                                           uint32_t num_iterations, uint32_t window_dimension,
                                           uint32_t calc_error, const nvxcu_tmp_buf_t *,
                                           const nvxcu_border_t *border, const nvxcu_exec_target_t *exec_target)
{
    hipStream_t stream;
    if (!get_stream(exec_target, &stream)) return NVXCU_ERROR_INVALID_PARAMETERS;
    if (border && border->mode == NVXCU_BORDER_MODE_CONSTANT) return NVXCU_ERROR_NOT_SUPPORTED;
    if (window_dimension < (uint32_t)kLkMinWin || window_dimension > (uint32_t)kLkMaxWin)
        return NVXCU_ERROR_INVALID_PARAMETERS;
    if (termination != NVXCU_TERM_CRITERIA_ITERATIONS && termination != NVXCU_TERM_CRITERIA_EPSILON &&
        termination != NVXCU_TERM_CRITERIA_BOTH)
        return NVXCU_ERROR_INVALID_PARAMETERS;

    PyrLKParams p{};
    nvxcu_error_status_e st = pyramid_view(old_pyramid, &p.prev);
    if (st != NVXCU_SUCCESS) return st;
    st = pyramid_view(new_pyramid, &p.curr);
    if (st != NVXCU_SUCCESS) return st;
    if (p.prev.levels != p.curr.levels || p.prev.width[0] != p.curr.width[0] || p.prev.height[0] != p.curr.height[0])
        return NVXCU_ERROR_INVALID_DIMENSION;

    const auto *op = as_plain(old_points);
    const auto *np = as_plain(new_points);
    if (!op || !np || !op->dev_ptr || !np->dev_ptr || !op->num_items_dev_ptr || !np->num_items_dev_ptr)
        return NVXCU_ERROR_INVALID_PARAMETERS;
    if (op->base.item_type != np->base.item_type || !item_kind(op->base.item_type, &p.item)) return NVXCU_ERROR_INVALID_TYPE;
    if (op->dev_ptr == np->dev_ptr) return NVXCU_ERROR_INVALID_PARAMETERS;

    p.prev_pts = op->dev_ptr;
    p.prev_count = op->num_items_dev_ptr;
    p.estimates = np->dev_ptr;
    p.out = np->dev_ptr;
    p.out_count = np->num_items_dev_ptr;
    p.capacity = op->base.capacity < np->base.capacity ? op->base.capacity : np->base.capacity;
    p.use_epsilon = termination != NVXCU_TERM_CRITERIA_ITERATIONS;
    p.max_iterations = termination == NVXCU_TERM_CRITERIA_EPSILON ? 100u : num_iterations;
    p.epsilon = epsilon;
    p.window = window_dimension;
    p.calc_error = calc_error != 0;      // fills nvxcu_keypointf_t::error; POINT2F items have no such field
    return from_hip(launch_pyrlk(p, stream));
}

/* not part of the reference API: lets language bindings size nvxcu_stream_exec_target_t */
size_t visualslam_amd_sizeof_stream_exec_target(void) { return sizeof(nvxcu_stream_exec_target_t); }
size_t visualslam_amd_offsetof_stream(void) { return offsetof(nvxcu_stream_exec_target_t, stream); }
const char *visualslam_amd_version(void) { return "visualslam_amd 0.1 (gfx950)"; }

}  // extern "C"
