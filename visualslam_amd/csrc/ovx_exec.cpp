// ovx_exec.cpp -- node execution and graph runs: each node kind's launch through kernels.hpp, the graph's enqueue on its HIP stream,
// per-node HIP-event timing, and the pinned mirrors through which counts and matrices reach the host without read-back copies.
// Part of the OpenVX-subset runtime (ovx_internal.hpp).
#include "ovx_internal.hpp"

#include <sched.h>

namespace ovxrt {

using namespace vsa;

// ---- node execution ----------------------------------------------------------------------------

// is `gray` the memory of level 0 of `pyr`: the level object itself, or a whole-image view of it (alias_virtual_pyramid_inputs)
bool is_level0(const _vx_image *gray, const _vx_pyramid *pyr)
{
    if (!gray || !pyr || pyr->levels.empty()) return false;
    const _vx_image *l0 = pyr->levels[0];
    return gray == l0 || (gray->parent == l0 && gray->roi_x == 0 && gray->roi_y == 0 && gray->width == l0->width && gray->height == l0->height);
}

vx_status pyramid_view(_vx_pyramid *p, PyramidView *v)
{
    if (p->levels.empty() || p->levels.size() > (size_t)kMaxPyrLevels) return VX_ERROR_INVALID_PARAMETERS;
    if (p->scale != VX_SCALE_PYRAMID_HALF) return VX_ERROR_NOT_SUPPORTED;
    if (p->format != VX_DF_IMAGE_U8) return VX_ERROR_INVALID_FORMAT;
    vx_status s = ensure_pyramid(p);
    if (s != VX_SUCCESS) return s;
    v->levels = (uint32_t)p->levels.size();
    for (size_t i = 0; i < p->levels.size(); ++i) {
        v->ptr[i] = p->levels[i]->dev;
        v->pitch[i] = p->levels[i]->pitch;
        v->width[i] = p->levels[i]->width;
        v->height[i] = p->levels[i]->height;
    }
    return VX_SUCCESS;
}

vx_status do_color_convert(_vx_image *in, _vx_image *out, hipStream_t stream)
{
    if (!in || !out) return VX_ERROR_INVALID_PARAMETERS;
    if (in->width != out->width || in->height != out->height) return VX_ERROR_INVALID_DIMENSION;
    if (out->format != VX_DF_IMAGE_U8) return VX_ERROR_INVALID_FORMAT;
    vx_status s = ensure_image(in);
    if (s == VX_SUCCESS) s = ensure_image(out);
    if (s != VX_SUCCESS) return s;
    if (in->format == VX_DF_IMAGE_RGBX) return hip_status(launch_rgbx_to_u8(view_of(in), out->dev, out->pitch, stream));
    if (in->format == VX_DF_IMAGE_U8) {
        if (in->dev == out->dev) return VX_SUCCESS;
        return hip_status(hipMemcpy2DAsync(out->dev, out->pitch, in->dev, in->pitch, in->width, in->height,
                                           hipMemcpyDeviceToDevice, stream));
    }
    return VX_ERROR_INVALID_FORMAT;
}

vx_status do_copy_image(_vx_image *in, _vx_image *out, hipStream_t stream)
{
    if (!in || !out) return VX_ERROR_INVALID_PARAMETERS;
    if (in->width != out->width || in->height != out->height) return VX_ERROR_INVALID_DIMENSION;
    if (in->format != out->format) return VX_ERROR_INVALID_FORMAT;
    vx_status s = ensure_image(in);
    if (s == VX_SUCCESS) s = ensure_image(out);
    if (s != VX_SUCCESS) return s;
    return hip_status(hipMemcpy2DAsync(out->dev, out->pitch, in->dev, in->pitch, (size_t)in->width * in->bpp, in->height,
                                       hipMemcpyDeviceToDevice, stream));
}

vx_status do_gaussian_pyramid(_vx_image *in, _vx_pyramid *pyr, hipStream_t stream)
{
    if (!in || !pyr) return VX_ERROR_INVALID_PARAMETERS;
    if (in->format != VX_DF_IMAGE_U8) return VX_ERROR_INVALID_FORMAT;
    if (in->width != pyr->width || in->height != pyr->height) return VX_ERROR_INVALID_DIMENSION;
    PyramidView v{};
    vx_status s = ensure_image(in);
    if (s == VX_SUCCESS) s = pyramid_view(pyr, &v);
    if (s != VX_SUCCESS) return s;
    // level 0 usually IS the input (feature_tracker.cpp:360-362,393): no copy then
    if (in->dev != pyr->levels[0]->dev) {
        s = hip_status(hipMemcpy2DAsync(pyr->levels[0]->dev, pyr->levels[0]->pitch, in->dev, in->pitch, in->width,
                                        in->height, hipMemcpyDeviceToDevice, stream));
        if (s != VX_SUCCESS) return s;
    }
    return hip_status(launch_gaussian_pyramid(v, stream));
}

// ColorConvert(RGBX -> level 0) + GaussianPyramid(level 0 -> pyramid) as one launch when the buffers allow it, as two
// otherwise; the bytes are the same either way
vx_status do_color_pyramid(_vx_image *rgbx, _vx_image *gray, _vx_pyramid *pyr, hipStream_t stream)
{
    if (!rgbx || !gray || !pyr) return VX_ERROR_INVALID_PARAMETERS;
    if (rgbx->format == VX_DF_IMAGE_RGBX && rgbx->width == gray->width && rgbx->height == gray->height && gray->format == VX_DF_IMAGE_U8 &&
        gray->width == pyr->width && gray->height == pyr->height && is_level0(gray, pyr)) {
        PyramidView v{};
        vx_status s = ensure_image(rgbx);
        if (s == VX_SUCCESS) s = pyramid_view(pyr, &v);
        if (s != VX_SUCCESS) return s;
        if (color_pyramid_fusable(view_of(rgbx), v)) return hip_status(launch_color_pyramid(view_of(rgbx), v, stream));
    }
    vx_status s = do_color_convert(rgbx, gray, stream);
    return s == VX_SUCCESS ? do_gaussian_pyramid(gray, pyr, stream) : s;
}

bool item_kind(vx_enum t, ItemKind *k)
{
    if (t == NVX_TYPE_POINT2F) { *k = kPoint2f; return true; }
    if (t == NVX_TYPE_KEYPOINTF) { *k = kKeypointf; return true; }
    return false;
}

vx_status prep_pyrlk(_vx_pyramid *oldp, _vx_pyramid *newp, _vx_array *oldpts, _vx_array *est, _vx_array *out,
                     vx_enum term, _vx_scalar *eps, _vx_scalar *iters, _vx_scalar *use_est, vx_size win, PyrLKParams *pp)
{
    if (!oldp || !newp || !oldpts || !out || !eps || !iters) return VX_ERROR_INVALID_PARAMETERS;
    if (win < (vx_size)kLkMinWin || win > (vx_size)kLkMaxWin) return VX_ERROR_INVALID_VALUE;
    if (term != VX_TERM_CRITERIA_ITERATIONS && term != VX_TERM_CRITERIA_EPSILON && term != VX_TERM_CRITERIA_BOTH)
        return VX_ERROR_INVALID_VALUE;
    if (eps->data_type != VX_TYPE_FLOAT32 || iters->data_type != VX_TYPE_UINT32) return VX_ERROR_INVALID_TYPE;
    PyrLKParams &p = *pp;
    p = PyrLKParams{};
    if (oldpts->item_type != out->item_type || !item_kind(oldpts->item_type, &p.item)) return VX_ERROR_NOT_SUPPORTED;
    if (oldpts == out) return VX_ERROR_INVALID_PARAMETERS;
    const bool with_est = use_est && use_est->v.b != vx_false_e && est;
    const _vx_array *e = with_est ? est : oldpts;
    if (e->item_type != oldpts->item_type) return VX_ERROR_INVALID_TYPE;
    vx_status s = pyramid_view(oldp, &p.prev);
    if (s == VX_SUCCESS) s = pyramid_view(newp, &p.curr);
    if (s == VX_SUCCESS) s = ensure_array(oldpts);
    if (s == VX_SUCCESS) s = ensure_array(out);
    if (s == VX_SUCCESS && with_est) s = ensure_array(est);
    if (s != VX_SUCCESS) return s;
    if (p.prev.levels != p.curr.levels || p.prev.width[0] != p.curr.width[0] || p.prev.height[0] != p.curr.height[0])
        return VX_ERROR_INVALID_DIMENSION;
    p.prev_pts = oldpts->dev;
    p.prev_count = oldpts->dev_count;
    p.estimates = e->dev;
    p.out = out->dev;
    p.out_count = out->dev_count;
    p.out_count_host = out->count_mirror;
    p.capacity = (uint32_t)std::min(std::min(oldpts->capacity, out->capacity), e->capacity);
    p.use_epsilon = term != VX_TERM_CRITERIA_ITERATIONS;
    p.max_iterations = term == VX_TERM_CRITERIA_EPSILON ? 100u : iters->v.u32;
    p.epsilon = eps->v.f32;
    p.window = (uint32_t)win;
    return VX_SUCCESS;
}

vx_status do_pyrlk(_vx_pyramid *oldp, _vx_pyramid *newp, _vx_array *oldpts, _vx_array *est, _vx_array *out,
                   vx_enum term, _vx_scalar *eps, _vx_scalar *iters, _vx_scalar *use_est, vx_size win, hipStream_t stream)
{
    PyrLKParams p{};
    const vx_status s = prep_pyrlk(oldp, newp, oldpts, est, out, term, eps, iters, use_est, win, &p);
    return s != VX_SUCCESS ? s : hip_status(launch_pyrlk(p, stream));
}

vx_status prep_corner_track(Detector det, _vx_image *in, _vx_array *out, _vx_image *mask, _vx_array *tracked,
                            float k, float thr, uint32_t fast_n, uint32_t cell, size_t *dev_num_corners,
                            uint64_t *num_corners_mirror, void *scratch, size_t scratch_size, CornerTrackParams *pp)
{
    if (!in || !out) return VX_ERROR_INVALID_PARAMETERS;
    if (in->format != VX_DF_IMAGE_U8) return VX_ERROR_INVALID_FORMAT;
    if (cell == 0) return VX_ERROR_INVALID_VALUE;
    if (det == kFast && (fast_n < 9 || fast_n > 12 || thr >= 255.0f)) return VX_ERROR_INVALID_VALUE;
    CornerTrackParams &p = *pp;
    p = CornerTrackParams{};
    if (!item_kind(out->item_type, &p.item)) return VX_ERROR_NOT_SUPPORTED;
    vx_status s = ensure_image(in);
    if (s == VX_SUCCESS) s = ensure_array(out);
    if (s == VX_SUCCESS && mask) s = ensure_image(mask);
    if (s == VX_SUCCESS && tracked) s = ensure_array(tracked);
    if (s != VX_SUCCESS) return s;
    if (mask && (mask->format != VX_DF_IMAGE_U8 || mask->width != in->width || mask->height != in->height))
        return VX_ERROR_INVALID_DIMENSION;
    if (tracked && (tracked->item_type != out->item_type || tracked == out)) return VX_ERROR_INVALID_PARAMETERS;
    if (scratch_size < corner_track_scratch_bytes(in->width, in->height, cell)) return VX_ERROR_NO_MEMORY;
    p.detector = det;
    p.image = view_of(in);
    p.mask = mask ? mask->dev : nullptr;
    p.mask_pitch = mask ? mask->pitch : 0;
    p.tracked = tracked ? tracked->dev : nullptr;
    p.tracked_count = tracked ? tracked->dev_count : nullptr;
    p.tracked_capacity = tracked ? (uint32_t)tracked->capacity : 0;
    p.out = out->dev;
    p.out_count = out->dev_count;
    p.out_capacity = (uint32_t)out->capacity;
    p.num_corners = dev_num_corners;
    p.out_count_host = out->count_mirror;
    p.num_corners_host = dev_num_corners ? num_corners_mirror : nullptr;
    p.harris_k = k;
    p.threshold = thr;
    p.fast_n = fast_n;
    p.cell_size = cell;
    p.scratch = scratch;
    return VX_SUCCESS;
}

// merge_only: the cell pass of this execution has already run (inside the PyrLK node's launch)
vx_status do_corner_track(Detector det, _vx_image *in, _vx_array *out, _vx_image *mask, _vx_array *tracked,
                          float k, float thr, uint32_t fast_n, uint32_t cell, size_t *dev_num_corners,
                          uint64_t *num_corners_mirror, void *scratch, size_t scratch_size, hipStream_t stream, bool merge_only)
{
    CornerTrackParams p{};
    const vx_status s = prep_corner_track(det, in, out, mask, tracked, k, thr, fast_n, cell, dev_num_corners, num_corners_mirror,
                                          scratch, scratch_size, &p);
    if (s != VX_SUCCESS) return s;
    return hip_status(merge_only ? launch_corner_merge(p, stream) : launch_corner_track(p, stream));
}

vx_status prep_find_homography(_vx_array *src, _vx_array *dst, _vx_matrix *mat, vx_enum method, float threshold,
                               uint32_t max_estimate, uint32_t max_refine, float confidence, float outlier_ratio,
                               _vx_array *mask, void *scratch, size_t scratch_size, HomographyParams *out)
{
    if (!src || !dst || !mat) return VX_ERROR_INVALID_PARAMETERS;
    if (method != NVX_FIND_HOMOGRAPHY_METHOD_RANSAC) return VX_ERROR_NOT_SUPPORTED;
    if (src->item_type != NVX_TYPE_POINT2F || dst->item_type != NVX_TYPE_POINT2F) return VX_ERROR_INVALID_TYPE;
    if (mat->data_type != VX_TYPE_FLOAT32 || mat->cols != 3 || mat->rows != 3) return VX_ERROR_INVALID_DIMENSION;
    if (mask && mask->item_type != VX_TYPE_UINT8) return VX_ERROR_INVALID_TYPE;
    if (!(threshold > 0.0f)) return VX_ERROR_INVALID_VALUE;
    HomographyParams &p = *out;
    p = HomographyParams{};
    p.hypotheses = homography_hypotheses(confidence, outlier_ratio, max_estimate);
    if (scratch_size < homography_scratch_bytes(p.hypotheses)) return VX_ERROR_NO_MEMORY;
    vx_status s = ensure_array(src);
    if (s == VX_SUCCESS) s = ensure_array(dst);
    if (s == VX_SUCCESS && mask) s = ensure_array(mask);
    if (s == VX_SUCCESS) s = ensure_matrix(mat);
    if (s != VX_SUCCESS) return s;
    p.src = src->dev; p.dst = dst->dev;
    p.src_count = src->dev_count; p.dst_count = dst->dev_count;
    p.src_capacity = (uint32_t)src->capacity; p.dst_capacity = (uint32_t)dst->capacity;
    p.threshold = threshold;
    p.max_refine_iters = max_refine;
    p.matrix = static_cast<float *>(mat->dev);
    p.matrix_host = mat->mirror;
    p.mask = mask ? mask->dev : nullptr;
    p.mask_count = mask ? mask->dev_count : nullptr;
    p.mask_count_host = mask ? mask->count_mirror : nullptr;
    p.mask_capacity = mask ? (uint32_t)mask->capacity : 0;
    p.scratch = scratch;
    return VX_SUCCESS;
}

vx_status do_find_homography(_vx_array *src, _vx_array *dst, _vx_matrix *mat, vx_enum method, float threshold,
                             uint32_t max_estimate, uint32_t max_refine, float confidence, float outlier_ratio,
                             _vx_array *mask, void *scratch, size_t scratch_size, hipStream_t stream)
{
    HomographyParams p{};
    const vx_status s = prep_find_homography(src, dst, mat, method, threshold, max_estimate, max_refine, confidence, outlier_ratio,
                                             mask, scratch, scratch_size, &p);
    return s != VX_SUCCESS ? s : hip_status(launch_find_homography(p, stream));
}

// the matrix as stored is M column by column: data[3c + r] = M[r][c] (truncate_transform_node.cpp:163-165)
vx_status do_warp_perspective(_vx_image *in, _vx_matrix *mat, vx_enum interp, _vx_image *out, const vx_border_t &border,
                              hipStream_t stream)
{
    if (!in || !mat || !out) return VX_ERROR_INVALID_PARAMETERS;
    if (in->format != out->format || (in->format != VX_DF_IMAGE_U8 && in->format != VX_DF_IMAGE_RGBX)) return VX_ERROR_INVALID_FORMAT;
    if (mat->data_type != VX_TYPE_FLOAT32 || mat->cols != 3 || mat->rows != 3) return VX_ERROR_INVALID_DIMENSION;
    if (interp != VX_INTERPOLATION_BILINEAR && interp != VX_INTERPOLATION_NEAREST_NEIGHBOR) return VX_ERROR_NOT_SUPPORTED;
    if (border.mode == VX_BORDER_REPLICATE) return VX_ERROR_NOT_SUPPORTED;
    vx_status s = ensure_image(in);
    if (s == VX_SUCCESS) s = ensure_image(out);
    if (s != VX_SUCCESS) return s;
    WarpPerspectiveParams p{};
    p.src = view_of(in);
    p.dst = out->dev; p.dst_pitch = out->pitch; p.dst_width = out->width; p.dst_height = out->height;
    p.channels = in->bpp;
    const float *d = reinterpret_cast<const float *>(mat->host.data());
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) p.m[3 * r + c] = d[3 * c + r];
    p.bilinear = interp == VX_INTERPOLATION_BILINEAR;
    p.border = 0;
    if (border.mode == VX_BORDER_CONSTANT) {
        if (in->bpp == 4) memcpy(&p.border, border.constant_value.RGBX, 4);
        else p.border = border.constant_value.U8;
    }
    return hip_status(launch_warp_perspective(p, stream));
}


vx_status node_scratch(_vx_node *n)
{
    auto *in = param<_vx_image>(n, 0, VX_TYPE_IMAGE);
    if (!in) return VX_ERROR_INVALID_PARAMETERS;
    const size_t need = corner_track_scratch_bytes(in->width, in->height, scalar_u32(n, 6));
    if (need > n->scratch_size) {
        if (n->ctx->no_device) return (vx_status)NVX_ERROR_NO_GPU;
        if (n->scratch) (void)hipFree(n->scratch);
        n->scratch = nullptr;
        n->scratch_size = 0;
        if (hipMalloc(&n->scratch, need) != hipSuccess) return VX_ERROR_NO_MEMORY;
        n->scratch_size = need;
    }
    if (param<_vx_scalar>(n, 7, VX_TYPE_SCALAR) && !n->dev_num_corners) {
        if (n->ctx->no_device) return (vx_status)NVX_ERROR_NO_GPU;
        if (hipMalloc(reinterpret_cast<void **>(&n->dev_num_corners), sizeof(size_t)) != hipSuccess) return VX_ERROR_NO_MEMORY;
    }
    return VX_SUCCESS;
}

vx_status homography_node_scratch(_vx_node *n, hipStream_t stream)
{
    const size_t need = homography_scratch_bytes(homography_hypotheses(scalar_f32(n, 7), scalar_f32(n, 8), scalar_u32(n, 5)));
    if (need > n->scratch_size) {
        if (n->ctx->no_device) return (vx_status)NVX_ERROR_NO_GPU;
        if (n->scratch) (void)hipFree(n->scratch);
        n->scratch = nullptr;
        n->scratch_size = 0;
        if (hipMalloc(&n->scratch, need) != hipSuccess) return VX_ERROR_NO_MEMORY;
        n->scratch_size = need;
        // the ticket word, cleared on the stream the kernel runs on (a null-stream memset is not ordered against the
        // graph's non-blocking stream); this buffer is the node's alone and the kernel leaves the word zero
        if (hipMemsetAsync(n->scratch, 0, 16, stream) != hipSuccess) return VX_FAILURE;
    }
    return VX_SUCCESS;
}

// The two nodes' parameters as they are NOW (applications rebind them between runs); false: anything keeps the pair from sharing a launch
static bool prep_pair(_vx_node *fh, _vx_node *ht, hipStream_t stream, HomographyParams *hp, CornerTrackParams *ct)
{
    if (!ht->cells_done) return false;      // this run's cell pass did not ride in the PyrLK launch
    if (homography_node_scratch(fh, stream) != VX_SUCCESS || node_scratch(ht) != VX_SUCCESS) return false;
    auto *m = param<_vx_scalar>(fh, 3, VX_TYPE_SCALAR);
    if (prep_find_homography(param<_vx_array>(fh, 0, VX_TYPE_ARRAY), param<_vx_array>(fh, 1, VX_TYPE_ARRAY),
                             param<_vx_matrix>(fh, 2, VX_TYPE_MATRIX), m ? m->v.e : 0, scalar_f32(fh, 4), scalar_u32(fh, 5),
                             scalar_u32(fh, 6), scalar_f32(fh, 7), scalar_f32(fh, 8), param<_vx_array>(fh, 9, VX_TYPE_ARRAY),
                             fh->scratch, fh->scratch_size, hp) != VX_SUCCESS) return false;
    const bool harris = ht->kind == K_HARRIS_TRACK;
    if (prep_corner_track(harris ? kHarris : kFast, param<_vx_image>(ht, 0, VX_TYPE_IMAGE), param<_vx_array>(ht, 1, VX_TYPE_ARRAY),
                          param<_vx_image>(ht, 2, VX_TYPE_IMAGE), param<_vx_array>(ht, 3, VX_TYPE_ARRAY),
                          harris ? scalar_f32(ht, 4) : 0.0f, harris ? scalar_f32(ht, 5) : (float)scalar_u32(ht, 5),
                          harris ? 0u : scalar_u32(ht, 4), scalar_u32(ht, 6), param<_vx_scalar>(ht, 7, VX_TYPE_SCALAR) ? ht->dev_num_corners : nullptr,
                          ht->num_corners_mirror, ht->scratch, ht->scratch_size, ct) != VX_SUCCESS) return false;
    return find_homography_merge_fusable(*hp, *ct);
}

vx_status run_node(_vx_node *n, hipStream_t stream)
{
    if (n->pair) {
        if (!n->pair_leader && n->pair_done) {      // the leader's launch did this node's work
            n->pair_done = false;
            return VX_SUCCESS;
        }
        if (n->pair_leader) {
            _vx_node *fh = n->kind == K_FIND_HOMOGRAPHY ? n : n->pair, *ht = fh == n ? n->pair : n;
            HomographyParams hp{};
            CornerTrackParams ct{};
            ht->pair_done = fh->pair_done = false;
            if (prep_pair(fh, ht, stream, &hp, &ct)) {
                const vx_status s = hip_status(launch_find_homography_with_merge(hp, ct, stream));
                ht->cells_done = false;
                n->pair->pair_done = s == VX_SUCCESS;
                return s;
            }
        }
    }
    switch (n->kind) {
    case K_FIND_HOMOGRAPHY: {
        vx_status s = homography_node_scratch(n, stream);
        if (s != VX_SUCCESS) return s;
        auto *m = param<_vx_scalar>(n, 3, VX_TYPE_SCALAR);
        return do_find_homography(param<_vx_array>(n, 0, VX_TYPE_ARRAY), param<_vx_array>(n, 1, VX_TYPE_ARRAY),
                                  param<_vx_matrix>(n, 2, VX_TYPE_MATRIX), m ? m->v.e : 0, scalar_f32(n, 4), scalar_u32(n, 5),
                                  scalar_u32(n, 6), scalar_f32(n, 7), scalar_f32(n, 8), param<_vx_array>(n, 9, VX_TYPE_ARRAY),
                                  n->scratch, n->scratch_size, stream);
    }
    case K_WARP_PERSPECTIVE: {
        auto *m = param<_vx_matrix>(n, 1, VX_TYPE_MATRIX);
        auto *t = param<_vx_scalar>(n, 2, VX_TYPE_SCALAR);
        if (m && m->mirror_pending && n->graph) {   // written by a kernel of this run: its values reach the host at stream completion
            vx_status s = sync_host_view(n->graph);
            if (s != VX_SUCCESS) return s;
        }
        return do_warp_perspective(param<_vx_image>(n, 0, VX_TYPE_IMAGE), m, t ? t->v.e : 0, param<_vx_image>(n, 3, VX_TYPE_IMAGE),
                                   n->border, stream);
    }
    case K_USER: {
        if (!n->kernel || !n->kernel->func || !n->graph) return VX_ERROR_INVALID_NODE;
        // host code: everything the earlier nodes produced must be complete and visible through the accessors
        vx_status s = sync_host_view(n->graph);
        if (s != VX_SUCCESS) return s;
        return n->kernel->func(n, n->params.data(), (vx_uint32)n->params.size());
    }
    case K_COLOR_CONVERT:
        if (n->fused_away) return VX_SUCCESS;       // the GaussianPyramid node it feeds converts the frame itself
        return do_color_convert(param<_vx_image>(n, 0, VX_TYPE_IMAGE), param<_vx_image>(n, 1, VX_TYPE_IMAGE), stream);
    case K_COPY_IMAGE:
        return do_copy_image(param<_vx_image>(n, 0, VX_TYPE_IMAGE), param<_vx_image>(n, 1, VX_TYPE_IMAGE), stream);
    case K_GAUSSIAN_PYRAMID:
        if (n->fused_cvt)       // the ColorConvert node's CURRENT input: applications rebind it every frame
            return do_color_pyramid(param<_vx_image>(n->fused_cvt, 0, VX_TYPE_IMAGE), param<_vx_image>(n, 0, VX_TYPE_IMAGE),
                                    param<_vx_pyramid>(n, 1, VX_TYPE_PYRAMID), stream);
        return do_gaussian_pyramid(param<_vx_image>(n, 0, VX_TYPE_IMAGE), param<_vx_pyramid>(n, 1, VX_TYPE_PYRAMID), stream);
    case K_PYRLK: {
        PyrLKParams lk{};
        vx_status s = prep_pyrlk(param<_vx_pyramid>(n, 0, VX_TYPE_PYRAMID), param<_vx_pyramid>(n, 1, VX_TYPE_PYRAMID),
                                 param<_vx_array>(n, 2, VX_TYPE_ARRAY), param<_vx_array>(n, 3, VX_TYPE_ARRAY),
                                 param<_vx_array>(n, 4, VX_TYPE_ARRAY), pyrlk_termination(n), param<_vx_scalar>(n, 6, VX_TYPE_SCALAR),
                                 param<_vx_scalar>(n, 7, VX_TYPE_SCALAR), param<_vx_scalar>(n, 8, VX_TYPE_SCALAR), pyrlk_window(n), &lk);
        if (s != VX_SUCCESS) return s;
        if (_vx_node *h = n->fused_cells) {
            // the HarrisTrack / FastTrack node that merges this node's output: its cell pass (which reads the image only) rides in this
            // launch, with that node's CURRENT parameters; whatever keeps it from doing so leaves the node to run in full
            h->cells_done = false;
            CornerTrackParams ct{};
            const bool harris = h->kind == K_HARRIS_TRACK;
            if (node_scratch(h) == VX_SUCCESS &&
                prep_corner_track(harris ? kHarris : kFast, param<_vx_image>(h, 0, VX_TYPE_IMAGE), param<_vx_array>(h, 1, VX_TYPE_ARRAY),
                                  param<_vx_image>(h, 2, VX_TYPE_IMAGE), param<_vx_array>(h, 3, VX_TYPE_ARRAY),
                                  harris ? scalar_f32(h, 4) : 0.0f, harris ? scalar_f32(h, 5) : (float)scalar_u32(h, 5),
                                  harris ? 0u : scalar_u32(h, 4), scalar_u32(h, 6), nullptr, nullptr, h->scratch, h->scratch_size,
                                  &ct) == VX_SUCCESS &&
                pyrlk_cells_fusable(lk, ct)) {
                s = hip_status(launch_pyrlk_with_cells(lk, ct, stream));
                h->cells_done = s == VX_SUCCESS;
                return s;
            }
        }
        return hip_status(launch_pyrlk(lk, stream));
    }
    case K_HARRIS_TRACK:
    case K_FAST_TRACK: {
        vx_status s = node_scratch(n);
        if (s != VX_SUCCESS) return s;
        const bool harris = n->kind == K_HARRIS_TRACK;
        const bool merge_only = n->cells_done;
        n->cells_done = false;
        return do_corner_track(harris ? kHarris : kFast, param<_vx_image>(n, 0, VX_TYPE_IMAGE),
                               param<_vx_array>(n, 1, VX_TYPE_ARRAY), param<_vx_image>(n, 2, VX_TYPE_IMAGE),
                               param<_vx_array>(n, 3, VX_TYPE_ARRAY), harris ? scalar_f32(n, 4) : 0.0f,
                               harris ? scalar_f32(n, 5) : (float)scalar_u32(n, 5), harris ? 0u : scalar_u32(n, 4),
                               scalar_u32(n, 6), param<_vx_scalar>(n, 7, VX_TYPE_SCALAR) ? n->dev_num_corners : nullptr,
                               n->num_corners_mirror, n->scratch, n->scratch_size, stream, merge_only);
    }
    }
    return VX_ERROR_NOT_IMPLEMENTED;
}

void update_perf(vx_perf_t *p, uint64_t ns)
{
    p->tmp = ns;
    p->sum += ns;
    p->num += 1;
    p->avg = p->sum / p->num;
    p->min = (p->num == 1 || ns < p->min) ? ns : p->min;
    p->max = ns > p->max ? ns : p->max;
}

// How a host thread waits for its graph's stream.  hipStreamSynchronize spins on the completion signal without ever giving up
// its core: fine with a core per camera-stream thread, but with more such threads than cores the ones that hold a core spin
// through their whole time slice while the others' finished frames wait milliseconds to be noticed (a 20-frame run of 32 streams
// on 16 cores: threads starting up to 14 ms late, streams finishing 10 ms apart -- tools/exp_r3_cold.sh).
// VISUALSLAM_AMD_YIELD_SYNC=1 polls instead and offers the core to any other runnable thread between two polls.
static hipError_t wait_for_stream(hipStream_t stream)
{
    static const bool yield_sync = [] { const char *e = getenv("VISUALSLAM_AMD_YIELD_SYNC"); return e && e[0] == '1'; }();
    if (!yield_sync) return hipStreamSynchronize(stream);
    hipError_t e;
    while ((e = hipStreamQuery(stream)) == hipErrorNotReady) sched_yield();
    return e;
}

// Drains the graph's stream and brings the host's view (array counts, num_corners scalars, matrix values) up to date
// with what the kernels enqueued so far have written.  Called before every host (user-kernel) node and at the end.
vx_status sync_host_view(_vx_graph *g)
{
    if (wait_for_stream(g->stream) != hipSuccess) { g->state = VX_GRAPH_STATE_ABANDONED; return VX_FAILURE; }
    walk_mirrors(g,
        [](_vx_array *a, uint32_t *slot) {
            if (!a->mirror_pending) return;
            a->num_items = std::min<size_t>(*slot, a->capacity);
            a->mirror_pending = false;
        },
        [](_vx_node *n, uint64_t *slot) {
            if (!n->corners_pending) return;
            n->corners_pending = false;
            uint64_t v;
            memcpy(&v, slot, 8);
            if (auto *s = param<_vx_scalar>(n, 7, VX_TYPE_SCALAR)) {
                if (s->data_type == VX_TYPE_SIZE) s->v.sz = (vx_size)v;
                else if (s->data_type == VX_TYPE_UINT32) s->v.u32 = (vx_uint32)v;
                else if (s->data_type == VX_TYPE_UINT64) s->v.u64 = v;
            }
        },
        [](_vx_matrix *m, float *slot) {
            if (!m->mirror_pending) return;
            memcpy(m->host.data(), slot, m->host.size());
            m->mirror_pending = false;
        });
    return VX_SUCCESS;
}

vx_status enqueue_graph(_vx_graph *g)
{
    _vx_context *ctx = g->ctx;
    if (ctx->no_device) return (vx_status)NVX_ERROR_NO_GPU;
    if (!g->verified) {
        vx_status s = verify_graph(g);
        if (s != VX_SUCCESS) return s;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return VX_FAILURE;
    g->t0 = std::chrono::steady_clock::now();
    walk_mirrors(g, [](_vx_array *a, uint32_t *slot) { a->count_mirror = slot; a->mirror_pending = false; },
                 [](_vx_node *n, uint64_t *slot) { n->num_corners_mirror = slot; n->corners_pending = false; },
                 [](_vx_matrix *m, float *slot) { m->mirror = slot; m->mirror_pending = false; });
    vx_status result = VX_SUCCESS;
    for (_vx_node *n : g->order) {
        n->timed = ctx->perf || g->perf_enabled || n->perf_enabled;
        const bool host_node = n->kind == K_USER;
        std::chrono::steady_clock::time_point h0;
        if (n->timed && !host_node) {
            if (!n->ev0 && (hipEventCreate(&n->ev0) != hipSuccess || hipEventCreate(&n->ev1) != hipSuccess)) { result = VX_FAILURE; break; }
            (void)hipEventRecord(n->ev0, g->stream);
        }
        if (host_node) h0 = std::chrono::steady_clock::now();
        n->last_status = run_node(n, g->stream);
        if (host_node) {
            // a host node is timed on the host clock (its wait for the stream to drain excluded would need a second
            // clock read inside run_node; the callbacks here are microseconds of matrix arithmetic)
            if (n->timed) update_perf(&n->perf, (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - h0).count());
        } else {
            if (n->timed) (void)hipEventRecord(n->ev1, g->stream);
            // what this kernel stores to the pinned slots is valid once the stream has passed it
            for (size_t p = 0; p < n->params.size(); ++p) {
                _vx_reference *r = n->params[p];
                if (!r || n->spec[p].dir == VX_INPUT) continue;
                if (r->type == VX_TYPE_ARRAY) static_cast<_vx_array *>(r)->mirror_pending = true;
                else if (r->type == VX_TYPE_MATRIX) static_cast<_vx_matrix *>(r)->mirror_pending = true;
            }
            if ((n->kind == K_HARRIS_TRACK || n->kind == K_FAST_TRACK) && param<_vx_scalar>(n, 7, VX_TYPE_SCALAR)) n->corners_pending = true;
        }
        if (n->last_status != VX_SUCCESS) {
            char msg[160];
            snprintf(msg, sizeof(msg), "node %d of the graph failed with status %d", (int)n->kind, (int)n->last_status);
            log_msg(ctx, n, n->last_status, msg);
            result = n->last_status;
            break;
        }
    }
    walk_mirrors(g, [](_vx_array *a, uint32_t *) { a->count_mirror = nullptr; }, [](_vx_node *n, uint64_t *) { n->num_corners_mirror = nullptr; },
                 [](_vx_matrix *m, float *) { m->mirror = nullptr; });
    if (result != VX_SUCCESS) {
        (void)sync_host_view(g);
        g->state = VX_GRAPH_STATE_ABANDONED;
        return result;
    }
    g->scheduled = true;
    g->state = VX_GRAPH_STATE_RUNNING;
    return VX_SUCCESS;
}

vx_status finish_graph(_vx_graph *g)
{
    if (!g->scheduled) return VX_FAILURE;
    g->scheduled = false;
    vx_status s = sync_host_view(g);
    if (s != VX_SUCCESS) return s;
    for (_vx_node *n : g->order)
        if (n->timed && n->kind != K_USER) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, n->ev0, n->ev1) == hipSuccess) update_perf(&n->perf, (uint64_t)(ms * 1e6));
        }
    const auto dt = std::chrono::steady_clock::now() - g->t0;
    update_perf(&g->perf, (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(dt).count());
    g->state = VX_GRAPH_STATE_COMPLETED;
    return VX_SUCCESS;
}

}  // namespace ovxrt
