// ovx_objects.cpp -- the runtime's data objects: reference counting, error objects, device memory (HBM) of images, pyramids, arrays and
// matrices, creation and destruction.  Part of the OpenVX-subset runtime (ovx_internal.hpp); no kernels are launched here.
#include "ovx_internal.hpp"

namespace ovxrt {

using namespace vsa;

// ---- reference helpers -------------------------------------------------------------------------

void log_msg(_vx_context *ctx, _vx_reference *ref, vx_status st, const char *msg)
{
    if (ctx && ctx->log_cb && ctx->logging) ctx->log_cb(ctx, ref, st, msg);
}

_vx_reference *error_object(_vx_context *ctx, vx_status st)
{
    if (!valid(ctx)) return nullptr;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    auto it = ctx->errors.find(st);
    if (it != ctx->errors.end()) return it->second;
    auto *e = new _vx_reference();
    e->type = VX_TYPE_ERROR;
    e->ctx = ctx;
    e->status = st;
    ctx->errors[st] = e;
    return e;
}


void track(_vx_context *ctx, _vx_reference *r)
{
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->live.insert(r);
}




// ---- device memory -----------------------------------------------------------------------------
vx_status ensure_image(_vx_image *im)
{
    if (im->parent) {   // a view: the parent's memory at the ROI's origin (recomputed: the parent's memory may have moved)
        vx_status s = ensure_image(im->parent);
        if (s != VX_SUCCESS) return s;
        im->pitch = im->parent->pitch;
        im->dev = im->parent->dev + (size_t)im->roi_y * im->pitch + (size_t)im->roi_x * im->bpp;
        return VX_SUCCESS;
    }
    if (im->dev) return VX_SUCCESS;
    if (im->ctx->no_device) return (vx_status)NVX_ERROR_NO_GPU;
    if (im->width == 0 || im->height == 0) return VX_ERROR_INVALID_DIMENSION;      // a virtual image nobody sized
    im->pitch = (im->width * im->bpp + kPitchAlign - 1) / kPitchAlign * kPitchAlign;
    void *p = nullptr;
    const size_t bytes = (size_t)im->pitch * im->height + kRowSlack;
    if (hipMalloc(&p, bytes) != hipSuccess) return VX_ERROR_NO_MEMORY;
    // hipMemset is asynchronous on the null stream and our streams are non-blocking: finish it now,
    // or it could land after the first kernel that writes this image
    hipError_t e = hipSuccess;
    if (im->has_fill && im->bpp == 4) e = hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p), (int)im->fill, bytes / 4);
    else e = hipMemset(p, im->has_fill ? (int)(im->fill & 0xffu) : 0, bytes);
    if (e != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipFree(p); return VX_FAILURE; }
    im->dev = static_cast<uint8_t *>(p);
    return VX_SUCCESS;
}

vx_status ensure_pyramid(_vx_pyramid *p)
{
    for (auto *l : p->levels) {
        vx_status s = ensure_image(l);
        if (s != VX_SUCCESS) return s;
    }
    return VX_SUCCESS;
}

vx_status ensure_array(_vx_array *a)
{
    if (a->dev) return VX_SUCCESS;
    if (a->ctx->no_device) return (vx_status)NVX_ERROR_NO_GPU;
    const size_t payload = (a->capacity * a->item_size + 15) & ~(size_t)15;
    void *p = nullptr;
    if (hipMalloc(&p, payload + 16) != hipSuccess) return VX_ERROR_NO_MEMORY;
    if (hipMemset(p, 0, payload + 16) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipFree(p); return VX_FAILURE; }
    a->dev = static_cast<uint8_t *>(p);
    a->dev_count = reinterpret_cast<uint32_t *>(a->dev + payload);
    if (a->num_items) {
        const uint32_t n = (uint32_t)a->num_items;
        if (hipMemcpy(a->dev_count, &n, 4, hipMemcpyHostToDevice) != hipSuccess) return VX_FAILURE;
    }
    return VX_SUCCESS;
}

size_t item_size_of(vx_enum t)
{
    switch (t) {
    case VX_TYPE_CHAR: case VX_TYPE_INT8: case VX_TYPE_UINT8: return 1;
    case VX_TYPE_INT16: case VX_TYPE_UINT16: return 2;
    case VX_TYPE_INT32: case VX_TYPE_UINT32: case VX_TYPE_FLOAT32: case VX_TYPE_ENUM: case VX_TYPE_DF_IMAGE: return 4;
    case VX_TYPE_INT64: case VX_TYPE_UINT64: case VX_TYPE_FLOAT64: return 8;
    case VX_TYPE_SIZE: return sizeof(vx_size);
    case VX_TYPE_BOOL: return sizeof(vx_bool);
    case VX_TYPE_RECTANGLE: return sizeof(vx_rectangle_t);
    case VX_TYPE_KEYPOINT: return sizeof(vx_keypoint_t);
    case VX_TYPE_COORDINATES2D: return sizeof(vx_coordinates2d_t);
    case VX_TYPE_COORDINATES3D: return sizeof(vx_coordinates3d_t);
    case NVX_TYPE_POINT2F: return sizeof(nvx_point2f_t);
    case NVX_TYPE_POINT3F: return sizeof(nvx_point3f_t);
    case NVX_TYPE_POINT4F: return sizeof(nvx_point4f_t);
    case NVX_TYPE_KEYPOINTF: return sizeof(nvx_keypointf_t);
    default: return 0;
    }
}

// ---- destruction -------------------------------------------------------------------------------
void destroy(_vx_reference *r)
{
    _vx_context *ctx = r->ctx;
    switch (r->type) {
    case VX_TYPE_IMAGE: {
        auto *im = static_cast<_vx_image *>(r);
        if (im->parent) release_ref(im->parent);
        else if (im->dev) (void)hipFree(im->dev);
        break;
    }
    case VX_TYPE_PYRAMID: {
        auto *p = static_cast<_vx_pyramid *>(r);
        for (auto *l : p->levels) { l->pyramid = nullptr; release_ref(l); }
        break;
    }
    case VX_TYPE_ARRAY: {
        auto *a = static_cast<_vx_array *>(r);
        if (a->dev) (void)hipFree(a->dev);
        break;
    }
    case VX_TYPE_DELAY: {
        auto *d = static_cast<_vx_delay *>(r);
        for (auto *s : d->slots) { s->delay = nullptr; release_ref(s); }
        break;
    }
    case VX_TYPE_MATRIX: {
        auto *m = static_cast<_vx_matrix *>(r);
        if (m->dev) (void)hipFree(m->dev);
        break;
    }
    case VX_TYPE_PARAMETER: {
        auto *pr = static_cast<_vx_parameter *>(r);
        release_ref(pr->node);
        break;
    }
    case VX_TYPE_NODE: {
        auto *n = static_cast<_vx_node *>(r);
        if (n->kernel) {
            if (n->user_initialized && n->kernel->deinit) (void)n->kernel->deinit(n, n->params.data(), (vx_uint32)n->params.size());
            release_ref(n->kernel);
        }
        for (auto *p : n->params) release_ref(p);
        if (n->ev0) (void)hipEventDestroy(n->ev0);
        if (n->ev1) (void)hipEventDestroy(n->ev1);
        if (n->scratch) (void)hipFree(n->scratch);
        if (n->dev_num_corners) (void)hipFree(n->dev_num_corners);
        break;
    }
    case VX_TYPE_GRAPH: {
        auto *g = static_cast<_vx_graph *>(r);
        if (g->stream) { (void)hipStreamSynchronize(g->stream); }
        for (auto *n : g->nodes) { n->graph = nullptr; release_ref(n); }
        if (g->stream) (void)hipStreamDestroy(g->stream);
        if (g->pinned) (void)hipHostFree(g->pinned);
        break;
    }
    default: break;
    }
    if (ctx) {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        ctx->live.erase(r);
    }
    r->magic = 0;
    switch (r->type) {
    case VX_TYPE_IMAGE: delete static_cast<_vx_image *>(r); break;
    case VX_TYPE_PYRAMID: delete static_cast<_vx_pyramid *>(r); break;
    case VX_TYPE_ARRAY: delete static_cast<_vx_array *>(r); break;
    case VX_TYPE_SCALAR: delete static_cast<_vx_scalar *>(r); break;
    case VX_TYPE_DELAY: delete static_cast<_vx_delay *>(r); break;
    case VX_TYPE_MATRIX: delete static_cast<_vx_matrix *>(r); break;
    case VX_TYPE_NODE: delete static_cast<_vx_node *>(r); break;
    case VX_TYPE_GRAPH: delete static_cast<_vx_graph *>(r); break;
    case VX_TYPE_KERNEL: delete static_cast<_vx_kernel *>(r); break;
    case VX_TYPE_PARAMETER: delete static_cast<_vx_parameter *>(r); break;
    case VX_TYPE_META_FORMAT: delete static_cast<_vx_meta_format *>(r); break;
    default: delete r; break;
    }
}

void release_ref(_vx_reference *r)
{
    if (!valid(r) || r->type == VX_TYPE_ERROR || r->type == VX_TYPE_CONTEXT) return;
    if (r->refs.fetch_sub(1, std::memory_order_acq_rel) <= 1) destroy(r);
}


// ---- creation ----------------------------------------------------------------------------------
_vx_image *new_image(_vx_context *ctx, uint32_t w, uint32_t h, vx_df_image fmt)
{
    auto *im = new _vx_image();
    im->type = VX_TYPE_IMAGE;
    im->ctx = ctx;
    im->width = w; im->height = h; im->format = fmt;
    im->bpp = (fmt == VX_DF_IMAGE_RGBX) ? 4u : 1u;
    track(ctx, im);
    return im;
}

// float ceil sizing of level i, as the callers compute it (feature_tracker_nvxcu.cpp:164-176)
_vx_pyramid *new_pyramid(_vx_context *ctx, size_t levels, float scale, uint32_t w, uint32_t h, vx_df_image fmt)
{
    auto *p = new _vx_pyramid();
    p->type = VX_TYPE_PYRAMID;
    p->ctx = ctx;
    p->scale = scale; p->width = w; p->height = h; p->format = fmt;
    float cur = 1.0f;
    uint32_t cw = w, ch = h;
    for (size_t i = 0; i < levels; ++i) {
        _vx_image *l = new_image(ctx, cw, ch, fmt);
        l->pyramid = p;
        p->levels.push_back(l);
        cur *= scale;
        cw = (uint32_t)ceilf((float)w * cur);
        ch = (uint32_t)ceilf((float)h * cur);
    }
    track(ctx, p);
    return p;
}

_vx_array *new_array(_vx_context *ctx, vx_enum item_type, size_t capacity)
{
    auto *a = new _vx_array();
    a->type = VX_TYPE_ARRAY;
    a->ctx = ctx;
    a->item_type = item_type;
    a->item_size = item_size_of(item_type);
    a->capacity = capacity;
    track(ctx, a);
    return a;
}

_vx_scalar *new_scalar(_vx_context *ctx, vx_enum type, const void *ptr)
{
    auto *s = new _vx_scalar();
    s->type = VX_TYPE_SCALAR;
    s->ctx = ctx;
    s->data_type = type;
    if (ptr) memcpy(&s->v, ptr, item_size_of(type));
    track(ctx, s);
    return s;
}

_vx_matrix *new_matrix(_vx_context *ctx, vx_enum data_type, size_t cols, size_t rows)
{
    auto *m = new _vx_matrix();
    m->type = VX_TYPE_MATRIX;
    m->ctx = ctx;
    m->data_type = data_type;
    m->cols = cols; m->rows = rows;
    m->host.assign(cols * rows * item_size_of(data_type), 0);
    track(ctx, m);
    return m;
}

vx_status ensure_matrix(_vx_matrix *m)
{
    if (m->dev) return VX_SUCCESS;
    if (m->ctx->no_device) return (vx_status)NVX_ERROR_NO_GPU;
    return hipMalloc(&m->dev, m->host.size() + 16) == hipSuccess ? VX_SUCCESS : VX_ERROR_NO_MEMORY;
}

// ---- pinned count read-back --------------------------------------------------------------------
vx_status refresh_count_now(_vx_array *a, hipStream_t stream)
{
    _vx_context *ctx = a->ctx;
    if (hipMemcpyAsync(ctx->pinned, a->dev_count, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return VX_FAILURE;
    if (hipStreamSynchronize(stream) != hipSuccess) return VX_FAILURE;
    a->num_items = std::min<size_t>(ctx->pinned[0], a->capacity);
    return VX_SUCCESS;
}

}  // namespace ovxrt
