// ovx_runtime.cpp -- the C API (extern "C") of the OpenVX-subset runtime behind include/VX/vx.h, VX/vxu.h and
// NVX/nvx.h (layer L2 of SURVEY.md section 1): argument checking, object creation and queries, map / copy accessors, delays,
// node constructors, user kernels, immediate-mode calls.  The object model and device memory live in ovx_objects.cpp, graph
// verification (order, fusions) in ovx_verify.cpp, node execution and graph runs in ovx_exec.cpp; ovx_internal.hpp is what they share.
//
// It replaces the closed libvisionworks runtime for the call sequence of
// demos/feature_tracker/feature_tracker.cpp (cited per function in the headers).  All compute is
// the HIP kernels of this library (kernels.hpp); there is no host fallback: without a GPU
// vxCreateContext returns a context whose status is NVX_ERROR_NO_GPU.
#include "ovx_internal.hpp"

namespace {

using namespace vsa;
using namespace ovxrt;

// ---- node construction -------------------------------------------------------------------------
vx_node make_node(vx_graph graph, KernelKind kind, std::vector<ParamSpec> spec, std::vector<_vx_reference *> params,
                  bool allow_unset = false)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    _vx_context *ctx = g->ctx;
    for (size_t i = 0; i < params.size(); ++i) {
        _vx_reference *r = params[i];
        if (!r) {
            if (!spec[i].optional && !allow_unset) return make_error<vx_node>(ctx, VX_ERROR_INVALID_PARAMETERS, "node creation: a required parameter is NULL");
            continue;
        }
        if (!valid(r, spec[i].type)) return make_error<vx_node>(ctx, VX_ERROR_INVALID_REFERENCE, "node creation: a parameter has the wrong type");
        if (r->ctx != ctx) return make_error<vx_node>(ctx, VX_ERROR_INVALID_SCOPE, "node creation: parameter from another context");
    }
    auto *n = new _vx_node();
    n->type = VX_TYPE_NODE;
    n->ctx = ctx;
    n->graph = g;
    n->kind = kind;
    n->spec = std::move(spec);
    n->params = std::move(params);
    for (auto *p : n->params) retain(p);
    track(ctx, n);
    retain(n);                       // one reference for the graph, one for the caller
    g->nodes.push_back(n);
    g->verified = false;
    g->state = VX_GRAPH_STATE_UNVERIFIED;
    return n;
}


vx_node corner_node(vx_graph graph, KernelKind kind, vx_image input, vx_array output, vx_image mask, vx_array tracked,
                    _vx_scalar *p4, _vx_scalar *p5, vx_uint32 cell, vx_scalar num_corners)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) { release_ref(p4); release_ref(p5); return nullptr; }
    _vx_scalar *p6 = new_scalar(g->ctx, VX_TYPE_UINT32, &cell);
    vx_node n = make_node(graph, kind,
                          { { VX_INPUT, VX_TYPE_IMAGE, false }, { VX_OUTPUT, VX_TYPE_ARRAY, false },
                            { VX_INPUT, VX_TYPE_IMAGE, true }, { VX_INPUT, VX_TYPE_ARRAY, true },
                            { VX_INPUT, VX_TYPE_SCALAR, false }, { VX_INPUT, VX_TYPE_SCALAR, false },
                            { VX_INPUT, VX_TYPE_SCALAR, false }, { VX_OUTPUT, VX_TYPE_SCALAR, true } },
                          { R(input), R(output), R(mask), R(tracked), p4, p5, p6, R(num_corners) });
    release_ref(p4); release_ref(p5); release_ref(p6);   // the node keeps its own references
    return n;
}

// ---- immediate mode ----------------------------------------------------------------------------
vx_status immediate_scratch(_vx_context *ctx, size_t need)
{
    if (need <= ctx->scratch_size) return VX_SUCCESS;
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_size = 0;
    if (hipMalloc(&ctx->scratch, need + sizeof(size_t)) != hipSuccess) return VX_ERROR_NO_MEMORY;
    ctx->scratch_size = need;
    return VX_SUCCESS;
}

vx_status immediate_corner(vx_context context, Detector det, vx_image input, vx_array output, vx_image mask,
                           vx_array tracked, float k, float thr, uint32_t fast_n, uint32_t cell, vx_scalar num_corners)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    if (context->no_device) return (vx_status)NVX_ERROR_NO_GPU;
    auto *in = as<_vx_image>(input, VX_TYPE_IMAGE);
    auto *out = as<_vx_array>(output, VX_TYPE_ARRAY);
    if (!in || !out) return VX_ERROR_INVALID_REFERENCE;
    if (mask && !valid(mask, VX_TYPE_IMAGE)) return VX_ERROR_INVALID_REFERENCE;
    if (tracked && !valid(tracked, VX_TYPE_ARRAY)) return VX_ERROR_INVALID_REFERENCE;
    if (cell == 0) return VX_ERROR_INVALID_VALUE;
    std::lock_guard<std::mutex> imm(context->imm_mu);
    if (hipSetDevice(context->device) != hipSuccess) return VX_FAILURE;
    const size_t need = corner_track_scratch_bytes(in->width, in->height, cell);
    vx_status s = immediate_scratch(context, need);
    if (s != VX_SUCCESS) return s;
    size_t *dev_nc = num_corners ? reinterpret_cast<size_t *>(static_cast<uint8_t *>(context->scratch) + context->scratch_size) : nullptr;
    s = do_corner_track(det, in, out, mask, tracked, k, thr, fast_n, cell, dev_nc, nullptr, context->scratch, context->scratch_size,
                        context->stream);
    if (s != VX_SUCCESS) return s;
    if (dev_nc) (void)hipMemcpyAsync(&context->pinned[2], dev_nc, 8, hipMemcpyDeviceToHost, context->stream);
    s = refresh_count_now(out, context->stream);
    if (s == VX_SUCCESS && dev_nc) {
        uint64_t v;
        memcpy(&v, &context->pinned[2], 8);
        auto *sc = as<_vx_scalar>(num_corners, VX_TYPE_SCALAR);
        if (sc && sc->data_type == VX_TYPE_SIZE) sc->v.sz = (vx_size)v;
        else if (sc && sc->data_type == VX_TYPE_UINT32) sc->v.u32 = (vx_uint32)v;
    }
    return s;
}

vx_status query_out(void *ptr, vx_size size, const void *src, size_t n)
{
    if (!ptr || size != n) return VX_ERROR_INVALID_PARAMETERS;
    memcpy(ptr, src, n);
    return VX_SUCCESS;
}

}  // namespace

// ================================================================================================
// C API
// ================================================================================================
extern "C" {

// ---- context -----------------------------------------------------------------------------------
vx_context vxCreateContext(void)
{
    auto *ctx = new _vx_context();
    ctx->type = VX_TYPE_CONTEXT;
    ctx->ctx = ctx;
    int count = 0;
    const bool have_gpu = hipGetDeviceCount(&count) == hipSuccess && count > 0;
    if (!have_gpu) {
        const char *allow = getenv("VISUALSLAM_AMD_ALLOW_NO_GPU");   // host-logic tests only: nothing can run
        if (allow && allow[0] == '1') { ctx->no_device = true; return ctx; }
        ctx->status = (vx_status)NVX_ERROR_NO_GPU;
        return ctx;
    }
    // the context binds the calling thread's current device; VISUALSLAM_AMD_DEVICE=<ordinal> overrides it (one process
    // per GPU launchers that cannot call hipSetDevice themselves, e.g. bench.py's ranks)
    if (const char *dev = getenv("VISUALSLAM_AMD_DEVICE")) {
        const int want = atoi(dev);
        if (want < 0 || want >= count || hipSetDevice(want) != hipSuccess) { ctx->status = VX_ERROR_INVALID_PARAMETERS; return ctx; }
    }
    // VISUALSLAM_AMD_BLOCKING_SYNC=1: the host waits for a graph asleep (hipDeviceScheduleBlockingSync) instead of polling.
    // HIP's default keeps a core busy per waiting thread; an application with more camera-stream threads than cores then
    // has threads waiting milliseconds for a core while others poll.  Process-wide, so opt-in.
    if (const char *b = getenv("VISUALSLAM_AMD_BLOCKING_SYNC"))
        if (b[0] == '1') (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    if (hipGetDevice(&ctx->device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&ctx->pinned), 16 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        ctx->status = VX_FAILURE;
    }
    return ctx;
}

vx_status vxReleaseContext(vx_context *context)
{
    if (!context) return VX_ERROR_INVALID_REFERENCE;
    _vx_context *ctx = *context;
    if (!ctx) return VX_SUCCESS;
    if (!valid(ctx) || ctx->type != VX_TYPE_CONTEXT) return VX_ERROR_INVALID_REFERENCE;
    *context = nullptr;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    // tear down whatever the application left behind: graphs first, then containers, then leaves
    ctx->kernels.clear();
    static const vx_enum order[] = { VX_TYPE_PARAMETER, VX_TYPE_GRAPH, VX_TYPE_NODE, VX_TYPE_KERNEL, VX_TYPE_DELAY, VX_TYPE_PYRAMID, VX_TYPE_IMAGE,
                                     VX_TYPE_ARRAY, VX_TYPE_MATRIX, VX_TYPE_SCALAR };
    for (vx_enum t : order) {
        for (;;) {
            _vx_reference *victim = nullptr;
            // views before the images they look into
            if (t == VX_TYPE_IMAGE)
                for (auto *r : ctx->live) if (r->type == t && static_cast<_vx_image *>(r)->parent) { victim = r; break; }
            if (!victim)
                for (auto *r : ctx->live) if (r->type == t) { victim = r; break; }
            if (!victim) break;
            victim->refs = 0;
            destroy(victim);
        }
    }
    for (auto &m : ctx->maps) free(m.second.host);
    for (auto &e : ctx->errors) { e.second->magic = 0; delete e.second; }
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->magic = 0;
    delete ctx;
    return VX_SUCCESS;
}

vx_context vxGetContext(vx_reference reference) { return valid(reference) ? reference->ctx : nullptr; }

vx_status vxQueryContext(vx_context context, vx_enum attribute, void *ptr, vx_size size)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_CONTEXT_VENDOR_ID: { vx_uint16 v = VX_ID_DEFAULT; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_CONTEXT_VERSION: { vx_uint16 v = (vx_uint16)VX_VERSION; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_CONTEXT_REFERENCES: {
        std::lock_guard<std::recursive_mutex> lk(context->mu);
        vx_uint32 v = (vx_uint32)context->live.size();
        return query_out(ptr, size, &v, sizeof(v));
    }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_status vxGetStatus(vx_reference reference)
{
    if (!reference) return VX_ERROR_NO_RESOURCES;
    if (!valid(reference)) return VX_ERROR_INVALID_REFERENCE;
    return reference->status;
}

vx_status vxDirective(vx_reference reference, vx_enum directive)
{
    if (!valid(reference)) return VX_ERROR_INVALID_REFERENCE;
    _vx_context *ctx = reference->ctx;
    // performance directives given a node or a graph apply to that object only: timing a node costs two HIP
    // events on the graph's stream per execution, which an application may not want on every node
    if (directive == VX_DIRECTIVE_ENABLE_PERFORMANCE || directive == VX_DIRECTIVE_DISABLE_PERFORMANCE) {
        const bool on = directive == VX_DIRECTIVE_ENABLE_PERFORMANCE;
        if (reference->type == VX_TYPE_NODE) static_cast<_vx_node *>(reference)->perf_enabled = on;
        else if (reference->type == VX_TYPE_GRAPH) static_cast<_vx_graph *>(reference)->perf_enabled = on;
        else ctx->perf = on;
        return VX_SUCCESS;
    }
    switch (directive) {
    case VX_DIRECTIVE_ENABLE_LOGGING: ctx->logging = true; return VX_SUCCESS;
    case VX_DIRECTIVE_DISABLE_LOGGING: ctx->logging = false; return VX_SUCCESS;
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

void vxRegisterLogCallback(vx_context context, vx_log_callback_f callback, vx_bool)
{
    if (valid(context) && context->type == VX_TYPE_CONTEXT) context->log_cb = callback;
}

void vxAddLogEntry(vx_reference ref, vx_status status, const char *message, ...)
{
    if (!valid(ref) || !message || status == VX_SUCCESS) return;
    char buf[VX_MAX_LOG_MESSAGE_LEN];
    va_list ap;
    va_start(ap, message);
    vsnprintf(buf, sizeof(buf), message, ap);
    va_end(ap);
    log_msg(ref->ctx, ref, status, buf);
}

vx_status vxQueryReference(vx_reference ref, vx_enum attribute, void *ptr, vx_size size)
{
    if (!valid(ref)) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_REFERENCE_COUNT: { vx_uint32 v = (vx_uint32)ref->refs.load(); return query_out(ptr, size, &v, sizeof(v)); }
    case VX_REFERENCE_TYPE: { vx_enum v = ref->type; return query_out(ptr, size, &v, sizeof(v)); }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_status vxRetainReference(vx_reference ref)
{
    if (!valid(ref) || ref->type == VX_TYPE_ERROR) return VX_ERROR_INVALID_REFERENCE;
    retain(ref);
    return VX_SUCCESS;
}

vx_status vxReleaseReference(vx_reference *ref)
{
    if (!ref) return VX_ERROR_INVALID_REFERENCE;
    if (!*ref) return VX_SUCCESS;
    if (!valid(*ref)) return VX_ERROR_INVALID_REFERENCE;
    if ((*ref)->type == VX_TYPE_CONTEXT) return vxReleaseContext(reinterpret_cast<vx_context *>(ref));
    release_ref(*ref);
    *ref = nullptr;
    return VX_SUCCESS;
}

// ---- image -------------------------------------------------------------------------------------
vx_image vxCreateImage(vx_context context, vx_uint32 width, vx_uint32 height, vx_df_image color)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_image>(context, context->status);
    if (width == 0 || height == 0) return make_error<vx_image>(context, VX_ERROR_INVALID_DIMENSION, "vxCreateImage: zero dimension");
    if (color != VX_DF_IMAGE_U8 && color != VX_DF_IMAGE_RGBX)
        return make_error<vx_image>(context, VX_ERROR_INVALID_FORMAT, "vxCreateImage: only VX_DF_IMAGE_U8 and VX_DF_IMAGE_RGBX are offered");
    return new_image(context, width, height, color);
}

vx_status vxReleaseImage(vx_image *image) { return release_handle(image, VX_TYPE_IMAGE); }

vx_image vxCreateUniformImage(vx_context context, vx_uint32 width, vx_uint32 height, vx_df_image color, const vx_pixel_value_t *value)
{
    vx_image im = vxCreateImage(context, width, height, color);
    if (!valid(im, VX_TYPE_IMAGE)) return im;
    if (!value) { release_ref(im); return make_error<vx_image>(context, VX_ERROR_INVALID_PARAMETERS, "vxCreateUniformImage: no value"); }
    im->has_fill = true;
    if (color == VX_DF_IMAGE_RGBX) memcpy(&im->fill, value->RGBX, 4);
    else im->fill = value->U8;
    return im;
}

vx_image vxCreateImageFromROI(vx_image img, const vx_rectangle_t *rect)
{
    auto *parent = as<_vx_image>(img, VX_TYPE_IMAGE);
    if (!parent) return nullptr;
    _vx_context *ctx = parent->ctx;
    if (!rect || rect->start_x >= rect->end_x || rect->start_y >= rect->end_y || rect->end_x > parent->width || rect->end_y > parent->height)
        return make_error<vx_image>(ctx, VX_ERROR_INVALID_PARAMETERS, "vxCreateImageFromROI: the rectangle must lie inside the image");
    _vx_image *v = new_image(ctx, rect->end_x - rect->start_x, rect->end_y - rect->start_y, parent->format);
    v->parent = parent;
    v->roi_x = rect->start_x;
    v->roi_y = rect->start_y;
    retain(parent);
    return v;
}

vx_status vxGetValidRegionImage(vx_image image, vx_rectangle_t *rect)
{
    auto *im = as<_vx_image>(image, VX_TYPE_IMAGE);
    if (!im) return VX_ERROR_INVALID_REFERENCE;
    if (!rect) return VX_ERROR_INVALID_PARAMETERS;
    rect->start_x = 0; rect->start_y = 0; rect->end_x = im->width; rect->end_y = im->height;
    return VX_SUCCESS;
}

vx_status vxQueryImage(vx_image image, vx_enum attribute, void *ptr, vx_size size)
{
    auto *im = as<_vx_image>(image, VX_TYPE_IMAGE);
    if (!im) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_IMAGE_WIDTH: return query_out(ptr, size, &im->width, sizeof(vx_uint32));
    case VX_IMAGE_HEIGHT: return query_out(ptr, size, &im->height, sizeof(vx_uint32));
    case VX_IMAGE_FORMAT: return query_out(ptr, size, &im->format, sizeof(vx_df_image));
    case VX_IMAGE_PLANES: { vx_size v = 1; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_IMAGE_SPACE: { vx_enum v = VX_COLOR_SPACE_DEFAULT; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_IMAGE_RANGE: { vx_enum v = VX_CHANNEL_RANGE_FULL; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_IMAGE_SIZE: { vx_size v = (vx_size)im->width * im->bpp * im->height; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_IMAGE_MEMORY_TYPE: { vx_enum v = NVX_MEMORY_TYPE_HIP; return query_out(ptr, size, &v, sizeof(v)); }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

static vx_status image_rect(_vx_image *im, const vx_rectangle_t *rect, vx_rectangle_t *r)
{
    if (rect) *r = *rect; else *r = vx_rectangle_t{ 0, 0, im->width, im->height };
    if (r->start_x >= r->end_x || r->start_y >= r->end_y || r->end_x > im->width || r->end_y > im->height)
        return VX_ERROR_INVALID_PARAMETERS;
    return VX_SUCCESS;
}

vx_status vxMapImagePatch(vx_image image, const vx_rectangle_t *rect, vx_uint32 plane_index, vx_map_id *map_id,
                          vx_imagepatch_addressing_t *addr, void **ptr, vx_enum usage, vx_enum mem_type, vx_uint32)
{
    auto *im = as<_vx_image>(image, VX_TYPE_IMAGE);
    if (!im) return VX_ERROR_INVALID_REFERENCE;
    if (!map_id || !addr || !ptr || plane_index != 0) return VX_ERROR_INVALID_PARAMETERS;
    if (usage != VX_READ_ONLY && usage != VX_WRITE_ONLY && usage != VX_READ_AND_WRITE) return VX_ERROR_INVALID_PARAMETERS;
    vx_rectangle_t r;
    vx_status s = image_rect(im, rect, &r);
    if (s == VX_SUCCESS) s = ensure_image(im);
    if (s != VX_SUCCESS) return s;
    _vx_context *ctx = im->ctx;
    const size_t w = r.end_x - r.start_x, h = r.end_y - r.start_y;
    MapRecord rec{ im, usage, mem_type, nullptr, r.start_x, r.start_y, w, h };
    addr->dim_x = (vx_uint32)w; addr->dim_y = (vx_uint32)h;
    addr->stride_x = (vx_int32)im->bpp;
    addr->scale_x = addr->scale_y = VX_SCALE_UNITY;
    addr->step_x = addr->step_y = 1;
    uint8_t *dev = im->dev + (size_t)r.start_y * im->pitch + (size_t)r.start_x * im->bpp;
    if (mem_type == NVX_MEMORY_TYPE_HIP) {
        addr->stride_y = (vx_int32)im->pitch;
        *ptr = dev;
    } else if (mem_type == VX_MEMORY_TYPE_HOST) {
        const size_t row = w * im->bpp;
        rec.host = malloc(row * h);
        if (!rec.host) return VX_ERROR_NO_MEMORY;
        if (usage != VX_WRITE_ONLY) {
            if (hipMemcpy2D(rec.host, row, dev, im->pitch, row, h, hipMemcpyDeviceToHost) != hipSuccess) { free(rec.host); return VX_FAILURE; }
        }
        addr->stride_y = (vx_int32)row;
        *ptr = rec.host;
    } else {
        return VX_ERROR_NOT_SUPPORTED;
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    *map_id = ctx->next_map++;
    ctx->maps[*map_id] = rec;
    return VX_SUCCESS;
}

vx_status vxUnmapImagePatch(vx_image image, vx_map_id map_id)
{
    auto *im = as<_vx_image>(image, VX_TYPE_IMAGE);
    if (!im) return VX_ERROR_INVALID_REFERENCE;
    _vx_context *ctx = im->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    auto it = ctx->maps.find(map_id);
    if (it == ctx->maps.end() || it->second.obj != im) return VX_ERROR_INVALID_PARAMETERS;
    MapRecord rec = it->second;
    ctx->maps.erase(it);
    vx_status s = VX_SUCCESS;
    if (rec.host) {
        if (rec.usage != VX_READ_ONLY) {
            const size_t row = rec.w * im->bpp;
            uint8_t *dev = im->dev + rec.y0 * im->pitch + rec.x0 * im->bpp;
            s = hip_status(hipMemcpy2D(dev, im->pitch, rec.host, row, row, rec.h, hipMemcpyHostToDevice));
        }
        free(rec.host);
    }
    return s;
}

vx_status vxCopyImagePatch(vx_image image, const vx_rectangle_t *rect, vx_uint32 plane_index,
                           const vx_imagepatch_addressing_t *user_addr, void *user_ptr, vx_enum usage, vx_enum user_mem_type)
{
    auto *im = as<_vx_image>(image, VX_TYPE_IMAGE);
    if (!im) return VX_ERROR_INVALID_REFERENCE;
    if (!user_addr || !user_ptr || plane_index != 0 || user_mem_type != VX_MEMORY_TYPE_HOST) return VX_ERROR_INVALID_PARAMETERS;
    if (user_addr->stride_x != (vx_int32)im->bpp) return VX_ERROR_INVALID_PARAMETERS;
    vx_rectangle_t r;
    vx_status s = image_rect(im, rect, &r);
    if (s == VX_SUCCESS) s = ensure_image(im);
    if (s != VX_SUCCESS) return s;
    const size_t row = (size_t)(r.end_x - r.start_x) * im->bpp, h = r.end_y - r.start_y;
    if ((size_t)user_addr->stride_y < row) return VX_ERROR_INVALID_PARAMETERS;
    uint8_t *dev = im->dev + (size_t)r.start_y * im->pitch + (size_t)r.start_x * im->bpp;
    if (usage == VX_READ_ONLY) return hip_status(hipMemcpy2D(user_ptr, user_addr->stride_y, dev, im->pitch, row, h, hipMemcpyDeviceToHost));
    if (usage == VX_WRITE_ONLY) return hip_status(hipMemcpy2D(dev, im->pitch, user_ptr, user_addr->stride_y, row, h, hipMemcpyHostToDevice));
    return VX_ERROR_INVALID_PARAMETERS;
}

// ---- pyramid -----------------------------------------------------------------------------------
vx_pyramid vxCreatePyramid(vx_context context, vx_size levels, vx_float32 scale, vx_uint32 width, vx_uint32 height, vx_df_image format)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_pyramid>(context, context->status);
    if (levels < 1 || levels > (vx_size)vsa::kMaxPyrLevels || width == 0 || height == 0)
        return make_error<vx_pyramid>(context, VX_ERROR_INVALID_PARAMETERS, "vxCreatePyramid: 1..10 levels, non-zero size");
    if (scale != VX_SCALE_PYRAMID_HALF) return make_error<vx_pyramid>(context, VX_ERROR_NOT_SUPPORTED, "vxCreatePyramid: only VX_SCALE_PYRAMID_HALF is offered");
    if (format != VX_DF_IMAGE_U8) return make_error<vx_pyramid>(context, VX_ERROR_INVALID_FORMAT, "vxCreatePyramid: only VX_DF_IMAGE_U8 is offered");
    return new_pyramid(context, levels, scale, width, height, format);
}

vx_status vxReleasePyramid(vx_pyramid *pyr) { return release_handle(pyr, VX_TYPE_PYRAMID); }

vx_status vxQueryPyramid(vx_pyramid pyr, vx_enum attribute, void *ptr, vx_size size)
{
    auto *p = as<_vx_pyramid>(pyr, VX_TYPE_PYRAMID);
    if (!p) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_PYRAMID_LEVELS: { vx_size v = p->levels.size(); return query_out(ptr, size, &v, sizeof(v)); }
    case VX_PYRAMID_SCALE: return query_out(ptr, size, &p->scale, sizeof(vx_float32));
    case VX_PYRAMID_WIDTH: return query_out(ptr, size, &p->width, sizeof(vx_uint32));
    case VX_PYRAMID_HEIGHT: return query_out(ptr, size, &p->height, sizeof(vx_uint32));
    case VX_PYRAMID_FORMAT: return query_out(ptr, size, &p->format, sizeof(vx_df_image));
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_image vxGetPyramidLevel(vx_pyramid pyr, vx_uint32 index)
{
    auto *p = as<_vx_pyramid>(pyr, VX_TYPE_PYRAMID);
    if (!p) return nullptr;
    if (index >= p->levels.size()) return make_error<vx_image>(p->ctx, VX_ERROR_INVALID_PARAMETERS, "vxGetPyramidLevel: index out of range");
    retain(p->levels[index]);        // owned by the caller (feature_tracker.cpp:362 ... :448)
    return p->levels[index];
}

// ---- array -------------------------------------------------------------------------------------
vx_array vxCreateArray(vx_context context, vx_enum item_type, vx_size capacity)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_array>(context, context->status);
    if (item_size_of(item_type) == 0) return make_error<vx_array>(context, VX_ERROR_INVALID_TYPE, "vxCreateArray: unknown item type");
    if (capacity == 0 || capacity > 0xffffffffull) return make_error<vx_array>(context, VX_ERROR_INVALID_PARAMETERS, "vxCreateArray: capacity");
    return new_array(context, item_type, capacity);
}

vx_status vxReleaseArray(vx_array *arr) { return release_handle(arr, VX_TYPE_ARRAY); }

vx_status vxQueryArray(vx_array arr, vx_enum attribute, void *ptr, vx_size size)
{
    auto *a = as<_vx_array>(arr, VX_TYPE_ARRAY);
    if (!a) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_ARRAY_ITEMTYPE: return query_out(ptr, size, &a->item_type, sizeof(vx_enum));
    case VX_ARRAY_NUMITEMS: { vx_size v = a->num_items; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_ARRAY_CAPACITY: { vx_size v = a->capacity; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_ARRAY_ITEMSIZE: { vx_size v = a->item_size; return query_out(ptr, size, &v, sizeof(v)); }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

static vx_status set_count(_vx_array *a, size_t n)
{
    a->num_items = n;
    const uint32_t v = (uint32_t)n;
    return hip_status(hipMemcpy(a->dev_count, &v, 4, hipMemcpyHostToDevice));
}

vx_status vxAddArrayItems(vx_array arr, vx_size count, const void *ptr, vx_size stride)
{
    auto *a = as<_vx_array>(arr, VX_TYPE_ARRAY);
    if (!a) return VX_ERROR_INVALID_REFERENCE;
    if (count == 0) return VX_SUCCESS;
    if (!ptr || stride < a->item_size) return VX_ERROR_INVALID_PARAMETERS;
    if (a->num_items + count > a->capacity) return VX_FAILURE;
    vx_status s = ensure_array(a);
    if (s != VX_SUCCESS) return s;
    uint8_t *dst = a->dev + a->num_items * a->item_size;
    if (hipMemcpy2D(dst, a->item_size, ptr, stride, a->item_size, count, hipMemcpyHostToDevice) != hipSuccess) return VX_FAILURE;
    return set_count(a, a->num_items + count);
}

vx_status vxTruncateArray(vx_array arr, vx_size new_num_items)
{
    auto *a = as<_vx_array>(arr, VX_TYPE_ARRAY);
    if (!a) return VX_ERROR_INVALID_REFERENCE;
    if (new_num_items > a->num_items) return VX_ERROR_INVALID_PARAMETERS;
    if (!a->dev) { a->num_items = new_num_items; return VX_SUCCESS; }
    return set_count(a, new_num_items);
}

vx_status vxMapArrayRange(vx_array arr, vx_size range_start, vx_size range_end, vx_map_id *map_id, vx_size *stride,
                          void **ptr, vx_enum usage, vx_enum mem_type, vx_uint32)
{
    auto *a = as<_vx_array>(arr, VX_TYPE_ARRAY);
    if (!a) return VX_ERROR_INVALID_REFERENCE;
    if (!map_id || !stride || !ptr || range_start >= range_end || range_end > a->num_items) return VX_ERROR_INVALID_PARAMETERS;
    if (usage != VX_READ_ONLY && usage != VX_WRITE_ONLY && usage != VX_READ_AND_WRITE) return VX_ERROR_INVALID_PARAMETERS;
    vx_status s = ensure_array(a);
    if (s != VX_SUCCESS) return s;
    _vx_context *ctx = a->ctx;
    MapRecord rec{ a, usage, mem_type, nullptr, range_start, 0, range_end - range_start, 1 };
    *stride = a->item_size;
    uint8_t *dev = a->dev + range_start * a->item_size;
    if (mem_type == NVX_MEMORY_TYPE_HIP) {
        *ptr = dev;
    } else if (mem_type == VX_MEMORY_TYPE_HOST) {
        const size_t bytes = rec.w * a->item_size;
        rec.host = malloc(bytes);
        if (!rec.host) return VX_ERROR_NO_MEMORY;
        if (usage != VX_WRITE_ONLY && hipMemcpy(rec.host, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(rec.host); return VX_FAILURE; }
        *ptr = rec.host;
    } else {
        return VX_ERROR_NOT_SUPPORTED;
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    *map_id = ctx->next_map++;
    ctx->maps[*map_id] = rec;
    return VX_SUCCESS;
}

vx_status vxUnmapArrayRange(vx_array arr, vx_map_id map_id)
{
    auto *a = as<_vx_array>(arr, VX_TYPE_ARRAY);
    if (!a) return VX_ERROR_INVALID_REFERENCE;
    _vx_context *ctx = a->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    auto it = ctx->maps.find(map_id);
    if (it == ctx->maps.end() || it->second.obj != a) return VX_ERROR_INVALID_PARAMETERS;
    MapRecord rec = it->second;
    ctx->maps.erase(it);
    vx_status s = VX_SUCCESS;
    if (rec.host) {
        if (rec.usage != VX_READ_ONLY)
            s = hip_status(hipMemcpy(a->dev + rec.x0 * a->item_size, rec.host, rec.w * a->item_size, hipMemcpyHostToDevice));
        free(rec.host);
    }
    return s;
}

vx_status vxCopyArrayRange(vx_array arr, vx_size range_start, vx_size range_end, vx_size user_stride, void *user_ptr,
                           vx_enum usage, vx_enum user_mem_type)
{
    auto *a = as<_vx_array>(arr, VX_TYPE_ARRAY);
    if (!a) return VX_ERROR_INVALID_REFERENCE;
    if (!user_ptr || user_mem_type != VX_MEMORY_TYPE_HOST || user_stride < a->item_size) return VX_ERROR_INVALID_PARAMETERS;
    if (range_start >= range_end || range_end > a->num_items) return VX_ERROR_INVALID_PARAMETERS;
    vx_status s = ensure_array(a);
    if (s != VX_SUCCESS) return s;
    uint8_t *dev = a->dev + range_start * a->item_size;
    const size_t n = range_end - range_start;
    if (usage == VX_READ_ONLY) return hip_status(hipMemcpy2D(user_ptr, user_stride, dev, a->item_size, a->item_size, n, hipMemcpyDeviceToHost));
    if (usage == VX_WRITE_ONLY) return hip_status(hipMemcpy2D(dev, a->item_size, user_ptr, user_stride, a->item_size, n, hipMemcpyHostToDevice));
    return VX_ERROR_INVALID_PARAMETERS;
}

// OpenVX 1.0 access/commit: implemented on top of map/unmap (one outstanding access per pointer)
vx_status vxAccessArrayRange(vx_array arr, vx_size start, vx_size end, vx_size *stride, void **ptr, vx_enum usage)
{
    if (!ptr || *ptr != nullptr) return VX_ERROR_NOT_SUPPORTED;   // user-supplied buffers are not offered
    vx_map_id id;
    // the NVX usages hand out the HBM address instead of a host copy (sfm/SfM.cpp:769-776: device-to-device copy between arrays)
    if (usage == NVX_READ_ONLY_HIP || usage == NVX_WRITE_ONLY_HIP || usage == NVX_READ_AND_WRITE_HIP) {
        const vx_enum u = usage == NVX_READ_ONLY_HIP ? VX_READ_ONLY : usage == NVX_WRITE_ONLY_HIP ? VX_WRITE_ONLY : VX_READ_AND_WRITE;
        return vxMapArrayRange(arr, start, end, &id, stride, ptr, u, NVX_MEMORY_TYPE_HIP, 0);
    }
    return vxMapArrayRange(arr, start, end, &id, stride, ptr, usage, VX_MEMORY_TYPE_HOST, 0);
}

vx_status vxCommitArrayRange(vx_array arr, vx_size, vx_size, const void *ptr)
{
    auto *a = as<_vx_array>(arr, VX_TYPE_ARRAY);
    if (!a) return VX_ERROR_INVALID_REFERENCE;
    _vx_context *ctx = a->ctx;
    vx_map_id id = 0;
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        for (auto &m : ctx->maps) {
            const MapRecord &r = m.second;
            const void *mapped = r.mem_type == NVX_MEMORY_TYPE_HIP ? (a->dev ? a->dev + r.x0 * a->item_size : nullptr) : r.host;
            if (r.obj == a && mapped && mapped == ptr) { id = m.first; break; }
        }
    }
    return id ? vxUnmapArrayRange(arr, id) : VX_ERROR_INVALID_PARAMETERS;
}

// ---- scalar ------------------------------------------------------------------------------------
vx_scalar vxCreateScalar(vx_context context, vx_enum data_type, const void *ptr)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_scalar>(context, context->status);
    const size_t n = item_size_of(data_type);
    if (n == 0 || n > 8) return make_error<vx_scalar>(context, VX_ERROR_INVALID_TYPE, "vxCreateScalar: unsupported type");
    return new_scalar(context, data_type, ptr);
}

vx_status vxReleaseScalar(vx_scalar *scalar) { return release_handle(scalar, VX_TYPE_SCALAR); }

vx_status vxQueryScalar(vx_scalar scalar, vx_enum attribute, void *ptr, vx_size size)
{
    auto *s = as<_vx_scalar>(scalar, VX_TYPE_SCALAR);
    if (!s) return VX_ERROR_INVALID_REFERENCE;
    if (attribute == VX_SCALAR_TYPE) return query_out(ptr, size, &s->data_type, sizeof(vx_enum));
    return VX_ERROR_NOT_SUPPORTED;
}

vx_status vxReadScalarValue(vx_scalar scalar, void *ptr)
{
    auto *s = as<_vx_scalar>(scalar, VX_TYPE_SCALAR);
    if (!s) return VX_ERROR_INVALID_REFERENCE;
    if (!ptr) return VX_ERROR_INVALID_PARAMETERS;
    memcpy(ptr, &s->v, item_size_of(s->data_type));
    return VX_SUCCESS;
}

vx_status vxWriteScalarValue(vx_scalar scalar, const void *ptr)
{
    auto *s = as<_vx_scalar>(scalar, VX_TYPE_SCALAR);
    if (!s) return VX_ERROR_INVALID_REFERENCE;
    if (!ptr) return VX_ERROR_INVALID_PARAMETERS;
    memcpy(&s->v, ptr, item_size_of(s->data_type));
    return VX_SUCCESS;
}

vx_status vxCopyScalar(vx_scalar scalar, void *user_ptr, vx_enum usage, vx_enum user_mem_type)
{
    if (user_mem_type != VX_MEMORY_TYPE_HOST) return VX_ERROR_INVALID_PARAMETERS;
    if (usage == VX_READ_ONLY) return vxReadScalarValue(scalar, user_ptr);
    if (usage == VX_WRITE_ONLY) return vxWriteScalarValue(scalar, user_ptr);
    return VX_ERROR_INVALID_PARAMETERS;
}

// ---- delay -------------------------------------------------------------------------------------
vx_delay vxCreateDelay(vx_context context, vx_reference exemplar, vx_size num_slots)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_delay>(context, context->status);
    if (!valid(exemplar) || exemplar->status != VX_SUCCESS) return make_error<vx_delay>(context, VX_ERROR_INVALID_REFERENCE, "vxCreateDelay: bad exemplar");
    if (num_slots < 1 || num_slots > 64) return make_error<vx_delay>(context, VX_ERROR_INVALID_PARAMETERS, "vxCreateDelay: 1..64 slots");
    auto *d = new _vx_delay();
    d->type = VX_TYPE_DELAY;
    d->ctx = context;
    d->slot_type = exemplar->type;
    for (vx_size i = 0; i < num_slots; ++i) {
        _vx_reference *s = nullptr;
        switch (exemplar->type) {
        case VX_TYPE_IMAGE: { auto *e = static_cast<_vx_image *>(exemplar); s = new_image(context, e->width, e->height, e->format); break; }
        case VX_TYPE_PYRAMID: { auto *e = static_cast<_vx_pyramid *>(exemplar); s = new_pyramid(context, e->levels.size(), e->scale, e->width, e->height, e->format); break; }
        case VX_TYPE_ARRAY: { auto *e = static_cast<_vx_array *>(exemplar); s = new_array(context, e->item_type, e->capacity); break; }
        case VX_TYPE_SCALAR: { auto *e = static_cast<_vx_scalar *>(exemplar); s = new_scalar(context, e->data_type, &e->v); break; }
        case VX_TYPE_MATRIX: {   // demos/video_stabilizer/stabilizer.cpp:438
            auto *e = static_cast<_vx_matrix *>(exemplar);
            _vx_matrix *m = new_matrix(context, e->data_type, e->cols, e->rows);
            m->host = e->host;
            s = m;
            break;
        }
        default: break;
        }
        if (!s) {
            for (auto *x : d->slots) { x->delay = nullptr; release_ref(x); }
            delete d;
            return make_error<vx_delay>(context, VX_ERROR_NOT_SUPPORTED, "vxCreateDelay: exemplar type not offered");
        }
        s->delay = d;
        d->slots.push_back(s);
    }
    track(context, d);
    return d;
}

vx_status vxReleaseDelay(vx_delay *delay) { return release_handle(delay, VX_TYPE_DELAY); }

vx_status vxQueryDelay(vx_delay delay, vx_enum attribute, void *ptr, vx_size size)
{
    auto *d = as<_vx_delay>(delay, VX_TYPE_DELAY);
    if (!d) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_DELAY_TYPE: return query_out(ptr, size, &d->slot_type, sizeof(vx_enum));
    case VX_DELAY_SLOTS: { vx_size v = d->slots.size(); return query_out(ptr, size, &v, sizeof(v)); }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_reference vxGetReferenceFromDelay(vx_delay delay, vx_int32 index)
{
    auto *d = as<_vx_delay>(delay, VX_TYPE_DELAY);
    if (!d) return nullptr;
    if (index > 0 || (size_t)(-index) >= d->slots.size())
        return make_error<vx_reference>(d->ctx, VX_ERROR_INVALID_PARAMETERS, "vxGetReferenceFromDelay: index must be 0 .. 1-slots");
    return d->slots[(size_t)(-index)];      // borrowed
}

// The slot objects keep their index; the DATA moves: what was at index 0 is at -1 afterwards, the
// oldest data wraps to index 0 (where the next graph execution overwrites it).  Nodes hold the slot
// objects, so they see the aged data without re-verification (feature_tracker.cpp:216-224).
static void swap_image_data(_vx_image *a, _vx_image *b) { std::swap(a->dev, b->dev); std::swap(a->pitch, b->pitch); }

vx_status vxAgeDelay(vx_delay delay)
{
    auto *d = as<_vx_delay>(delay, VX_TYPE_DELAY);
    if (!d) return VX_ERROR_INVALID_REFERENCE;
    const size_t n = d->slots.size();
    for (size_t i = n - 1; i >= 1; --i) {   // rotate towards older indices by successive swaps
        _vx_reference *older = d->slots[i], *newer = d->slots[i - 1];
        switch (d->slot_type) {
        case VX_TYPE_IMAGE: swap_image_data(static_cast<_vx_image *>(older), static_cast<_vx_image *>(newer)); break;
        case VX_TYPE_PYRAMID: {
            auto *po = static_cast<_vx_pyramid *>(older), *pn = static_cast<_vx_pyramid *>(newer);
            for (size_t l = 0; l < po->levels.size(); ++l) swap_image_data(po->levels[l], pn->levels[l]);
            break;
        }
        case VX_TYPE_ARRAY: {
            auto *ao = static_cast<_vx_array *>(older), *an = static_cast<_vx_array *>(newer);
            std::swap(ao->dev, an->dev); std::swap(ao->dev_count, an->dev_count); std::swap(ao->num_items, an->num_items);
            break;
        }
        case VX_TYPE_SCALAR: std::swap(static_cast<_vx_scalar *>(older)->v, static_cast<_vx_scalar *>(newer)->v); break;
        case VX_TYPE_MATRIX: {
            auto *mo = static_cast<_vx_matrix *>(older), *mn = static_cast<_vx_matrix *>(newer);
            std::swap(mo->host, mn->host); std::swap(mo->dev, mn->dev);
            break;
        }
        default: return VX_ERROR_NOT_SUPPORTED;
        }
    }
    return VX_SUCCESS;
}

// ---- graph -------------------------------------------------------------------------------------
vx_graph vxCreateGraph(vx_context context)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_graph>(context, context->status);
    auto *g = new _vx_graph();
    g->type = VX_TYPE_GRAPH;
    g->ctx = context;
    track(context, g);
    return g;
}

vx_status vxReleaseGraph(vx_graph *graph) { return release_handle(graph, VX_TYPE_GRAPH); }

vx_status vxVerifyGraph(vx_graph graph)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return VX_ERROR_INVALID_REFERENCE;
    return verify_graph(g);
}

vx_status vxScheduleGraph(vx_graph graph)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return VX_ERROR_INVALID_REFERENCE;
    if (g->scheduled) return VX_ERROR_GRAPH_SCHEDULED;
    return enqueue_graph(g);
}

vx_status vxWaitGraph(vx_graph graph)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return VX_ERROR_INVALID_REFERENCE;
    return finish_graph(g);
}

vx_status vxProcessGraph(vx_graph graph)
{
    vx_status s = vxScheduleGraph(graph);
    return s == VX_SUCCESS ? vxWaitGraph(graph) : s;
}

vx_status vxQueryGraph(vx_graph graph, vx_enum attribute, void *ptr, vx_size size)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_GRAPH_PERFORMANCE: return query_out(ptr, size, &g->perf, sizeof(vx_perf_t));
    case VX_GRAPH_NUMNODES: { vx_uint32 v = (vx_uint32)g->nodes.size(); return query_out(ptr, size, &v, sizeof(v)); }
    case VX_GRAPH_STATE: return query_out(ptr, size, &g->state, sizeof(vx_enum));
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_status vxSetGraphAttribute(vx_graph graph, vx_enum attribute, const void *ptr, vx_size size)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return VX_ERROR_INVALID_REFERENCE;
    if (attribute == NVX_GRAPH_VERIFY_OPTIONS) {
        if (!ptr) return VX_ERROR_INVALID_PARAMETERS;
        g->options.assign(static_cast<const char *>(ptr), size);
        return VX_SUCCESS;
    }
    return VX_ERROR_NOT_SUPPORTED;
}

vx_bool vxIsGraphVerified(vx_graph graph)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    return (g && g->verified) ? vx_true_e : vx_false_e;
}

// ---- node --------------------------------------------------------------------------------------
vx_status vxReleaseNode(vx_node *node) { return release_handle(node, VX_TYPE_NODE); }

vx_status vxQueryNode(vx_node node, vx_enum attribute, void *ptr, vx_size size)
{
    auto *n = as<_vx_node>(node, VX_TYPE_NODE);
    if (!n) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_NODE_PERFORMANCE: return query_out(ptr, size, &n->perf, sizeof(vx_perf_t));
    case VX_NODE_STATUS: return query_out(ptr, size, &n->last_status, sizeof(vx_status));
    case VX_NODE_PARAMETERS: { vx_uint32 v = (vx_uint32)n->params.size(); return query_out(ptr, size, &v, sizeof(v)); }
    case VX_NODE_BORDER: {
        vx_border_t b{};
        b.mode = VX_BORDER_REPLICATE;
        if (n->kind == K_WARP_PERSPECTIVE) { b = n->border; if (b.mode == VX_BORDER_UNDEFINED) b.mode = VX_BORDER_CONSTANT; }
        return query_out(ptr, size, &b, sizeof(b));
    }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_status vxSetNodeAttribute(vx_node node, vx_enum attribute, const void *ptr, vx_size size)
{
    auto *n = as<_vx_node>(node, VX_TYPE_NODE);
    if (!n) return VX_ERROR_INVALID_REFERENCE;
    if (attribute == VX_NODE_BORDER) {
        // UNDEFINED is refined to REPLICATE and REPLICATE is what the kernels do; CONSTANT is not offered
        if (!ptr || size != sizeof(vx_border_t)) return VX_ERROR_INVALID_PARAMETERS;
        const vx_border_t *b = static_cast<const vx_border_t *>(ptr);
        if (n->kind == K_WARP_PERSPECTIVE) {   // samples outside the input: a constant (UNDEFINED is refined to constant 0)
            if (b->mode == VX_BORDER_REPLICATE) return VX_ERROR_NOT_SUPPORTED;
            n->border = *b;
            if (b->mode == VX_BORDER_UNDEFINED) memset(&n->border.constant_value, 0, sizeof(n->border.constant_value));
            return VX_SUCCESS;
        }
        return b->mode == VX_BORDER_CONSTANT ? VX_ERROR_NOT_SUPPORTED : VX_SUCCESS;
    }
    return VX_ERROR_NOT_SUPPORTED;
}

vx_status vxSetParameterByIndex(vx_node node, vx_uint32 index, vx_reference value)
{
    auto *n = as<_vx_node>(node, VX_TYPE_NODE);
    if (!n) return VX_ERROR_INVALID_REFERENCE;
    if (index >= n->params.size()) return VX_ERROR_INVALID_PARAMETERS;
    const ParamSpec &sp = n->spec[index];
    if (!value) {
        if (!sp.optional) return VX_ERROR_INVALID_PARAMETERS;
    } else {
        if (!valid(value, sp.type) || value->ctx != n->ctx) return VX_ERROR_INVALID_REFERENCE;
    }
    _vx_reference *old = n->params[index];
    if (old == value) return VX_SUCCESS;
    // same metadata keeps the graph verified (feature_tracker.cpp:216-217 rebinds every frame)
    bool same_meta = false;
    if (old && value && sp.type == VX_TYPE_IMAGE) {
        auto *a = static_cast<_vx_image *>(old), *b = static_cast<_vx_image *>(value);
        same_meta = a->width == b->width && a->height == b->height && a->format == b->format && !a->pyramid && !b->pyramid;
    } else if (sp.type == VX_TYPE_SCALAR && old && value) {
        same_meta = static_cast<_vx_scalar *>(old)->data_type == static_cast<_vx_scalar *>(value)->data_type;
    } else if (sp.optional && sp.dir == VX_INPUT && sp.type == VX_TYPE_IMAGE) {
        same_meta = true;             // an optional read-only image (the mask) appearing or disappearing changes no order
    }
    // ... unless the old or the new image is produced inside this graph (directly, as an ROI view or through its
    // parent): the data dependencies then change and the execution order has to be rebuilt
    if (same_meta && sp.type == VX_TYPE_IMAGE && n->graph && n->graph->verified &&
        (sp.dir != VX_INPUT || written_in_graph(n->graph, old) || written_in_graph(n->graph, value)))
        same_meta = false;
    retain(value);
    n->params[index] = value;
    release_ref(old);
    if (!same_meta && n->graph) { n->graph->verified = false; n->graph->state = VX_GRAPH_STATE_UNVERIFIED; }
    return VX_SUCCESS;
}

// ---- user kernels ------------------------------------------------------------------------------
vx_status vxAllocateUserKernelId(vx_context context, vx_enum *pKernelEnumId)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    if (!pKernelEnumId) return VX_ERROR_INVALID_PARAMETERS;
    std::lock_guard<std::recursive_mutex> lk(context->mu);
    if (context->next_kernel_id >= VX_KERNEL_BASE(VX_ID_DEFAULT, 0) + 0x1000) return VX_ERROR_NO_RESOURCES;
    *pKernelEnumId = context->next_kernel_id++;
    return VX_SUCCESS;
}

vx_kernel vxAddUserKernel(vx_context context, const vx_char name[VX_MAX_KERNEL_NAME], vx_enum enumeration, vx_kernel_f func_ptr,
                          vx_uint32 numParams, vx_kernel_validate_f validate, vx_kernel_initialize_f init,
                          vx_kernel_deinitialize_f deinit)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_kernel>(context, context->status);
    if (!name || !name[0] || strnlen(name, VX_MAX_KERNEL_NAME) == VX_MAX_KERNEL_NAME || !func_ptr || numParams == 0 || numParams > 32)
        return make_error<vx_kernel>(context, VX_ERROR_INVALID_PARAMETERS, "vxAddUserKernel: a name, a function and 1..32 parameters are required");
    {
        std::lock_guard<std::recursive_mutex> lk(context->mu);
        if (context->kernels.count(name)) return make_error<vx_kernel>(context, VX_ERROR_INVALID_PARAMETERS, "vxAddUserKernel: the name is taken");
    }
    auto *k = new _vx_kernel();
    k->type = VX_TYPE_KERNEL;
    k->ctx = context;
    k->name = name;
    k->enumeration = enumeration;
    k->func = func_ptr; k->validate = validate; k->init = init; k->deinit = deinit;
    k->spec.assign(numParams, ParamSpec{ VX_INPUT, VX_TYPE_INVALID, false });
    k->spec_set.assign(numParams, false);
    track(context, k);
    return k;
}

vx_kernel vxAddKernel(vx_context context, const vx_char name[VX_MAX_KERNEL_NAME], vx_enum enumeration, vx_kernel_f func_ptr,
                      vx_uint32 numParams, vx_kernel_input_validate_f input, vx_kernel_output_validate_f output,
                      vx_kernel_initialize_f init, vx_kernel_deinitialize_f deinit)
{
    vx_kernel k = vxAddUserKernel(context, name, enumeration, func_ptr, numParams, nullptr, init, deinit);
    if (valid(k, VX_TYPE_KERNEL)) { k->in_validate = input; k->out_validate = output; }
    return k;
}

vx_parameter vxGetParameterByIndex(vx_node node, vx_uint32 index)
{
    auto *n = as<_vx_node>(node, VX_TYPE_NODE);
    if (!n) return nullptr;
    if (index >= n->params.size()) return make_error<vx_parameter>(n->ctx, VX_ERROR_INVALID_PARAMETERS, "vxGetParameterByIndex: no such parameter");
    auto *p = new _vx_parameter();
    p->type = VX_TYPE_PARAMETER;
    p->ctx = n->ctx;
    p->node = n;
    p->index = index;
    retain(n);
    track(n->ctx, p);
    return p;
}

vx_status vxQueryParameter(vx_parameter param, vx_enum attribute, void *ptr, vx_size size)
{
    auto *p = as<_vx_parameter>(param, VX_TYPE_PARAMETER);
    if (!p || !valid(p->node, VX_TYPE_NODE)) return VX_ERROR_INVALID_REFERENCE;
    const ParamSpec &sp = p->node->spec[p->index];
    switch (attribute) {
    case VX_PARAMETER_INDEX: return query_out(ptr, size, &p->index, sizeof(vx_uint32));
    case VX_PARAMETER_DIRECTION: return query_out(ptr, size, &sp.dir, sizeof(vx_enum));
    case VX_PARAMETER_TYPE: return query_out(ptr, size, &sp.type, sizeof(vx_enum));
    case VX_PARAMETER_STATE: { vx_enum v = sp.optional ? VX_PARAMETER_STATE_OPTIONAL : VX_PARAMETER_STATE_REQUIRED; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_PARAMETER_REF: {
        _vx_reference *r = p->node->params[p->index];
        retain(r);                                     // the caller releases it (SfM.cpp:1233)
        return query_out(ptr, size, &r, sizeof(r));
    }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_status vxReleaseParameter(vx_parameter *param) { return release_handle(param, VX_TYPE_PARAMETER); }

vx_status vxAddParameterToKernel(vx_kernel kernel, vx_uint32 index, vx_enum dir, vx_enum data_type, vx_enum state)
{
    auto *k = as<_vx_kernel>(kernel, VX_TYPE_KERNEL);
    if (!k) return VX_ERROR_INVALID_REFERENCE;
    if (k->finalized) return VX_ERROR_NOT_SUPPORTED;
    if (index >= k->spec.size() || (dir != VX_INPUT && dir != VX_OUTPUT && dir != VX_BIDIRECTIONAL) ||
        (state != VX_PARAMETER_STATE_REQUIRED && state != VX_PARAMETER_STATE_OPTIONAL))
        return VX_ERROR_INVALID_PARAMETERS;
    switch (data_type) {
    case VX_TYPE_IMAGE: case VX_TYPE_ARRAY: case VX_TYPE_MATRIX: case VX_TYPE_SCALAR: case VX_TYPE_PYRAMID: case VX_TYPE_DELAY: break;
    default: return VX_ERROR_NOT_SUPPORTED;
    }
    k->spec[index] = ParamSpec{ dir, data_type, state == VX_PARAMETER_STATE_OPTIONAL };
    k->spec_set[index] = true;
    return VX_SUCCESS;
}

vx_status vxFinalizeKernel(vx_kernel kernel)
{
    auto *k = as<_vx_kernel>(kernel, VX_TYPE_KERNEL);
    if (!k) return VX_ERROR_INVALID_REFERENCE;
    if (k->finalized) return VX_SUCCESS;
    for (bool set : k->spec_set) if (!set) return VX_ERROR_INVALID_PARAMETERS;
    std::lock_guard<std::recursive_mutex> lk(k->ctx->mu);
    if (k->ctx->kernels.count(k->name)) return VX_ERROR_INVALID_PARAMETERS;
    k->finalized = true;
    retain(k);                         // the registry's reference
    k->ctx->kernels[k->name] = k;
    return VX_SUCCESS;
}

vx_status vxRemoveKernel(vx_kernel kernel)
{
    auto *k = as<_vx_kernel>(kernel, VX_TYPE_KERNEL);
    if (!k) return VX_ERROR_INVALID_REFERENCE;
    {
        std::lock_guard<std::recursive_mutex> lk(k->ctx->mu);
        auto it = k->ctx->kernels.find(k->name);
        if (it != k->ctx->kernels.end() && it->second == k) {
            k->ctx->kernels.erase(it);
            release_ref(k);            // the registry's
        }
    }
    release_ref(k);                    // the caller's
    return VX_SUCCESS;
}

vx_status vxReleaseKernel(vx_kernel *kernel) { return release_handle(kernel, VX_TYPE_KERNEL); }

vx_status vxQueryKernel(vx_kernel kernel, vx_enum attribute, void *ptr, vx_size size)
{
    auto *k = as<_vx_kernel>(kernel, VX_TYPE_KERNEL);
    if (!k) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_KERNEL_PARAMETERS: { vx_uint32 v = (vx_uint32)k->spec.size(); return query_out(ptr, size, &v, sizeof(v)); }
    case VX_KERNEL_ENUM: return query_out(ptr, size, &k->enumeration, sizeof(vx_enum));
    case VX_KERNEL_NAME: {
        if (!ptr || size != VX_MAX_KERNEL_NAME) return VX_ERROR_INVALID_PARAMETERS;
        memset(ptr, 0, size);
        memcpy(ptr, k->name.data(), k->name.size());
        return VX_SUCCESS;
    }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_kernel vxGetKernelByName(vx_context context, const vx_char *name)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_kernel>(context, context->status);
    std::lock_guard<std::recursive_mutex> lk(context->mu);
    auto it = name ? context->kernels.find(name) : context->kernels.end();
    if (it == context->kernels.end()) return make_error<vx_kernel>(context, VX_ERROR_INVALID_PARAMETERS, "vxGetKernelByName: no such kernel");
    retain(it->second);
    return it->second;
}

vx_kernel vxGetKernelByEnum(vx_context context, vx_enum kernel)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_kernel>(context, context->status);
    std::lock_guard<std::recursive_mutex> lk(context->mu);
    for (auto &kv : context->kernels)
        if (kv.second->enumeration == kernel) { retain(kv.second); return kv.second; }
    return make_error<vx_kernel>(context, VX_ERROR_INVALID_PARAMETERS, "vxGetKernelByEnum: no such kernel");
}

vx_node vxCreateGenericNode(vx_graph graph, vx_kernel kernel)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    auto *k = as<_vx_kernel>(kernel, VX_TYPE_KERNEL);
    if (!k || !k->finalized || k->ctx != g->ctx) return make_error<vx_node>(g->ctx, VX_ERROR_INVALID_REFERENCE, "vxCreateGenericNode: not a finalized kernel of this context");
    vx_node n = make_node(graph, K_USER, k->spec, std::vector<_vx_reference *>(k->spec.size(), nullptr), true);
    if (valid(n, VX_TYPE_NODE)) {
        n->kernel = k;
        retain(k);
    }
    return n;
}

vx_status vxSetMetaFormatAttribute(vx_meta_format meta, vx_enum attribute, const void *ptr, vx_size size)
{
    if (!valid(meta) || meta->type != VX_TYPE_META_FORMAT) return VX_ERROR_INVALID_REFERENCE;
    if (!ptr || size == 0 || size > 64) return VX_ERROR_INVALID_PARAMETERS;
    const uint8_t *b = static_cast<const uint8_t *>(ptr);
    meta->attrs[attribute].assign(b, b + size);
    return VX_SUCCESS;
}

vx_status vxSetMetaFormatFromReference(vx_meta_format meta, vx_reference exemplar)
{
    if (!valid(meta) || meta->type != VX_TYPE_META_FORMAT || !valid(exemplar)) return VX_ERROR_INVALID_REFERENCE;
    auto put = [&](vx_enum attr, size_t n) {
        std::vector<uint8_t> v(n);
        if (query_any(exemplar, attr, v.data(), n) == VX_SUCCESS) meta->attrs[attr] = v;
    };
    switch (exemplar->type) {
    case VX_TYPE_IMAGE: put(VX_IMAGE_WIDTH, 4); put(VX_IMAGE_HEIGHT, 4); put(VX_IMAGE_FORMAT, 4); break;
    case VX_TYPE_ARRAY: put(VX_ARRAY_ITEMTYPE, 4); put(VX_ARRAY_CAPACITY, sizeof(vx_size)); break;
    case VX_TYPE_MATRIX: put(VX_MATRIX_TYPE, 4); put(VX_MATRIX_ROWS, sizeof(vx_size)); put(VX_MATRIX_COLUMNS, sizeof(vx_size)); break;
    case VX_TYPE_SCALAR: put(VX_SCALAR_TYPE, 4); break;
    default: return VX_ERROR_NOT_SUPPORTED;
    }
    return VX_SUCCESS;
}

vx_node vxWarpPerspectiveNode(vx_graph graph, vx_image input, vx_matrix matrix, vx_enum type, vx_image output)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    _vx_scalar *t = new_scalar(g->ctx, VX_TYPE_ENUM, &type);
    vx_node n = make_node(graph, K_WARP_PERSPECTIVE,
                          { { VX_INPUT, VX_TYPE_IMAGE, false }, { VX_INPUT, VX_TYPE_MATRIX, false }, { VX_INPUT, VX_TYPE_SCALAR, false },
                            { VX_OUTPUT, VX_TYPE_IMAGE, false } },
                          { R(input), R(matrix), t, R(output) });
    release_ref(t);
    if (valid(n, VX_TYPE_NODE)) n->border.mode = VX_BORDER_UNDEFINED;
    return n;
}

vx_node vxColorConvertNode(vx_graph graph, vx_image input, vx_image output)
{
    return make_node(graph, K_COLOR_CONVERT, { { VX_INPUT, VX_TYPE_IMAGE, false }, { VX_OUTPUT, VX_TYPE_IMAGE, false } },
                     { R(input), R(output) });
}

vx_node nvxCopyImageNode(vx_graph graph, vx_image src, vx_image dst)
{
    return make_node(graph, K_COPY_IMAGE, { { VX_INPUT, VX_TYPE_IMAGE, false }, { VX_OUTPUT, VX_TYPE_IMAGE, false } },
                     { R(src), R(dst) });
}

vx_node vxGaussianPyramidNode(vx_graph graph, vx_image input, vx_pyramid gaussian)
{
    return make_node(graph, K_GAUSSIAN_PYRAMID, { { VX_INPUT, VX_TYPE_IMAGE, false }, { VX_OUTPUT, VX_TYPE_PYRAMID, false } },
                     { R(input), R(gaussian) });
}

vx_node vxOpticalFlowPyrLKNode(vx_graph graph, vx_pyramid old_images, vx_pyramid new_images, vx_array old_points,
                               vx_array new_points_estimates, vx_array new_points, vx_enum termination, vx_scalar epsilon,
                               vx_scalar num_iterations, vx_scalar use_initial_estimate, vx_size window_dimension)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    vx_enum term = termination;
    _vx_scalar *t = new_scalar(g->ctx, VX_TYPE_ENUM, &term);
    _vx_scalar *w = new_scalar(g->ctx, VX_TYPE_SIZE, &window_dimension);
    vx_node n = make_node(graph, K_PYRLK,
                          { { VX_INPUT, VX_TYPE_PYRAMID, false }, { VX_INPUT, VX_TYPE_PYRAMID, false },
                            { VX_INPUT, VX_TYPE_ARRAY, false }, { VX_INPUT, VX_TYPE_ARRAY, true },
                            { VX_OUTPUT, VX_TYPE_ARRAY, false }, { VX_INPUT, VX_TYPE_SCALAR, false },
                            { VX_INPUT, VX_TYPE_SCALAR, false }, { VX_INPUT, VX_TYPE_SCALAR, false },
                            { VX_INPUT, VX_TYPE_SCALAR, true }, { VX_INPUT, VX_TYPE_SCALAR, false } },
                          { R(old_images), R(new_images), R(old_points), R(new_points_estimates), R(new_points), t,
                            R(epsilon), R(num_iterations), R(use_initial_estimate), w });
    release_ref(t); release_ref(w);
    return n;
}

vx_node nvxHarrisTrackNode(vx_graph graph, vx_image input, vx_array output, vx_image mask, vx_array tracked_points,
                           vx_float32 k, vx_float32 threshold, vx_uint32 cell_size, vx_scalar num_corners)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    return corner_node(graph, K_HARRIS_TRACK, input, output, mask, tracked_points, new_scalar(g->ctx, VX_TYPE_FLOAT32, &k),
                       new_scalar(g->ctx, VX_TYPE_FLOAT32, &threshold), cell_size, num_corners);
}

vx_node nvxFastTrackNode(vx_graph graph, vx_image input, vx_array output, vx_image mask, vx_array tracked_points,
                         vx_uint32 type, vx_uint32 threshold, vx_uint32 cell_size, vx_scalar num_corners)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    return corner_node(graph, K_FAST_TRACK, input, output, mask, tracked_points, new_scalar(g->ctx, VX_TYPE_UINT32, &type),
                       new_scalar(g->ctx, VX_TYPE_UINT32, &threshold), cell_size, num_corners);
}

vx_node nvxFindHomographyNode(vx_graph graph, vx_array srcPoints, vx_array dstPoints, vx_matrix homography, vx_enum method,
                              vx_float32 threshold, vx_uint32 maxEstimateIters, vx_uint32 maxRefineIters,
                              vx_float32 confidence, vx_float32 outlierRatio, vx_array mask)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    _vx_scalar *sc[6] = { new_scalar(g->ctx, VX_TYPE_ENUM, &method), new_scalar(g->ctx, VX_TYPE_FLOAT32, &threshold),
                          new_scalar(g->ctx, VX_TYPE_UINT32, &maxEstimateIters), new_scalar(g->ctx, VX_TYPE_UINT32, &maxRefineIters),
                          new_scalar(g->ctx, VX_TYPE_FLOAT32, &confidence), new_scalar(g->ctx, VX_TYPE_FLOAT32, &outlierRatio) };
    vx_node n = make_node(graph, K_FIND_HOMOGRAPHY,
                          { { VX_INPUT, VX_TYPE_ARRAY, false }, { VX_INPUT, VX_TYPE_ARRAY, false }, { VX_OUTPUT, VX_TYPE_MATRIX, false },
                            { VX_INPUT, VX_TYPE_SCALAR, false }, { VX_INPUT, VX_TYPE_SCALAR, false }, { VX_INPUT, VX_TYPE_SCALAR, false },
                            { VX_INPUT, VX_TYPE_SCALAR, false }, { VX_INPUT, VX_TYPE_SCALAR, false }, { VX_INPUT, VX_TYPE_SCALAR, false },
                            { VX_OUTPUT, VX_TYPE_ARRAY, true } },
                          { R(srcPoints), R(dstPoints), R(homography), sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], R(mask) });
    for (auto *x : sc) release_ref(x);   // the node keeps its own references
    return n;
}

// ---- matrix --------------------------------------------------------------------------------------
vx_matrix vxCreateMatrix(vx_context context, vx_enum data_type, vx_size columns, vx_size rows)
{
    if (!valid(context) || context->type != VX_TYPE_CONTEXT) return nullptr;
    if (context->status != VX_SUCCESS) return make_error<vx_matrix>(context, context->status);
    if (data_type != VX_TYPE_FLOAT32 && data_type != VX_TYPE_INT32 && data_type != VX_TYPE_UINT8)
        return make_error<vx_matrix>(context, VX_ERROR_INVALID_TYPE, "vxCreateMatrix: unsupported element type");
    if (columns == 0 || rows == 0) return make_error<vx_matrix>(context, VX_ERROR_INVALID_DIMENSION, "vxCreateMatrix: empty matrix");
    return new_matrix(context, data_type, columns, rows);
}

vx_status vxReleaseMatrix(vx_matrix *mat) { return release_handle(mat, VX_TYPE_MATRIX); }

vx_status vxQueryMatrix(vx_matrix mat, vx_enum attribute, void *ptr, vx_size size)
{
    auto *m = as<_vx_matrix>(mat, VX_TYPE_MATRIX);
    if (!m) return VX_ERROR_INVALID_REFERENCE;
    switch (attribute) {
    case VX_MATRIX_TYPE: return query_out(ptr, size, &m->data_type, sizeof(vx_enum));
    case VX_MATRIX_ROWS: { vx_size v = m->rows; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_MATRIX_COLUMNS: { vx_size v = m->cols; return query_out(ptr, size, &v, sizeof(v)); }
    case VX_MATRIX_SIZE: { vx_size v = m->host.size(); return query_out(ptr, size, &v, sizeof(v)); }
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

vx_status vxCopyMatrix(vx_matrix mat, void *user_ptr, vx_enum usage, vx_enum user_mem_type)
{
    auto *m = as<_vx_matrix>(mat, VX_TYPE_MATRIX);
    if (!m) return VX_ERROR_INVALID_REFERENCE;
    if (!user_ptr || user_mem_type != VX_MEMORY_TYPE_HOST) return VX_ERROR_INVALID_PARAMETERS;
    if (usage == VX_READ_ONLY) memcpy(user_ptr, m->host.data(), m->host.size());
    else if (usage == VX_WRITE_ONLY) memcpy(m->host.data(), user_ptr, m->host.size());
    else return VX_ERROR_INVALID_PARAMETERS;
    return VX_SUCCESS;
}

vx_status vxReadMatrix(vx_matrix mat, void *array) { return vxCopyMatrix(mat, array, VX_READ_ONLY, VX_MEMORY_TYPE_HOST); }
vx_status vxWriteMatrix(vx_matrix mat, const void *array)
{
    return vxCopyMatrix(mat, const_cast<void *>(array), VX_WRITE_ONLY, VX_MEMORY_TYPE_HOST);
}

// ---- virtual objects -----------------------------------------------------------------------------
vx_array vxCreateVirtualArray(vx_graph graph, vx_enum item_type, vx_size capacity)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    return vxCreateArray(g->ctx, item_type, capacity);
}

vx_image vxCreateVirtualImage(vx_graph graph, vx_uint32 width, vx_uint32 height, vx_df_image color)
{
    auto *g = as<_vx_graph>(graph, VX_TYPE_GRAPH);
    if (!g) return nullptr;
    if (color != VX_DF_IMAGE_U8 && color != VX_DF_IMAGE_RGBX)
        return make_error<vx_image>(g->ctx, VX_ERROR_INVALID_FORMAT, "vxCreateVirtualImage: only U8 and RGBX virtual images are supported");
    if ((width == 0) != (height == 0)) return make_error<vx_image>(g->ctx, VX_ERROR_INVALID_DIMENSION, "vxCreateVirtualImage: both or neither dimension may be 0");
    _vx_image *im = new_image(g->ctx, width, height, color);
    im->is_virtual = true;
    return im;
}

// ---- immediate mode ----------------------------------------------------------------------------
static vx_status immediate_begin(vx_context context)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    if (context->no_device) return (vx_status)NVX_ERROR_NO_GPU;
    return hip_status(hipSetDevice(context->device));
}

vx_status vxuColorConvert(vx_context context, vx_image input, vx_image output)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    std::lock_guard<std::mutex> imm(context->imm_mu);
    vx_status s = immediate_begin(context);
    if (s != VX_SUCCESS) return s;
    s = do_color_convert(as<_vx_image>(input, VX_TYPE_IMAGE), as<_vx_image>(output, VX_TYPE_IMAGE), context->stream);
    return s == VX_SUCCESS ? hip_status(hipStreamSynchronize(context->stream)) : s;
}

vx_status nvxuFindHomography(vx_context context, vx_array srcPoints, vx_array dstPoints, vx_matrix homography, vx_enum method,
                             vx_float32 threshold, vx_uint32 maxEstimateIters, vx_uint32 maxRefineIters, vx_float32 confidence,
                             vx_float32 outlierRatio, vx_array mask)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    std::lock_guard<std::mutex> imm(context->imm_mu);
    vx_status s = immediate_begin(context);
    if (s != VX_SUCCESS) return s;
    auto *m = as<_vx_matrix>(homography, VX_TYPE_MATRIX);
    auto *k = mask ? as<_vx_array>(mask, VX_TYPE_ARRAY) : nullptr;
    if (!m || (mask && !k)) return VX_ERROR_INVALID_REFERENCE;
    s = immediate_scratch(context, homography_scratch_bytes(homography_hypotheses(confidence, outlierRatio, maxEstimateIters)));
    if (s != VX_SUCCESS) return s;
    // the context's scratch buffer is shared with the other immediate-mode calls: clear the ticket word every time
    if (hipMemsetAsync(context->scratch, 0, 16, context->stream) != hipSuccess) return VX_FAILURE;
    s = do_find_homography(as<_vx_array>(srcPoints, VX_TYPE_ARRAY), as<_vx_array>(dstPoints, VX_TYPE_ARRAY), m, method, threshold,
                           maxEstimateIters, maxRefineIters, confidence, outlierRatio, k, context->scratch, context->scratch_size,
                           context->stream);
    if (s != VX_SUCCESS) return s;
    if (hipMemcpyAsync(m->host.data(), m->dev, m->host.size(), hipMemcpyDeviceToHost, context->stream) != hipSuccess) return VX_FAILURE;
    if (k) return refresh_count_now(k, context->stream);
    return hip_status(hipStreamSynchronize(context->stream));
}

vx_status vxuWarpPerspective(vx_context context, vx_image input, vx_matrix matrix, vx_enum type, vx_image output)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    std::lock_guard<std::mutex> imm(context->imm_mu);
    vx_status s = immediate_begin(context);
    if (s != VX_SUCCESS) return s;
    vx_border_t b{};
    b.mode = VX_BORDER_UNDEFINED;
    s = do_warp_perspective(as<_vx_image>(input, VX_TYPE_IMAGE), as<_vx_matrix>(matrix, VX_TYPE_MATRIX), type,
                            as<_vx_image>(output, VX_TYPE_IMAGE), b, context->stream);
    return s == VX_SUCCESS ? hip_status(hipStreamSynchronize(context->stream)) : s;
}

vx_status nvxuCopyImage(vx_context context, vx_image src, vx_image dst)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    std::lock_guard<std::mutex> imm(context->imm_mu);
    vx_status s = immediate_begin(context);
    if (s != VX_SUCCESS) return s;
    s = do_copy_image(as<_vx_image>(src, VX_TYPE_IMAGE), as<_vx_image>(dst, VX_TYPE_IMAGE), context->stream);
    return s == VX_SUCCESS ? hip_status(hipStreamSynchronize(context->stream)) : s;
}

vx_status vxuGaussianPyramid(vx_context context, vx_image input, vx_pyramid gaussian)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    std::lock_guard<std::mutex> imm(context->imm_mu);
    vx_status s = immediate_begin(context);
    if (s != VX_SUCCESS) return s;
    s = do_gaussian_pyramid(as<_vx_image>(input, VX_TYPE_IMAGE), as<_vx_pyramid>(gaussian, VX_TYPE_PYRAMID), context->stream);
    return s == VX_SUCCESS ? hip_status(hipStreamSynchronize(context->stream)) : s;
}

vx_status vxuOpticalFlowPyrLK(vx_context context, vx_pyramid old_images, vx_pyramid new_images, vx_array old_points,
                              vx_array new_points_estimates, vx_array new_points, vx_enum termination, vx_scalar epsilon,
                              vx_scalar num_iterations, vx_scalar use_initial_estimate, vx_size window_dimension)
{
    if (!usable(context)) return VX_ERROR_INVALID_REFERENCE;
    std::lock_guard<std::mutex> imm(context->imm_mu);
    vx_status s = immediate_begin(context);
    if (s != VX_SUCCESS) return s;
    auto *out = as<_vx_array>(new_points, VX_TYPE_ARRAY);
    s = do_pyrlk(as<_vx_pyramid>(old_images, VX_TYPE_PYRAMID), as<_vx_pyramid>(new_images, VX_TYPE_PYRAMID),
                 as<_vx_array>(old_points, VX_TYPE_ARRAY), as<_vx_array>(new_points_estimates, VX_TYPE_ARRAY), out,
                 termination, as<_vx_scalar>(epsilon, VX_TYPE_SCALAR), as<_vx_scalar>(num_iterations, VX_TYPE_SCALAR),
                 as<_vx_scalar>(use_initial_estimate, VX_TYPE_SCALAR), window_dimension, context->stream);
    return s == VX_SUCCESS ? refresh_count_now(out, context->stream) : s;
}

vx_status nvxuHarrisTrack(vx_context context, vx_image input, vx_array output, vx_image mask, vx_array tracked_points,
                          vx_float32 k, vx_float32 threshold, vx_uint32 cell_size, vx_scalar num_corners)
{
    return immediate_corner(context, vsa::kHarris, input, output, mask, tracked_points, k, threshold, 0, cell_size, num_corners);
}

vx_status nvxuFastTrack(vx_context context, vx_image input, vx_array output, vx_image mask, vx_array tracked_points,
                        vx_uint32 type, vx_uint32 threshold, vx_uint32 cell_size, vx_scalar num_corners)
{
    return immediate_corner(context, vsa::kFast, input, output, mask, tracked_points, 0.0f, (float)threshold, type, cell_size, num_corners);
}

void nvxGetVersionInfo(nvx_version_info_t *info)
{
    if (!info) return;
    memset(info, 0, sizeof(*info));
    info->visionworks_version.major = 0;
    info->visionworks_version.minor = 1;
    info->visionworks_version.patch = 0;
    snprintf(info->visionworks_version.suffix, sizeof(info->visionworks_version.suffix), "-amd-gfx950");
    info->openvx_major_version = 1;
    info->openvx_minor_version = 1;
    info->openvx_patch_version = 0;
}

}  // extern "C"

/* not part of the reference API: execution order chosen by vxVerifyGraph (kernel kinds, for tests) */
extern "C" int visualslam_amd_graph_order(vx_graph graph, int *kinds, int max_kinds)
{
    if (!valid(graph, VX_TYPE_GRAPH) || !graph->verified) return -1;
    int n = 0;
    for (_vx_node *nd : graph->order) {
        if (n < max_kinds) kinds[n] = (int)nd->kind;
        ++n;
    }
    return n;
}

/* not part of the reference API: what vxVerifyGraph decided to fuse (for tests): out[0] = ColorConvert nodes converted inside the
 * pyramid's first launch, out[1] = detector nodes whose cell pass rides in the PyrLK launch, out[2] = virtual grey images made
 * views of a pyramid's level 0, out[3] (when max_out >= 4) = FindHomography nodes sharing a launch with a detector node's merge pass */
extern "C" int visualslam_amd_graph_fusions(vx_graph graph, int *out, int max_out)
{
    if (!valid(graph, VX_TYPE_GRAPH) || !graph->verified || !out || max_out < 3) return -1;
    out[0] = out[1] = 0;
    int pairs = 0;
    for (_vx_node *nd : graph->nodes) {
        if (nd->fused_away) ++out[0];
        if (nd->fused_cells) ++out[1];
        if (nd->pair && nd->kind == K_FIND_HOMOGRAPHY) ++pairs;
    }
    out[2] = (int)graph->aliased.size();
    if (max_out < 4) return 3;
    out[3] = pairs;
    return 4;
}
