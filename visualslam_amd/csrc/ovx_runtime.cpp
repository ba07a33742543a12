// ovx_runtime.cpp -- the OpenVX-subset object/graph runtime behind include/VX/vx.h, VX/vxu.h and
// NVX/nvx.h (layer L2 of SURVEY.md section 1): reference-counted data objects living in HBM,
// delays whose slots keep their index while the data ages, graphs that order their nodes by data
// dependency at verify time and enqueue them on the graph's own HIP stream, per-node HIP-event
// performance counters, error objects, log callback.
//
// It replaces the closed libvisionworks runtime for the call sequence of
// demos/feature_tracker/feature_tracker.cpp (cited per function in the headers).  All compute is
// the HIP kernels of this library (kernels.hpp); there is no host fallback: without a GPU
// vxCreateContext returns a context whose status is NVX_ERROR_NO_GPU.
#include <NVX/nvx.h>
#include <VX/vx.h>
#include <VX/vxu.h>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace {
constexpr uint32_t kMagic = 0x4f565831u;   // "1XVO"
constexpr uint32_t kPitchAlign = 256;
constexpr size_t kRowSlack = 256;          // kernels may read whole aligned dwords past a row end
}

// ------------------------------------------------------------------------------------------------
// object model
// ------------------------------------------------------------------------------------------------
struct _vx_reference {
    uint32_t magic = kMagic;
    vx_enum type = VX_TYPE_INVALID;
    _vx_context *ctx = nullptr;
    std::atomic<int> refs{1};         // graphs of different camera streams run on different host threads and share objects (frames)
    vx_status status = VX_SUCCESS;    // != SUCCESS only for error objects / a GPU-less context
    _vx_delay *delay = nullptr;       // owning delay, if this object is a delay slot
};

struct MapRecord {
    _vx_reference *obj;
    vx_enum usage, mem_type;
    void *host;
    size_t x0, y0, w, h;              // image rect (bytes are w*bpp) or array range [x0, x0+w)
};

struct _vx_kernel;
struct _vx_context : _vx_reference {
    int device = -1;
    bool no_device = false;           // host-logic-only mode (tests): objects exist, nothing can run
    hipStream_t stream = nullptr;     // immediate-mode work
    bool perf = false;
    bool logging = true;
    vx_log_callback_f log_cb = nullptr;
    std::recursive_mutex mu;
    std::set<_vx_reference *> live;
    std::map<vx_status, _vx_reference *> errors;
    std::map<vx_map_id, MapRecord> maps;
    vx_map_id next_map = 1;
    uint32_t *pinned = nullptr;       // 16 dwords of pinned host memory for count read-backs
    void *scratch = nullptr;          // immediate-mode detector scratch
    size_t scratch_size = 0;
    std::mutex imm_mu;                // immediate-mode calls share `stream`, `pinned` and `scratch`: one at a time, from
                                      // the first launch to the last read-back (camera streams may run init() on their own threads)
    std::map<std::string, _vx_kernel *> kernels;   // finalized user kernels by name (the registry holds a reference)
    vx_enum next_kernel_id = VX_KERNEL_BASE(VX_ID_DEFAULT, 0) + 0x100;
};

struct _vx_image : _vx_reference {
    bool is_virtual = false;          // vxCreateVirtualImage: dimensions may be 0 until vxVerifyGraph infers them
    uint32_t width = 0, height = 0, bpp = 1;
    vx_df_image format = VX_DF_IMAGE_VIRT;
    uint8_t *dev = nullptr;
    uint32_t pitch = 0;
    _vx_pyramid *pyramid = nullptr;   // set for pyramid levels (dependency analysis)
    _vx_image *parent = nullptr;      // vxCreateImageFromROI: a view of the parent's memory at (roi_x, roi_y)
    uint32_t roi_x = 0, roi_y = 0;
    bool has_fill = false;            // vxCreateUniformImage: written with `fill` when the memory is allocated
    uint32_t fill = 0;
};

struct _vx_pyramid : _vx_reference {
    std::vector<_vx_image *> levels;
    vx_float32 scale = 0.5f;
    uint32_t width = 0, height = 0;
    vx_df_image format = VX_DF_IMAGE_U8;
};

struct _vx_array : _vx_reference {
    vx_enum item_type = 0;
    size_t item_size = 0, capacity = 0;
    uint8_t *dev = nullptr;           // items, then a 16-byte tail holding the device-resident u32 count
    uint32_t *dev_count = nullptr;
    size_t num_items = 0;             // host mirror, refreshed after every operation that writes the array
    uint32_t *count_mirror = nullptr; // set while a graph is enqueued: pinned slot the writing kernel also stores the count to
    bool mirror_pending = false;      // a kernel of the running graph has been enqueued that writes the count to count_mirror
};

struct _vx_scalar : _vx_reference {
    vx_enum data_type = 0;
    union { vx_char c; vx_int8 i8; vx_uint8 u8; vx_int16 i16; vx_uint16 u16; vx_int32 i32; vx_uint32 u32;
            vx_int64 i64; vx_uint64 u64; vx_float32 f32; vx_float64 f64; vx_enum e; vx_size sz; vx_df_image df;
            vx_bool b; } v{};
};

struct _vx_matrix : _vx_reference {
    vx_enum data_type = 0;
    size_t cols = 0, rows = 0;
    std::vector<uint8_t> host;        // the values as the application sees them (row-major)
    void *dev = nullptr;              // device copy, allocated when a node first writes or reads it
    float *mirror = nullptr;          // set while a graph is enqueued: pinned slots the writing kernel also stores to
    bool mirror_pending = false;      // a kernel of the running graph has been enqueued that writes the values to `mirror`
};

struct _vx_delay : _vx_reference {
    vx_enum slot_type = 0;
    std::vector<_vx_reference *> slots;   // slots[i] <-> index -i
};

enum KernelKind { K_COLOR_CONVERT, K_GAUSSIAN_PYRAMID, K_PYRLK, K_HARRIS_TRACK, K_FAST_TRACK, K_COPY_IMAGE, K_FIND_HOMOGRAPHY,
                  K_WARP_PERSPECTIVE, K_USER };

struct ParamSpec { vx_enum dir; vx_enum type; bool optional; };

// a user kernel: host callbacks the graph runs between its device nodes (video_stabilizer/*_node.cpp)
struct _vx_kernel : _vx_reference {
    std::string name;
    vx_enum enumeration = 0;
    vx_kernel_f func = nullptr;
    vx_kernel_validate_f validate = nullptr;
    vx_kernel_input_validate_f in_validate = nullptr;      // OpenVX 1.0 form (vxAddKernel)
    vx_kernel_output_validate_f out_validate = nullptr;
    vx_kernel_initialize_f init = nullptr;
    vx_kernel_deinitialize_f deinit = nullptr;
    std::vector<ParamSpec> spec;
    std::vector<bool> spec_set;
    bool finalized = false;
};

struct _vx_parameter : _vx_reference {
    _vx_node *node = nullptr;      // retained
    vx_uint32 index = 0;
};

// what a validator says about one output: attribute -> value bytes, compared with the bound object at vxVerifyGraph
struct _vx_meta_format : _vx_reference {
    std::map<vx_enum, std::vector<uint8_t>> attrs;
};

struct _vx_node : _vx_reference {
    _vx_graph *graph = nullptr;
    KernelKind kind = K_COLOR_CONVERT;
    std::vector<_vx_reference *> params;
    std::vector<ParamSpec> spec;
    vx_perf_t perf{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool fused_away = false;          // K_COLOR_CONVERT: its work is done by the GaussianPyramid node it feeds (set by vxVerifyGraph)
    _vx_node *fused_cvt = nullptr;    // K_GAUSSIAN_PYRAMID: the ColorConvert node whose RGBX input this node converts itself
    _vx_node *fused_cells = nullptr;  // K_PYRLK: the HarrisTrack node whose cell pass may run inside this node's launch (vxVerifyGraph)
    bool cells_done = false;          // K_HARRIS_TRACK: this execution's cell pass has run in the PyrLK node's launch; merge only
    bool perf_enabled = false;        // vxDirective on this node alone (extension: narrower than the context-wide switch)
    void *scratch = nullptr;
    size_t scratch_size = 0;
    size_t *dev_num_corners = nullptr;
    uint64_t *num_corners_mirror = nullptr;   // pinned slot the merge kernel also stores num_corners to (set per enqueue)
    bool corners_pending = false;
    vx_status last_status = VX_SUCCESS;
    _vx_kernel *kernel = nullptr;             // K_USER: the kernel (retained)
    bool user_initialized = false;            // K_USER: the kernel's initializer has run for the current parameters
    vx_border_t border{};                     // K_WARP_PERSPECTIVE: constant border value (mode UNDEFINED = constant 0)
};

struct _vx_graph : _vx_reference {
    std::vector<_vx_node *> nodes, order;
    bool verified = false;
    vx_enum state = VX_GRAPH_STATE_UNVERIFIED;
    hipStream_t stream = nullptr;
    vx_perf_t perf{};
    bool perf_enabled = false;            // vxDirective on this graph alone
    std::string options;
    std::vector<_vx_array *> written;     // arrays whose counts are read back after a run
    std::vector<_vx_node *> corner_nodes; // nodes with a num_corners scalar to refresh
    std::vector<_vx_matrix *> matrices;   // matrices written by nodes: values come back through pinned slots
    std::vector<_vx_image *> aliased;     // virtual images this graph's verification made views of a pyramid's level 0
    uint32_t *pinned = nullptr;           // count read-back staging (pinned)
    size_t pinned_slots = 0;
    bool scheduled = false;
    std::chrono::steady_clock::time_point t0;
};

namespace {

using namespace vsa;

// ---- reference helpers -------------------------------------------------------------------------
inline bool valid(const _vx_reference *r) { return r && r->magic == kMagic; }
inline bool valid(const _vx_reference *r, vx_enum type) { return valid(r) && r->type == type && r->status == VX_SUCCESS; }

template <typename T>
T *as(vx_reference r, vx_enum type) { return valid(r, type) ? static_cast<T *>(r) : nullptr; }

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

template <typename H>
H make_error(_vx_context *ctx, vx_status st, const char *why = nullptr)
{
    if (why) log_msg(ctx, nullptr, st, why);
    return reinterpret_cast<H>(error_object(ctx, st));
}

void track(_vx_context *ctx, _vx_reference *r)
{
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->live.insert(r);
}

void retain(_vx_reference *r) { if (r) r->refs.fetch_add(1, std::memory_order_relaxed); }
void release_ref(_vx_reference *r);

bool usable(_vx_context *ctx) { return valid(ctx, VX_TYPE_CONTEXT); }

vx_status hip_status(hipError_t e) { return e == hipSuccess ? VX_SUCCESS : VX_FAILURE; }

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

template <typename H>
vx_status release_handle(H *h, vx_enum type)
{
    if (!h) return VX_ERROR_INVALID_REFERENCE;
    if (!*h) return VX_SUCCESS;                       // releasing NULL is harmless (feature_tracker.cpp:254-270)
    _vx_reference *r = reinterpret_cast<_vx_reference *>(*h);
    if (!valid(r)) return VX_ERROR_INVALID_REFERENCE;
    if (r->type == VX_TYPE_ERROR) { *h = nullptr; return VX_SUCCESS; }
    if (r->type != type) return VX_ERROR_INVALID_REFERENCE;
    release_ref(r);
    *h = nullptr;
    return VX_SUCCESS;
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

// ---- node execution ----------------------------------------------------------------------------
ImageView view_of(const _vx_image *im) { return ImageView{ im->dev, im->pitch, im->width, im->height }; }

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
                          uint64_t *num_corners_mirror, void *scratch, size_t scratch_size, hipStream_t stream, bool merge_only = false)
{
    CornerTrackParams p{};
    const vx_status s = prep_corner_track(det, in, out, mask, tracked, k, thr, fast_n, cell, dev_num_corners, num_corners_mirror,
                                          scratch, scratch_size, &p);
    if (s != VX_SUCCESS) return s;
    return hip_status(merge_only ? launch_corner_merge(p, stream) : launch_corner_track(p, stream));
}

vx_status do_find_homography(_vx_array *src, _vx_array *dst, _vx_matrix *mat, vx_enum method, float threshold,
                             uint32_t max_estimate, uint32_t max_refine, float confidence, float outlier_ratio,
                             _vx_array *mask, void *scratch, size_t scratch_size, hipStream_t stream)
{
    if (!src || !dst || !mat) return VX_ERROR_INVALID_PARAMETERS;
    if (method != NVX_FIND_HOMOGRAPHY_METHOD_RANSAC) return VX_ERROR_NOT_SUPPORTED;
    if (src->item_type != NVX_TYPE_POINT2F || dst->item_type != NVX_TYPE_POINT2F) return VX_ERROR_INVALID_TYPE;
    if (mat->data_type != VX_TYPE_FLOAT32 || mat->cols != 3 || mat->rows != 3) return VX_ERROR_INVALID_DIMENSION;
    if (mask && mask->item_type != VX_TYPE_UINT8) return VX_ERROR_INVALID_TYPE;
    if (!(threshold > 0.0f)) return VX_ERROR_INVALID_VALUE;
    HomographyParams p{};
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
    return hip_status(launch_find_homography(p, stream));
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

vx_status sync_host_view(_vx_graph *g);

template <typename T>
T *param(_vx_node *n, size_t i, vx_enum type) { return i < n->params.size() ? as<T>(n->params[i], type) : nullptr; }

// vxOpticalFlowPyrLKNode publishes termination (index 5, VX_TYPE_ENUM) and window_dimension (index 9, VX_TYPE_SIZE) as scalar
// parameters: they are read where they are used, so a vxSetParameterByIndex on them takes effect like on any other scalar
vx_enum pyrlk_termination(_vx_node *n) { auto *s = param<_vx_scalar>(n, 5, VX_TYPE_SCALAR); return s ? s->v.e : 0; }
vx_size pyrlk_window(_vx_node *n) { auto *s = param<_vx_scalar>(n, 9, VX_TYPE_SCALAR); return s ? s->v.sz : 0; }

float scalar_f32(_vx_node *n, size_t i) { auto *s = param<_vx_scalar>(n, i, VX_TYPE_SCALAR); return s ? s->v.f32 : 0.0f; }
uint32_t scalar_u32(_vx_node *n, size_t i) { auto *s = param<_vx_scalar>(n, i, VX_TYPE_SCALAR); return s ? s->v.u32 : 0u; }

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

vx_status run_node(_vx_node *n, hipStream_t stream)
{
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

// ---- dependency analysis -----------------------------------------------------------------------
// keys an object stands for: itself, plus -- for pyramids and their levels -- each other
void keys_of(_vx_reference *r, std::vector<const void *> *out)
{
    if (!valid(r)) return;
    out->push_back(r);
    if (r->type == VX_TYPE_PYRAMID) {
        for (auto *l : static_cast<_vx_pyramid *>(r)->levels) out->push_back(l);
    } else if (r->type == VX_TYPE_IMAGE) {
        auto *im = static_cast<_vx_image *>(r);
        if (im->pyramid) out->push_back(im->pyramid);
        for (_vx_image *p = im->parent; p; p = p->parent) out->push_back(p);     // a view stands for its parent too
    } else if (r->type == VX_TYPE_DELAY) {                                        // a delay stands for all of its slots
        for (auto *sl : static_cast<_vx_delay *>(r)->slots) keys_of(sl, out);
    }
}

bool intersects(const std::vector<const void *> &a, const std::vector<const void *> &b)
{
    for (auto x : a) for (auto y : b) if (x == y) return true;
    return false;
}

// the attribute of whatever object `r` is, through the public query of its type
vx_status query_any(_vx_reference *r, vx_enum attribute, void *ptr, vx_size size)
{
    switch (r->type) {
    case VX_TYPE_IMAGE: return vxQueryImage(static_cast<_vx_image *>(r), attribute, ptr, size);
    case VX_TYPE_ARRAY: return vxQueryArray(static_cast<_vx_array *>(r), attribute, ptr, size);
    case VX_TYPE_MATRIX: return vxQueryMatrix(static_cast<_vx_matrix *>(r), attribute, ptr, size);
    case VX_TYPE_SCALAR: return vxQueryScalar(static_cast<_vx_scalar *>(r), attribute, ptr, size);
    case VX_TYPE_PYRAMID: return vxQueryPyramid(static_cast<_vx_pyramid *>(r), attribute, ptr, size);
    case VX_TYPE_DELAY: return vxQueryDelay(static_cast<_vx_delay *>(r), attribute, ptr, size);
    default: return VX_ERROR_NOT_SUPPORTED;
    }
}

// a user node at vxVerifyGraph: the kernel's validator sees the parameters and describes every output in a meta
// format (homography_filter_node.cpp:162-202); what it set must agree with the object bound to that output
vx_status verify_user_node(_vx_node *nd)
{
    _vx_context *ctx = nd->ctx;
    _vx_kernel *k = nd->kernel;
    if (!k || !k->finalized) return VX_ERROR_INVALID_NODE;
    const vx_uint32 num = (vx_uint32)nd->params.size();
    std::vector<_vx_meta_format *> metas(num, nullptr);
    for (auto &m : metas) {
        m = new _vx_meta_format();
        m->type = VX_TYPE_META_FORMAT;
        m->ctx = ctx;
    }
    vx_status st = VX_SUCCESS;
    if (k->validate) st = k->validate(nd, nd->params.data(), num, metas.data());
    for (vx_uint32 i = 0; i < num && st == VX_SUCCESS; ++i) {      // OpenVX 1.0 validators: one call per parameter
        if (!nd->params[i]) continue;                               // an optional parameter left unset
        if (nd->spec[i].dir == VX_INPUT) { if (k->in_validate) st = k->in_validate(nd, i); }
        else if (k->out_validate) st = k->out_validate(nd, i, metas[i]);
    }
    if (st != VX_SUCCESS) {
        std::string msg = "vxVerifyGraph: the validator of user kernel '" + k->name + "' rejected its parameters";
        log_msg(ctx, nd, st, msg.c_str());
    }
    for (vx_uint32 i = 0; i < num && st == VX_SUCCESS; ++i) {
        _vx_reference *r = nd->params[i];
        if (!r || nd->spec[i].dir == VX_INPUT) continue;
        if (r->type == VX_TYPE_IMAGE) {    // a virtual image takes what the validator says
            auto *im = static_cast<_vx_image *>(r);
            auto w = metas[i]->attrs.find(VX_IMAGE_WIDTH), h = metas[i]->attrs.find(VX_IMAGE_HEIGHT);
            if (im->is_virtual && im->width == 0 && im->height == 0 && !im->dev && w != metas[i]->attrs.end() &&
                h != metas[i]->attrs.end() && w->second.size() == 4 && h->second.size() == 4) {
                memcpy(&im->width, w->second.data(), 4);
                memcpy(&im->height, h->second.data(), 4);
            }
        }
        for (auto &kv : metas[i]->attrs) {
            std::vector<uint8_t> have(kv.second.size());
            if (query_any(r, kv.first, have.data(), have.size()) != VX_SUCCESS || have != kv.second) {
                std::string msg = "vxVerifyGraph: an output of user kernel '" + k->name + "' does not match the validator's meta format";
                log_msg(ctx, nd, VX_ERROR_INVALID_PARAMETERS, msg.c_str());
                st = VX_ERROR_INVALID_PARAMETERS;
                break;
            }
        }
    }
    for (auto *m : metas) { m->magic = 0; delete m; }
    if (st != VX_SUCCESS) return st;
    if (k->init && !nd->user_initialized) {
        st = k->init(nd, nd->params.data(), num);
        if (st != VX_SUCCESS) return st;
    }
    nd->user_initialized = true;
    return VX_SUCCESS;
}

// The video stabilizer and the SFM sample convert the frame into a VIRTUAL grey image and build the pyramid from that
// (video_stabilizer/stabilizer.cpp:237-251, sfm/SfM.cpp:980-988) where the feature tracker converts straight into level 0
// (feature_tracker.cpp:360-362).  "Virtual" tells the runtime that the application never accesses the image, so the runtime
// gives it no memory of its own: it becomes a whole-image view of level 0 of the pyramid it feeds.  The GaussianPyramid node
// then has nothing to copy, and the same node fusions apply as in the feature tracker's graph (4 launches per frame instead
// of 7).  The view follows the pyramid object, so it stays right when a delay ages the pyramid's memory.
void alias_virtual_pyramid_inputs(_vx_graph *g)
{
    for (_vx_image *im : g->aliased)          // a re-verification starts over: node parameters may have changed
        if (valid(im) && im->parent) { release_ref(im->parent); im->parent = nullptr; im->dev = nullptr; im->pitch = 0; }
    g->aliased.clear();
    if (getenv("VISUALSLAM_AMD_NO_FUSION") != nullptr) return;
    for (_vx_node *gp : g->nodes) {
        if (gp->kind != K_GAUSSIAN_PYRAMID) continue;
        auto *gray = param<_vx_image>(gp, 0, VX_TYPE_IMAGE);
        auto *pyr = param<_vx_pyramid>(gp, 1, VX_TYPE_PYRAMID);
        if (!gray || !pyr || pyr->levels.empty()) continue;
        _vx_image *l0 = pyr->levels[0];
        if (gray == l0 || !gray->is_virtual || gray->parent || gray->dev || gray->pyramid || gray->has_fill) continue;
        if (gray->format != VX_DF_IMAGE_U8 || l0->format != VX_DF_IMAGE_U8) continue;
        const bool unsized = gray->width == 0 && gray->height == 0;
        if (!unsized && (gray->width != l0->width || gray->height != l0->height)) continue;
        // exactly one node writes the grey image, no other node writes the pyramid, and no other pyramid is built from it
        int gray_writers = 0;
        bool clash = false;
        for (_vx_node *nd : g->nodes)
            for (size_t p = 0; p < nd->params.size(); ++p) {
                _vx_reference *r = nd->params[p];
                if (!r || nd->spec[p].type == VX_TYPE_SCALAR) continue;
                const bool writes = nd->spec[p].dir != VX_INPUT;
                if (r == gray && writes) { ++gray_writers; if (nd->kind != K_COLOR_CONVERT && nd->kind != K_COPY_IMAGE) clash = true; }
                if (r == gray && !writes && nd->kind == K_GAUSSIAN_PYRAMID && nd != gp) clash = true;
                if (nd != gp && writes) {
                    std::vector<const void *> keys, mine{ pyr };
                    keys_of(r, &keys);
                    for (auto *l : pyr->levels) mine.push_back(l);
                    if (intersects(keys, mine)) clash = true;
                }
            }
        if (gray_writers != 1 || clash) continue;
        gray->width = l0->width;
        gray->height = l0->height;
        gray->parent = l0;
        gray->roi_x = gray->roi_y = 0;
        retain(l0);
        g->aliased.push_back(gray);
    }
}

vx_status verify_graph(_vx_graph *g)
{
    _vx_context *ctx = g->ctx;
    const size_t n = g->nodes.size();
    if (n == 0) return VX_ERROR_INVALID_GRAPH;
    // streams and pinned memory created below belong to the context's GPU, whatever device the calling thread last used
    // (a fresh host thread starts on device 0)
    if (!ctx->no_device && hipSetDevice(ctx->device) != hipSuccess) return VX_FAILURE;
    alias_virtual_pyramid_inputs(g);
    std::vector<std::vector<const void *>> rd(n), wr(n);
    for (size_t i = 0; i < n; ++i) {
        _vx_node *nd = g->nodes[i];
        for (size_t p = 0; p < nd->params.size(); ++p) {
            _vx_reference *r = nd->params[p];
            const ParamSpec &sp = nd->spec[p];
            if (!r) {
                if (!sp.optional) {
                    log_msg(ctx, nd, VX_ERROR_INVALID_PARAMETERS, "vxVerifyGraph: a required node parameter is missing");
                    return VX_ERROR_INVALID_PARAMETERS;
                }
                continue;
            }
            if (!valid(r, sp.type)) return VX_ERROR_INVALID_REFERENCE;
            if (sp.type == VX_TYPE_SCALAR) continue;
            if (sp.dir == VX_INPUT || sp.dir == VX_BIDIRECTIONAL) keys_of(r, &rd[i]);
            if (sp.dir == VX_OUTPUT || sp.dir == VX_BIDIRECTIONAL) {
                if (nd->kind == K_COLOR_CONVERT || nd->kind == K_COPY_IMAGE) {   // the image only (not the pyramid it may be level 0 of)
                    wr[i].push_back(r);
                    for (_vx_image *pp = static_cast<_vx_image *>(r)->parent; pp; pp = pp->parent) wr[i].push_back(pp);
                }
                else keys_of(r, &wr[i]);
            }
        }
    }
    // two writers of one object make the order ambiguous
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            std::vector<const void *> wi, wj;
            for (auto k : wr[i]) if (static_cast<const _vx_reference *>(k)->type != VX_TYPE_PYRAMID) wi.push_back(k);
            for (auto k : wr[j]) if (static_cast<const _vx_reference *>(k)->type != VX_TYPE_PYRAMID) wj.push_back(k);
            // ColorConvert writing level 0 and GaussianPyramid "owning" the pyramid is the one legal overlap
            const bool cvt_pyr = (g->nodes[i]->kind == K_GAUSSIAN_PYRAMID) != (g->nodes[j]->kind == K_GAUSSIAN_PYRAMID);
            if (!cvt_pyr && intersects(wi, wj)) {
                log_msg(ctx, g, VX_ERROR_INVALID_GRAPH, "vxVerifyGraph: two nodes write the same object");
                return VX_ERROR_INVALID_GRAPH;
            }
        }
    // Kahn's algorithm, ties broken by creation order.  A GaussianPyramid node whose input is its own
    // level 0 depends on the writer of that level, never on itself.
    std::vector<std::vector<size_t>> succ(n);
    std::vector<int> indeg(n, 0);
    for (size_t a = 0; a < n; ++a)
        for (size_t b = 0; b < n; ++b) {
            if (a == b) continue;
            if (intersects(wr[a], rd[b])) {
                succ[a].push_back(b);
                ++indeg[b];
            }
        }
    g->order.clear();
    std::vector<bool> done(n, false);
    for (size_t it = 0; it < n; ++it) {
        size_t pick = n;
        for (size_t i = 0; i < n; ++i) if (!done[i] && indeg[i] == 0) { pick = i; break; }
        if (pick == n) {
            log_msg(ctx, g, VX_ERROR_INVALID_GRAPH, "vxVerifyGraph: the graph has a cycle");
            return VX_ERROR_INVALID_GRAPH;
        }
        done[pick] = true;
        g->order.push_back(g->nodes[pick]);
        for (size_t s : succ[pick]) --indeg[s];
    }
    // Node fusion (what the callers' "-O3" verify option asks for, feature_tracker.cpp:435-436): ColorConvert writing level 0
    // of a pyramid that a GaussianPyramid node then builds from that same level -- the tracker graphs' first two nodes
    // (feature_tracker.cpp:360-362,386,393) -- run as ONE launch that reads the colour frame.  Legal when nothing reads
    // the grey image between the two nodes; the ColorConvert node then does nothing when executed.
    {
        for (_vx_node *nd : g->nodes) { nd->fused_away = false; nd->fused_cvt = nullptr; nd->fused_cells = nullptr; nd->cells_done = false; }
        std::vector<size_t> pos(n, 0);
        for (size_t k = 0; k < n; ++k)
            for (size_t i = 0; i < n; ++i)
                if (g->nodes[i] == g->order[k]) pos[i] = k;
        for (size_t gi = 0; gi < n && getenv("VISUALSLAM_AMD_NO_FUSION") == nullptr; ++gi) {
            _vx_node *gp = g->nodes[gi];
            if (gp->kind != K_GAUSSIAN_PYRAMID) continue;
            auto *gray = param<_vx_image>(gp, 0, VX_TYPE_IMAGE);
            auto *pyr = param<_vx_pyramid>(gp, 1, VX_TYPE_PYRAMID);
            if (!gray || !pyr || pyr->levels.size() < 2 || !is_level0(gray, pyr)) continue;
            for (size_t ci = 0; ci < n; ++ci) {
                _vx_node *cv = g->nodes[ci];
                if (cv->kind != K_COLOR_CONVERT || param<_vx_image>(cv, 1, VX_TYPE_IMAGE) != gray || pos[ci] > pos[gi]) continue;
                auto *frame = param<_vx_image>(cv, 0, VX_TYPE_IMAGE);
                if (!frame || frame->format != VX_DF_IMAGE_RGBX) continue;
                // every other reader of the grey image runs after the pyramid node, and -- the conversion now happens at the
                // pyramid node's place in the order -- no node between the two writes the colour frame or the grey image
                bool ok = true;
                std::vector<const void *> only_gray{ gray }, moved;
                keys_of(frame, &moved);
                keys_of(gray, &moved);
                for (size_t b = 0; b < n && ok; ++b) {
                    if (b == ci || b == gi) continue;
                    if (intersects(rd[b], only_gray) && pos[b] < pos[gi]) ok = false;
                    if (pos[b] > pos[ci] && pos[b] < pos[gi] && intersects(wr[b], moved)) ok = false;
                }
                if (ok) { cv->fused_away = true; gp->fused_cvt = cv; }
                break;
            }
        }
        // A HarrisTrack / FastTrack node that merges a PyrLK node's output (feature_tracker.cpp:402-430): only its merge pass needs the
        // tracked points; its cell pass reads the image and the mask, which are ready when PyrLK starts if no node from the
        // PyrLK node up to the HarrisTrack node writes them.  The cell pass then runs inside the PyrLK launch, whose last
        // waves otherwise leave compute units idle, and the frame has one launch less.
        for (size_t hi = 0; hi < n && getenv("VISUALSLAM_AMD_NO_FUSION") == nullptr; ++hi) {
            _vx_node *ht = g->nodes[hi];
            if (ht->kind != K_HARRIS_TRACK && ht->kind != K_FAST_TRACK) continue;
            auto *tracked = param<_vx_array>(ht, 3, VX_TYPE_ARRAY);
            if (!tracked) continue;
            for (size_t li = 0; li < n; ++li) {
                _vx_node *lk = g->nodes[li];
                if (lk->kind != K_PYRLK || lk->fused_cells || param<_vx_array>(lk, 4, VX_TYPE_ARRAY) != tracked || pos[li] > pos[hi]) continue;
                std::vector<const void *> inputs;
                keys_of(ht->params[0], &inputs);
                if (ht->params.size() > 2 && ht->params[2]) keys_of(ht->params[2], &inputs);
                bool ok = true;
                for (size_t b = 0; b < n && ok; ++b)
                    if (pos[b] >= pos[li] && pos[b] < pos[hi] && intersects(wr[b], inputs)) ok = false;
                if (ok) lk->fused_cells = ht;
                break;
            }
        }
    }
    // virtual images created with width = height = 0 take their size from the node that writes them
    for (_vx_node *nd : g->order) {
        if (nd->kind != K_COLOR_CONVERT && nd->kind != K_COPY_IMAGE) continue;
        auto *in = param<_vx_image>(nd, 0, VX_TYPE_IMAGE), *out = param<_vx_image>(nd, 1, VX_TYPE_IMAGE);
        if (in && out && out->is_virtual && out->width == 0 && out->height == 0 && !out->dev) {
            out->width = in->width;
            out->height = in->height;
        }
    }
    // metadata checks that need no device
    for (_vx_node *nd : g->order) {
        if (nd->kind == K_USER) {
            vx_status us = verify_user_node(nd);
            if (us != VX_SUCCESS) return us;
        }
        if (nd->kind == K_WARP_PERSPECTIVE) {
            auto *in = param<_vx_image>(nd, 0, VX_TYPE_IMAGE), *out = param<_vx_image>(nd, 3, VX_TYPE_IMAGE);
            auto *m = param<_vx_matrix>(nd, 1, VX_TYPE_MATRIX);
            auto *t = param<_vx_scalar>(nd, 2, VX_TYPE_SCALAR);
            if (!in || !out || in->format != out->format) {
                log_msg(ctx, nd, VX_ERROR_INVALID_FORMAT, "vxWarpPerspectiveNode: input and output must both be U8 or both RGBX");
                return VX_ERROR_INVALID_FORMAT;
            }
            if (out->width == 0 || out->height == 0) return VX_ERROR_INVALID_DIMENSION;
            if (!m || m->data_type != VX_TYPE_FLOAT32 || m->cols != 3 || m->rows != 3) return VX_ERROR_INVALID_DIMENSION;
            if (!t || (t->v.e != VX_INTERPOLATION_BILINEAR && t->v.e != VX_INTERPOLATION_NEAREST_NEIGHBOR)) {
                log_msg(ctx, nd, VX_ERROR_NOT_SUPPORTED, "vxWarpPerspectiveNode: NEAREST_NEIGHBOR and BILINEAR are offered");
                return VX_ERROR_NOT_SUPPORTED;
            }
        }
    }
    for (_vx_node *nd : g->nodes) {
        if (nd->kind == K_FIND_HOMOGRAPHY) {
            auto *a = param<_vx_array>(nd, 0, VX_TYPE_ARRAY), *b = param<_vx_array>(nd, 1, VX_TYPE_ARRAY);
            auto *m = param<_vx_matrix>(nd, 2, VX_TYPE_MATRIX);
            auto *k = param<_vx_array>(nd, 9, VX_TYPE_ARRAY);
            auto *meth = param<_vx_scalar>(nd, 3, VX_TYPE_SCALAR);
            if (!a || !b || a->item_type != NVX_TYPE_POINT2F || b->item_type != NVX_TYPE_POINT2F || (k && k->item_type != VX_TYPE_UINT8)) {
                log_msg(ctx, nd, VX_ERROR_INVALID_TYPE, "nvxFindHomographyNode: points must be NVX_TYPE_POINT2F arrays, the mask a VX_TYPE_UINT8 array");
                return VX_ERROR_INVALID_TYPE;
            }
            if (!m || m->data_type != VX_TYPE_FLOAT32 || m->cols != 3 || m->rows != 3) return VX_ERROR_INVALID_DIMENSION;
            if (!meth || meth->v.e != NVX_FIND_HOMOGRAPHY_METHOD_RANSAC) {
                log_msg(ctx, nd, VX_ERROR_NOT_SUPPORTED, "nvxFindHomographyNode: only NVX_FIND_HOMOGRAPHY_METHOD_RANSAC is implemented");
                return VX_ERROR_NOT_SUPPORTED;
            }
            if (!(scalar_f32(nd, 4) > 0.0f)) return VX_ERROR_INVALID_VALUE;
        }
        if (nd->kind == K_PYRLK) {
            auto *a = param<_vx_array>(nd, 2, VX_TYPE_ARRAY), *c = param<_vx_array>(nd, 4, VX_TYPE_ARRAY);
            ItemKind k;
            if (!a || !c || a->item_type != c->item_type || !item_kind(a->item_type, &k)) {
                log_msg(ctx, nd, VX_ERROR_NOT_SUPPORTED, "vxOpticalFlowPyrLKNode: point arrays must both be NVX_TYPE_POINT2F or NVX_TYPE_KEYPOINTF");
                return VX_ERROR_NOT_SUPPORTED;
            }
            if (pyrlk_window(nd) < (vx_size)kLkMinWin || pyrlk_window(nd) > (vx_size)kLkMaxWin) return VX_ERROR_INVALID_VALUE;
        }
        if (nd->kind == K_HARRIS_TRACK || nd->kind == K_FAST_TRACK) {
            auto *o = param<_vx_array>(nd, 1, VX_TYPE_ARRAY);
            ItemKind k;
            if (!o || !item_kind(o->item_type, &k)) return VX_ERROR_NOT_SUPPORTED;
            if (scalar_u32(nd, 6) == 0) return VX_ERROR_INVALID_VALUE;
        }
    }
    // bookkeeping for the post-run read-backs
    g->written.clear();
    g->corner_nodes.clear();
    g->matrices.clear();
    for (_vx_node *nd : g->order) {
        for (size_t p = 0; p < nd->params.size(); ++p)
            if (nd->kind != K_USER && nd->params[p] && nd->spec[p].type == VX_TYPE_MATRIX && nd->spec[p].dir != VX_INPUT) {
                auto *m = static_cast<_vx_matrix *>(nd->params[p]);
                if (std::find(g->matrices.begin(), g->matrices.end(), m) == g->matrices.end()) g->matrices.push_back(m);
            }
        for (size_t p = 0; p < nd->params.size(); ++p)
            if (nd->params[p] && nd->spec[p].type == VX_TYPE_ARRAY && nd->spec[p].dir != VX_INPUT) {
                auto *a = static_cast<_vx_array *>(nd->params[p]);
                if (std::find(g->written.begin(), g->written.end(), a) == g->written.end()) g->written.push_back(a);
            }
        if ((nd->kind == K_HARRIS_TRACK || nd->kind == K_FAST_TRACK) && param<_vx_scalar>(nd, 7, VX_TYPE_SCALAR))
            g->corner_nodes.push_back(nd);
    }
    if (!ctx->no_device) {
        size_t slots = 2 * (g->written.size() + g->corner_nodes.size() + 1);   // 8 bytes per slot
        for (_vx_matrix *m : g->matrices) slots += 2 * ((m->host.size() + 7) / 8);
        if (slots > g->pinned_slots) {
            if (g->pinned) (void)hipHostFree(g->pinned);
            g->pinned = nullptr;
            if (hipHostMalloc(reinterpret_cast<void **>(&g->pinned), slots * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
                return VX_ERROR_NO_MEMORY;
            g->pinned_slots = slots;
        }
        if (!g->stream) {
            // VISUALSLAM_AMD_STREAM_PRIORITY=<p> (-1 greatest, 1 least; unset = normal) creates the graphs' streams with that
            // priority.  HIP keeps a pool of hardware queues per priority and deals a pool's streams round-robin over its queues:
            // at normal priority the graphs share the pool with the process's other streams (32 camera streams landed 9/7/7/9 on
            // the four queues), with a priority of their own they get 8/8/8/8 -- +2.8 % on the 32-stream bench, the same for -1
            // and 1 (tools/queue_probe.hip, profiles/r02_kernel_experiments.txt).  Opt-in: a priority also ranks the graphs'
            // kernels against the application's other GPU work.
            const char *e = getenv("VISUALSLAM_AMD_STREAM_PRIORITY");
            const hipError_t err = e ? hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, atoi(e))
                                     : hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
            if (err != hipSuccess) return VX_FAILURE;
        }
    }
    g->verified = true;
    g->state = VX_GRAPH_STATE_VERIFIED;
    return VX_SUCCESS;
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

// The pinned slots of a graph, in a fixed order: one per written array (its count), one per corner node with a
// num_corners scalar, a few per written matrix.  The kernels that produce these values also store them there, so the
// host has them at stream completion without a read-back copy.
template <typename FA, typename FN, typename FM>
void walk_mirrors(_vx_graph *g, FA on_array, FN on_node, FM on_matrix)
{
    size_t slot = 0;
    for (_vx_array *a : g->written) { on_array(a, &g->pinned[slot]); slot += 2; }
    for (_vx_node *n : g->corner_nodes) { on_node(n, reinterpret_cast<uint64_t *>(&g->pinned[slot])); slot += 2; }
    for (_vx_matrix *m : g->matrices) { on_matrix(m, reinterpret_cast<float *>(&g->pinned[slot])); slot += 2 * ((m->host.size() + 7) / 8); }
}

// Drains the graph's stream and brings the host's view (array counts, num_corners scalars, matrix values) up to date
// with what the kernels enqueued so far have written.  Called before every host (user-kernel) node and at the end.
vx_status sync_host_view(_vx_graph *g)
{
    if (hipStreamSynchronize(g->stream) != hipSuccess) { g->state = VX_GRAPH_STATE_ABANDONED; return VX_FAILURE; }
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

// does any node of the graph write `r` (or memory `r` stands for: its pyramid, its ROI parents, views of it)?
bool written_in_graph(_vx_graph *g, _vx_reference *r)
{
    if (!g || !r) return false;
    std::vector<const void *> mine, theirs;
    keys_of(r, &mine);
    for (_vx_node *nd : g->nodes)
        for (size_t p = 0; p < nd->params.size(); ++p) {
            if (!nd->params[p] || nd->spec[p].dir == VX_INPUT || nd->spec[p].type == VX_TYPE_SCALAR) continue;
            theirs.clear();
            keys_of(nd->params[p], &theirs);
            if (intersects(mine, theirs)) return true;
        }
    return false;
}

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

_vx_reference *R(void *p) { return static_cast<_vx_reference *>(p); }

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
 * views of a pyramid's level 0 */
extern "C" int visualslam_amd_graph_fusions(vx_graph graph, int *out, int max_out)
{
    if (!valid(graph, VX_TYPE_GRAPH) || !graph->verified || !out || max_out < 3) return -1;
    out[0] = out[1] = 0;
    for (_vx_node *nd : graph->nodes) {
        if (nd->fused_away) ++out[0];
        if (nd->fused_cells) ++out[1];
    }
    out[2] = (int)graph->aliased.size();
    return 3;
}
