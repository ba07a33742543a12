// ovx_internal.hpp -- what the translation units of the OpenVX-subset runtime share: the object model behind the opaque
// handles of include/VX/vx.h, and the internal functions each unit offers the others.  Nothing here is part of the C ABI.
//   ovx_objects.cpp   reference counting, error objects, device memory of the data objects, creation and destruction
//   ovx_verify.cpp    vxVerifyGraph: dependency analysis, execution order, node fusions, user-kernel validators
//   ovx_exec.cpp      node execution (the launches of kernels.hpp), graph enqueue / completion, pinned read-back mirrors
//   ovx_runtime.cpp   the C API (extern "C") on top of the three
#pragma once

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

constexpr uint32_t kMagic = 0x4f565831u;   // "1XVO"
constexpr uint32_t kPitchAlign = 256;
constexpr size_t kRowSlack = 256;          // kernels may read whole aligned dwords past a row end

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
    // A FindHomography node and a detector node whose merge pass may share ONE launch (vxVerifyGraph): each is the other's `pair`;
    // the one that comes first in the order leads (launches for both), the other then finds pair_done set and does nothing.
    _vx_node *pair = nullptr;
    bool pair_leader = false;
    bool pair_done = false;
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

namespace ovxrt {

// ---- reference helpers -------------------------------------------------------------------------
inline bool valid(const _vx_reference *r) { return r && r->magic == kMagic; }
inline bool valid(const _vx_reference *r, vx_enum type) { return valid(r) && r->type == type && r->status == VX_SUCCESS; }
template <typename T>
T *as(vx_reference r, vx_enum type) { return valid(r, type) ? static_cast<T *>(r) : nullptr; }

_vx_reference *error_object(_vx_context *ctx, vx_status st);
void log_msg(_vx_context *ctx, _vx_reference *ref, vx_status st, const char *msg);
void release_ref(_vx_reference *r);

template <typename H>
H make_error(_vx_context *ctx, vx_status st, const char *why = nullptr)
{
    if (why) log_msg(ctx, nullptr, st, why);
    return reinterpret_cast<H>(error_object(ctx, st));
}

inline void retain(_vx_reference *r) { if (r) r->refs.fetch_add(1, std::memory_order_relaxed); }
inline bool usable(_vx_context *ctx) { return valid(ctx, VX_TYPE_CONTEXT); }
inline vx_status hip_status(hipError_t e) { return e == hipSuccess ? VX_SUCCESS : VX_FAILURE; }
inline vsa::ImageView view_of(const _vx_image *im) { return vsa::ImageView{ im->dev, im->pitch, im->width, im->height }; }
inline _vx_reference *R(void *p) { return static_cast<_vx_reference *>(p); }

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

template <typename T>
T *param(_vx_node *n, size_t i, vx_enum type) { return i < n->params.size() ? as<T>(n->params[i], type) : nullptr; }

// vxOpticalFlowPyrLKNode publishes termination (index 5, VX_TYPE_ENUM) and window_dimension (index 9, VX_TYPE_SIZE) as scalar
// parameters: they are read where they are used, so a vxSetParameterByIndex on them takes effect like on any other scalar
inline vx_enum pyrlk_termination(_vx_node *n) { auto *s = param<_vx_scalar>(n, 5, VX_TYPE_SCALAR); return s ? s->v.e : 0; }
inline vx_size pyrlk_window(_vx_node *n) { auto *s = param<_vx_scalar>(n, 9, VX_TYPE_SCALAR); return s ? s->v.sz : 0; }

inline float scalar_f32(_vx_node *n, size_t i) { auto *s = param<_vx_scalar>(n, i, VX_TYPE_SCALAR); return s ? s->v.f32 : 0.0f; }
inline uint32_t scalar_u32(_vx_node *n, size_t i) { auto *s = param<_vx_scalar>(n, i, VX_TYPE_SCALAR); return s ? s->v.u32 : 0u; }


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


// objects (ovx_objects.cpp)
void log_msg(_vx_context *ctx, _vx_reference *ref, vx_status st, const char *msg);
_vx_reference *error_object(_vx_context *ctx, vx_status st);
void track(_vx_context *ctx, _vx_reference *r);
void release_ref(_vx_reference *r);
void destroy(_vx_reference *r);
vx_status ensure_image(_vx_image *im);
vx_status ensure_pyramid(_vx_pyramid *p);
vx_status ensure_array(_vx_array *a);
vx_status ensure_matrix(_vx_matrix *m);
size_t item_size_of(vx_enum t);
_vx_image *new_image(_vx_context *ctx, uint32_t w, uint32_t h, vx_df_image fmt);
_vx_pyramid *new_pyramid(_vx_context *ctx, size_t levels, float scale, uint32_t w, uint32_t h, vx_df_image fmt);
_vx_array *new_array(_vx_context *ctx, vx_enum item_type, size_t capacity);
_vx_scalar *new_scalar(_vx_context *ctx, vx_enum type, const void *ptr);
_vx_matrix *new_matrix(_vx_context *ctx, vx_enum data_type, size_t cols, size_t rows);
vx_status refresh_count_now(_vx_array *a, hipStream_t stream);

// node execution (ovx_exec.cpp)
bool is_level0(const _vx_image *gray, const _vx_pyramid *pyr);
vx_status pyramid_view(_vx_pyramid *p, vsa::PyramidView *v);
vx_status do_color_convert(_vx_image *in, _vx_image *out, hipStream_t stream);
vx_status do_copy_image(_vx_image *in, _vx_image *out, hipStream_t stream);
vx_status do_gaussian_pyramid(_vx_image *in, _vx_pyramid *pyr, hipStream_t stream);
vx_status do_color_pyramid(_vx_image *rgbx, _vx_image *gray, _vx_pyramid *pyr, hipStream_t stream);
bool item_kind(vx_enum t, vsa::ItemKind *k);
vx_status do_pyrlk(_vx_pyramid *oldp, _vx_pyramid *newp, _vx_array *oldpts, _vx_array *est, _vx_array *out,
                   vx_enum term, _vx_scalar *eps, _vx_scalar *iters, _vx_scalar *use_est, vx_size win, hipStream_t stream);
vx_status do_corner_track(vsa::Detector det, _vx_image *in, _vx_array *out, _vx_image *mask, _vx_array *tracked,
                          float k, float thr, uint32_t fast_n, uint32_t cell, size_t *dev_num_corners,
                          uint64_t *num_corners_mirror, void *scratch, size_t scratch_size, hipStream_t stream, bool merge_only = false);
vx_status do_find_homography(_vx_array *src, _vx_array *dst, _vx_matrix *mat, vx_enum method, float threshold,
                             uint32_t max_estimate, uint32_t max_refine, float confidence, float outlier_ratio,
                             _vx_array *mask, void *scratch, size_t scratch_size, hipStream_t stream);
vx_status do_warp_perspective(_vx_image *in, _vx_matrix *mat, vx_enum interp, _vx_image *out, const vx_border_t &border,
                              hipStream_t stream);
vx_status run_node(_vx_node *n, hipStream_t stream);
void update_perf(vx_perf_t *p, uint64_t ns);
vx_status sync_host_view(_vx_graph *g);
vx_status enqueue_graph(_vx_graph *g);
vx_status finish_graph(_vx_graph *g);

// graph verification (ovx_verify.cpp)
void keys_of(_vx_reference *r, std::vector<const void *> *out);
bool intersects(const std::vector<const void *> &a, const std::vector<const void *> &b);
vx_status verify_graph(_vx_graph *g);
bool written_in_graph(_vx_graph *g, _vx_reference *r);
vx_status query_any(_vx_reference *r, vx_enum attribute, void *ptr, vx_size size);

}  // namespace ovxrt
