/*
 * VX/vx.h -- the OpenVX 1.1 object/graph C API subset that the VisionWorks feature-tracking
 * front-end calls, as exported by libvisualslam_amd (HIP, MI355X).
 *
 * Written from scratch: the reference repository ships only callers of this API (closed
 * libvisionworks), so names, argument order and semantics follow the public Khronos OpenVX 1.1
 * specification and the call sites cited at each declaration (paths relative to
 * /root/reference/VisionWorks/VisionWorks-1.6-Samples/).  Callers recompile against this header;
 * numeric enum values are ours except where callers rely on them (FOURCC image formats,
 * VX_SUCCESS == 0, vx/nvxcu enum equality -- nvxio/src/OVX/Private/TypesOVX.cpp:56,97).
 *
 * Behavioural contract:
 *   - every vxCreate* / vx*Node returns a +1 reference released with vxRelease*(&ref) (sets it to
 *     NULL; releasing NULL is harmless -- demos/feature_tracker/feature_tracker.cpp:254-270);
 *   - creators never return NULL for a valid context: on failure they return an error object
 *     whose vxGetStatus() != VX_SUCCESS (NVXIO_CHECK_REFERENCE, nvxio/include/OVX/UtilityOVX.hpp:160-161);
 *   - vxGetReferenceFromDelay returns a borrowed reference bound to the slot INDEX; vxAgeDelay
 *     moves the data between slots, so nodes bound to delay slots see the aged data with no
 *     re-verification (feature_tracker.cpp:216-224, main_feature_tracker.cpp:245-246,392);
 *   - vxProcessGraph and every vxu / nvxu call are synchronous: results are readable on return.
 */
#ifndef VX_VX_H
#define VX_VX_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>   /* callers rely on it transitively (demos/feature_tracker/feature_tracker.cpp:436 uses strlen) */

#ifdef __cplusplus
extern "C" {
#endif

#define VX_API_ENTRY
#define VX_API_CALL
#define VX_CALLBACK

/* ---- scalar types ------------------------------------------------------------------------ */
typedef char     vx_char;
typedef uint8_t  vx_uint8;
typedef uint16_t vx_uint16;
typedef uint32_t vx_uint32;
typedef uint64_t vx_uint64;
typedef int8_t   vx_int8;
typedef int16_t  vx_int16;
typedef int32_t  vx_int32;
typedef int64_t  vx_int64;
typedef float    vx_float32;
typedef double   vx_float64;
typedef size_t   vx_size;
typedef int32_t  vx_enum;
typedef vx_enum  vx_status;
typedef uint32_t vx_df_image;
typedef uintptr_t vx_map_id;
#define VX_FMT_SIZE "zu"

typedef enum _vx_bool_e { vx_false_e = 0, vx_true_e = 1 } vx_bool;

/* ---- opaque objects ---------------------------------------------------------------------- */
typedef struct _vx_reference *vx_reference;
typedef struct _vx_context   *vx_context;
typedef struct _vx_matrix    *vx_matrix;
typedef struct _vx_graph     *vx_graph;
typedef struct _vx_node      *vx_node;
typedef struct _vx_image     *vx_image;
typedef struct _vx_pyramid   *vx_pyramid;
typedef struct _vx_array     *vx_array;
typedef struct _vx_delay     *vx_delay;
typedef struct _vx_scalar    *vx_scalar;
typedef struct _vx_kernel    *vx_kernel;
typedef struct _vx_meta_format *vx_meta_format;
typedef struct _vx_parameter *vx_parameter;

/* ---- status ------------------------------------------------------------------------------ */
enum vx_status_e {
    VX_SUCCESS = 0,
    VX_FAILURE = -1,
    VX_ERROR_NOT_IMPLEMENTED = -2,
    VX_ERROR_NOT_SUPPORTED = -3,
    VX_ERROR_NOT_SUFFICIENT = -4,
    VX_ERROR_NOT_ALLOCATED = -5,
    VX_ERROR_NOT_COMPATIBLE = -6,
    VX_ERROR_NO_RESOURCES = -7,
    VX_ERROR_NO_MEMORY = -8,
    VX_ERROR_OPTIMIZED_AWAY = -9,
    VX_ERROR_INVALID_PARAMETERS = -10,
    VX_ERROR_INVALID_MODULE = -11,
    VX_ERROR_INVALID_REFERENCE = -12,
    VX_ERROR_INVALID_LINK = -13,
    VX_ERROR_INVALID_FORMAT = -14,
    VX_ERROR_INVALID_DIMENSION = -15,
    VX_ERROR_INVALID_VALUE = -16,
    VX_ERROR_INVALID_TYPE = -17,
    VX_ERROR_INVALID_GRAPH = -18,
    VX_ERROR_INVALID_NODE = -19,
    VX_ERROR_INVALID_SCOPE = -20,
    VX_ERROR_GRAPH_SCHEDULED = -21,
    VX_ERROR_GRAPH_ABANDONED = -22
};

/* ---- type enumeration (numeric values shared with nvxcu_array_item_type_e) --------------- */
enum vx_type_e {
    VX_TYPE_INVALID = 0x000,
    VX_TYPE_CHAR = 0x001, VX_TYPE_INT8 = 0x002, VX_TYPE_UINT8 = 0x003, VX_TYPE_INT16 = 0x004,
    VX_TYPE_UINT16 = 0x005, VX_TYPE_INT32 = 0x006, VX_TYPE_UINT32 = 0x007, VX_TYPE_INT64 = 0x008,
    VX_TYPE_UINT64 = 0x009, VX_TYPE_FLOAT32 = 0x00A, VX_TYPE_FLOAT64 = 0x00B, VX_TYPE_ENUM = 0x00C,
    VX_TYPE_SIZE = 0x00D, VX_TYPE_DF_IMAGE = 0x00E, VX_TYPE_BOOL = 0x010,
    VX_TYPE_RECTANGLE = 0x020, VX_TYPE_KEYPOINT = 0x021, VX_TYPE_COORDINATES2D = 0x022,
    VX_TYPE_COORDINATES3D = 0x023,
    VX_TYPE_USER_STRUCT_START = 0x100,
    VX_TYPE_VENDOR_STRUCT_START = 0x400,
    VX_TYPE_REFERENCE = 0x800, VX_TYPE_CONTEXT = 0x801, VX_TYPE_GRAPH = 0x802, VX_TYPE_NODE = 0x803,
    VX_TYPE_KERNEL = 0x804, VX_TYPE_PARAMETER = 0x805, VX_TYPE_DELAY = 0x806, VX_TYPE_LUT = 0x807,
    VX_TYPE_DISTRIBUTION = 0x808, VX_TYPE_PYRAMID = 0x809, VX_TYPE_THRESHOLD = 0x80A,
    VX_TYPE_MATRIX = 0x80B, VX_TYPE_CONVOLUTION = 0x80C, VX_TYPE_SCALAR = 0x80D, VX_TYPE_ARRAY = 0x80E,
    VX_TYPE_IMAGE = 0x80F, VX_TYPE_REMAP = 0x810, VX_TYPE_ERROR = 0x811, VX_TYPE_META_FORMAT = 0x812
};

/* ---- image formats: FOURCCs (nvxio/src/NVX/Render/EventLogger.cpp:220-222) ---------------- */
#define VX_DF_IMAGE(a, b, c, d) ((vx_uint32)(vx_uint8)(a) | ((vx_uint32)(vx_uint8)(b) << 8) | \
                                 ((vx_uint32)(vx_uint8)(c) << 16) | ((vx_uint32)(vx_uint8)(d) << 24))
enum vx_df_image_e {
    VX_DF_IMAGE_VIRT = VX_DF_IMAGE('V', 'I', 'R', 'T'),
    VX_DF_IMAGE_RGB  = VX_DF_IMAGE('R', 'G', 'B', '2'),
    VX_DF_IMAGE_RGBX = VX_DF_IMAGE('R', 'G', 'B', 'A'),
    VX_DF_IMAGE_NV12 = VX_DF_IMAGE('N', 'V', '1', '2'),
    VX_DF_IMAGE_U8   = VX_DF_IMAGE('U', '0', '0', '8'),
    VX_DF_IMAGE_U16  = VX_DF_IMAGE('U', '0', '1', '6'),
    VX_DF_IMAGE_S16  = VX_DF_IMAGE('S', '0', '1', '6'),
    VX_DF_IMAGE_U32  = VX_DF_IMAGE('U', '0', '3', '2'),
    VX_DF_IMAGE_S32  = VX_DF_IMAGE('S', '0', '3', '2')
};

/* ---- attributes: OpenVX 1.0 spellings (used by the samples) and 1.1 spellings -------------- */
enum vx_image_attribute_e {
    VX_IMAGE_WIDTH = 0x80F00, VX_IMAGE_HEIGHT, VX_IMAGE_FORMAT, VX_IMAGE_PLANES, VX_IMAGE_SPACE,
    VX_IMAGE_RANGE, VX_IMAGE_SIZE, VX_IMAGE_MEMORY_TYPE,
    VX_IMAGE_ATTRIBUTE_WIDTH = VX_IMAGE_WIDTH, VX_IMAGE_ATTRIBUTE_HEIGHT = VX_IMAGE_HEIGHT,
    VX_IMAGE_ATTRIBUTE_FORMAT = VX_IMAGE_FORMAT, VX_IMAGE_ATTRIBUTE_PLANES = VX_IMAGE_PLANES,
    VX_IMAGE_ATTRIBUTE_SPACE = VX_IMAGE_SPACE, VX_IMAGE_ATTRIBUTE_RANGE = VX_IMAGE_RANGE,
    VX_IMAGE_ATTRIBUTE_SIZE = VX_IMAGE_SIZE
};
enum vx_array_attribute_e {
    VX_ARRAY_ITEMTYPE = 0x80E00, VX_ARRAY_NUMITEMS, VX_ARRAY_CAPACITY, VX_ARRAY_ITEMSIZE,
    VX_ARRAY_ATTRIBUTE_ITEMTYPE = VX_ARRAY_ITEMTYPE, VX_ARRAY_ATTRIBUTE_NUMITEMS = VX_ARRAY_NUMITEMS,
    VX_ARRAY_ATTRIBUTE_CAPACITY = VX_ARRAY_CAPACITY, VX_ARRAY_ATTRIBUTE_ITEMSIZE = VX_ARRAY_ITEMSIZE
};
enum vx_pyramid_attribute_e {
    VX_PYRAMID_LEVELS = 0x80900, VX_PYRAMID_SCALE, VX_PYRAMID_WIDTH, VX_PYRAMID_HEIGHT, VX_PYRAMID_FORMAT,
    VX_PYRAMID_ATTRIBUTE_LEVELS = VX_PYRAMID_LEVELS, VX_PYRAMID_ATTRIBUTE_SCALE = VX_PYRAMID_SCALE,
    VX_PYRAMID_ATTRIBUTE_WIDTH = VX_PYRAMID_WIDTH, VX_PYRAMID_ATTRIBUTE_HEIGHT = VX_PYRAMID_HEIGHT,
    VX_PYRAMID_ATTRIBUTE_FORMAT = VX_PYRAMID_FORMAT
};
enum vx_scalar_attribute_e { VX_SCALAR_TYPE = 0x80D00, VX_SCALAR_ATTRIBUTE_TYPE = VX_SCALAR_TYPE };
enum vx_delay_attribute_e {
    VX_DELAY_TYPE = 0x80600, VX_DELAY_SLOTS,
    VX_DELAY_ATTRIBUTE_TYPE = VX_DELAY_TYPE, VX_DELAY_ATTRIBUTE_SLOTS = VX_DELAY_SLOTS
};
enum vx_graph_attribute_e {
    VX_GRAPH_NUMNODES = 0x80200, VX_GRAPH_PERFORMANCE = 0x80202, VX_GRAPH_NUMPARAMETERS, VX_GRAPH_STATE,
    VX_GRAPH_ATTRIBUTE_NUMNODES = VX_GRAPH_NUMNODES, VX_GRAPH_ATTRIBUTE_PERFORMANCE = VX_GRAPH_PERFORMANCE,
    VX_GRAPH_ATTRIBUTE_STATE = VX_GRAPH_STATE
};
enum vx_node_attribute_e {
    VX_NODE_STATUS = 0x80300, VX_NODE_PERFORMANCE, VX_NODE_BORDER, VX_NODE_PARAMETERS = 0x80305,
    VX_NODE_ATTRIBUTE_STATUS = VX_NODE_STATUS, VX_NODE_ATTRIBUTE_PERFORMANCE = VX_NODE_PERFORMANCE,
    VX_NODE_ATTRIBUTE_BORDER = VX_NODE_BORDER
};
enum vx_context_attribute_e {
    VX_CONTEXT_VENDOR_ID = 0x80100, VX_CONTEXT_VERSION, VX_CONTEXT_UNIQUE_KERNELS, VX_CONTEXT_MODULES,
    VX_CONTEXT_REFERENCES, VX_CONTEXT_IMPLEMENTATION,
    VX_CONTEXT_ATTRIBUTE_VENDOR_ID = VX_CONTEXT_VENDOR_ID, VX_CONTEXT_ATTRIBUTE_VERSION = VX_CONTEXT_VERSION,
    VX_CONTEXT_ATTRIBUTE_REFERENCES = VX_CONTEXT_REFERENCES,
    VX_CONTEXT_ATTRIBUTE_IMPLEMENTATION = VX_CONTEXT_IMPLEMENTATION
};
enum vx_reference_attribute_e {
    VX_REFERENCE_COUNT = 0x80000, VX_REFERENCE_TYPE, VX_REFERENCE_NAME,
    VX_REF_ATTRIBUTE_COUNT = VX_REFERENCE_COUNT, VX_REF_ATTRIBUTE_TYPE = VX_REFERENCE_TYPE
};
enum vx_graph_state_e {
    VX_GRAPH_STATE_UNVERIFIED = 0x15000, VX_GRAPH_STATE_VERIFIED, VX_GRAPH_STATE_RUNNING,
    VX_GRAPH_STATE_ABANDONED, VX_GRAPH_STATE_COMPLETED
};

/* ---- misc enums -------------------------------------------------------------------------- */
enum vx_directive_e {
    VX_DIRECTIVE_DISABLE_LOGGING = 0x03000, VX_DIRECTIVE_ENABLE_LOGGING,
    VX_DIRECTIVE_DISABLE_PERFORMANCE, VX_DIRECTIVE_ENABLE_PERFORMANCE
};
enum vx_accessor_e {
    VX_READ_ONLY = 0x11001, VX_WRITE_ONLY, VX_READ_AND_WRITE
};
enum vx_memory_type_e {
    VX_MEMORY_TYPE_NONE = 0x0E000, VX_MEMORY_TYPE_HOST,
    VX_IMPORT_TYPE_NONE = VX_MEMORY_TYPE_NONE, VX_IMPORT_TYPE_HOST = VX_MEMORY_TYPE_HOST
};
enum vx_termination_criteria_e {
    VX_TERM_CRITERIA_ITERATIONS = 0x17000, VX_TERM_CRITERIA_EPSILON, VX_TERM_CRITERIA_BOTH
};
enum vx_direction_e { VX_INPUT = 0x04000, VX_OUTPUT, VX_BIDIRECTIONAL };
enum vx_parameter_state_e { VX_PARAMETER_STATE_REQUIRED = 0x06000, VX_PARAMETER_STATE_OPTIONAL };
enum vx_border_e {
    VX_BORDER_UNDEFINED = 0x0C000, VX_BORDER_CONSTANT, VX_BORDER_REPLICATE,
    VX_BORDER_MODE_UNDEFINED = VX_BORDER_UNDEFINED, VX_BORDER_MODE_CONSTANT = VX_BORDER_CONSTANT,
    VX_BORDER_MODE_REPLICATE = VX_BORDER_REPLICATE
};
enum vx_interpolation_type_e {      /* demos/video_stabilizer/stabilizer.cpp:295 */
    VX_INTERPOLATION_NEAREST_NEIGHBOR = 0x04000 + 0x10000, VX_INTERPOLATION_BILINEAR, VX_INTERPOLATION_AREA,
    VX_INTERPOLATION_TYPE_NEAREST_NEIGHBOR = VX_INTERPOLATION_NEAREST_NEIGHBOR,
    VX_INTERPOLATION_TYPE_BILINEAR = VX_INTERPOLATION_BILINEAR, VX_INTERPOLATION_TYPE_AREA = VX_INTERPOLATION_AREA
};
enum vx_color_space_e { VX_COLOR_SPACE_NONE = 0x09000, VX_COLOR_SPACE_BT601_525, VX_COLOR_SPACE_BT601_625,
                        VX_COLOR_SPACE_BT709, VX_COLOR_SPACE_DEFAULT = VX_COLOR_SPACE_BT709 };
enum vx_channel_range_e { VX_CHANNEL_RANGE_FULL = 0x0A000, VX_CHANNEL_RANGE_RESTRICTED };

#define VX_SCALE_PYRAMID_HALF 0.5f
#define VX_SCALE_PYRAMID_ORB  0.8408964f
#define VX_MAX_LOG_MESSAGE_LEN 1024
#define VX_MAX_KERNEL_NAME 256
#define VX_ID_KHRONOS 0x000
#define VX_ID_DEFAULT 0xFFE
#define VX_VERSION_1_1 ((1 << 16) | 1)
#define VX_VERSION VX_VERSION_1_1

/* ---- plain structs ----------------------------------------------------------------------- */
typedef struct _vx_perf_t {          /* all values in nanoseconds (nvxio/src/OVX/UtilityOVX.cpp:46-68) */
    vx_uint64 tmp, beg, end, sum, avg, min, num, max;
} vx_perf_t;

typedef struct _vx_rectangle_t { vx_uint32 start_x, start_y, end_x, end_y; } vx_rectangle_t;
typedef struct _vx_coordinates2d_t { vx_uint32 x, y; } vx_coordinates2d_t;
typedef struct _vx_coordinates3d_t { vx_uint32 x, y, z; } vx_coordinates3d_t;
typedef struct _vx_keypoint_t {      /* nvxio/src/OpenGLShaders.hpp:586-597 */
    vx_int32 x, y;
    vx_float32 strength, scale, orientation;
    vx_int32 tracking_status;
    vx_float32 error;
} vx_keypoint_t;
typedef struct _vx_imagepatch_addressing_t {
    vx_uint32 dim_x, dim_y;
    vx_int32 stride_x, stride_y;
    vx_uint32 scale_x, scale_y, step_x, step_y;
} vx_imagepatch_addressing_t;
#define VX_SCALE_UNITY 1024u
typedef struct _vx_border_t {
    vx_enum mode;
    union { vx_uint8 U8; vx_int16 S16; vx_uint16 U16; vx_int32 S32; vx_uint32 U32; vx_uint8 RGB[3]; vx_uint8 RGBX[4]; } constant_value;
} vx_border_t;

typedef union _vx_pixel_value_t {    /* demos/video_stabilizer/stabilizer.cpp:399-404 */
    vx_uint8 RGB[3];
    vx_uint8 RGBX[4];
    vx_uint8 YUV[3];
    vx_uint8 U8;
    vx_uint16 U16;
    vx_int16 S16;
    vx_uint32 U32;
    vx_int32 S32;
    vx_uint8 reserved[16];
} vx_pixel_value_t;

typedef void (VX_CALLBACK *vx_log_callback_f)(vx_context context, vx_reference ref, vx_status status,
                                             const vx_char string[]);

/* ---- context ----------------------------------------------------------------------------- */
/* nvxio/include/OVX/UtilityOVX.hpp:271-275.  Binds the calling thread's current HIP device. */
VX_API_ENTRY vx_context VX_API_CALL vxCreateContext(void);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseContext(vx_context *context);
VX_API_ENTRY vx_context VX_API_CALL vxGetContext(vx_reference reference);
VX_API_ENTRY vx_status VX_API_CALL vxQueryContext(vx_context context, vx_enum attribute, void *ptr, vx_size size);
/* nvxio/include/OVX/UtilityOVX.hpp:160-161 */
VX_API_ENTRY vx_status VX_API_CALL vxGetStatus(vx_reference reference);
/* demos/feature_tracker/main_feature_tracker.cpp:171.  Given the context (or any other object) the performance
 * directives switch per-node timing for the whole context, as in the reference; given a vx_node or a vx_graph
 * they apply to that node / graph only (extension: each timed node costs two HIP events per execution). */
VX_API_ENTRY vx_status VX_API_CALL vxDirective(vx_reference reference, vx_enum directive);
/* main_feature_tracker.cpp:178 */
VX_API_ENTRY void VX_API_CALL vxRegisterLogCallback(vx_context context, vx_log_callback_f callback, vx_bool reentrant);
/* demos/video_stabilizer/homography_filter_node.cpp:213 */
VX_API_ENTRY void VX_API_CALL vxAddLogEntry(vx_reference ref, vx_status status, const char *message, ...);
VX_API_ENTRY vx_status VX_API_CALL vxQueryReference(vx_reference ref, vx_enum attribute, void *ptr, vx_size size);
VX_API_ENTRY vx_status VX_API_CALL vxRetainReference(vx_reference ref);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseReference(vx_reference *ref);

/* ---- image ------------------------------------------------------------------------------- */
/* feature_tracker.cpp:319, main_feature_tracker.cpp:233-234 (U8 and RGBX, pitch-linear in HBM) */
VX_API_ENTRY vx_image VX_API_CALL vxCreateImage(vx_context context, vx_uint32 width, vx_uint32 height, vx_df_image color);
/* demos/video_stabilizer/stabilizer.cpp:237: graph-internal image; width = height = 0 are inferred from the node that
 * writes it at vxVerifyGraph */
VX_API_ENTRY vx_image VX_API_CALL vxCreateVirtualImage(vx_graph graph, vx_uint32 width, vx_uint32 height, vx_df_image color);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseImage(vx_image *image);
/* feature_tracker.cpp:135-137,192-194 */
VX_API_ENTRY vx_status VX_API_CALL vxQueryImage(vx_image image, vx_enum attribute, void *ptr, vx_size size);
/* nvxio/src/OVX/Private/TypesOVX.cpp:58-64: host staging copy or the device pointer itself
 * (mem_type NVX_MEMORY_TYPE_HIP, see NVX/nvx.h) */
VX_API_ENTRY vx_status VX_API_CALL vxMapImagePatch(vx_image image, const vx_rectangle_t *rect, vx_uint32 plane_index,
                                                   vx_map_id *map_id, vx_imagepatch_addressing_t *addr, void **ptr,
                                                   vx_enum usage, vx_enum mem_type, vx_uint32 flags);
VX_API_ENTRY vx_status VX_API_CALL vxUnmapImagePatch(vx_image image, vx_map_id map_id);
VX_API_ENTRY vx_status VX_API_CALL vxCopyImagePatch(vx_image image, const vx_rectangle_t *rect, vx_uint32 plane_index,
                                                    const vx_imagepatch_addressing_t *user_addr, void *user_ptr,
                                                    vx_enum usage, vx_enum user_mem_type);

/* stabilizer.cpp:404: an image every pixel of which holds `value` (U8 and RGBX); ordinary memory here, so it may also
 * be written */
VX_API_ENTRY vx_image VX_API_CALL vxCreateUniformImage(vx_context context, vx_uint32 width, vx_uint32 height, vx_df_image color,
                                                       const vx_pixel_value_t *value);
/* demos/video_stabilizer/main_video_stabilizer.cpp:233-235: a view of `rect` sharing the parent's memory (the parent
 * stays alive while the view does) */
VX_API_ENTRY vx_image VX_API_CALL vxCreateImageFromROI(vx_image img, const vx_rectangle_t *rect);
/* main_video_stabilizer.cpp:225: the whole image (every kernel here defines all of its output pixels) */
VX_API_ENTRY vx_status VX_API_CALL vxGetValidRegionImage(vx_image image, vx_rectangle_t *rect);

/* ---- pyramid ------------------------------------------------------------------------------ */
/* feature_tracker.cpp:287 (level i is ceil(base * scale^i), feature_tracker_nvxcu.cpp:164-176) */
VX_API_ENTRY vx_pyramid VX_API_CALL vxCreatePyramid(vx_context context, vx_size levels, vx_float32 scale,
                                                    vx_uint32 width, vx_uint32 height, vx_df_image format);
VX_API_ENTRY vx_status VX_API_CALL vxReleasePyramid(vx_pyramid *pyr);
VX_API_ENTRY vx_status VX_API_CALL vxQueryPyramid(vx_pyramid pyr, vx_enum attribute, void *ptr, vx_size size);
/* feature_tracker.cpp:362 -- returns an OWNED reference (released at :448) aliasing the level's memory */
VX_API_ENTRY vx_image VX_API_CALL vxGetPyramidLevel(vx_pyramid pyr, vx_uint32 index);

/* ---- matrix ------------------------------------------------------------------------------- */
enum vx_matrix_attribute_e {
    VX_MATRIX_TYPE = 0x80B00, VX_MATRIX_ROWS, VX_MATRIX_COLUMNS, VX_MATRIX_SIZE,
    VX_MATRIX_ATTRIBUTE_TYPE = VX_MATRIX_TYPE, VX_MATRIX_ATTRIBUTE_ROWS = VX_MATRIX_ROWS,
    VX_MATRIX_ATTRIBUTE_COLUMNS = VX_MATRIX_COLUMNS, VX_MATRIX_ATTRIBUTE_SIZE = VX_MATRIX_SIZE
};
/* demos/video_stabilizer/stabilizer.cpp:262,289; homography_filter_node.cpp:50-51 */
VX_API_ENTRY vx_matrix VX_API_CALL vxCreateMatrix(vx_context context, vx_enum data_type, vx_size columns, vx_size rows);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseMatrix(vx_matrix *mat);
VX_API_ENTRY vx_status VX_API_CALL vxQueryMatrix(vx_matrix mat, vx_enum attribute, void *ptr, vx_size size);
VX_API_ENTRY vx_status VX_API_CALL vxCopyMatrix(vx_matrix mat, void *user_ptr, vx_enum usage, vx_enum user_mem_type);
VX_API_ENTRY vx_status VX_API_CALL vxReadMatrix(vx_matrix mat, void *array);          /* OpenVX 1.0 */
VX_API_ENTRY vx_status VX_API_CALL vxWriteMatrix(vx_matrix mat, const void *array);   /* OpenVX 1.0 */

/* ---- array -------------------------------------------------------------------------------- */
/* feature_tracker.cpp:298,308 */
VX_API_ENTRY vx_array VX_API_CALL vxCreateArray(vx_context context, vx_enum item_type, vx_size capacity);
/* demos/video_stabilizer/stabilizer.cpp:254,263: graph-internal array (same object kind; access from the host is
 * not restricted here) */
VX_API_ENTRY vx_array VX_API_CALL vxCreateVirtualArray(vx_graph graph, vx_enum item_type, vx_size capacity);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseArray(vx_array *arr);
/* feature_tracker.cpp:240, nvxio/src/OVX/Private/TypesOVX.cpp:88-97 */
VX_API_ENTRY vx_status VX_API_CALL vxQueryArray(vx_array arr, vx_enum attribute, void *ptr, vx_size size);
VX_API_ENTRY vx_status VX_API_CALL vxAddArrayItems(vx_array arr, vx_size count, const void *ptr, vx_size stride);
VX_API_ENTRY vx_status VX_API_CALL vxTruncateArray(vx_array arr, vx_size new_num_items);
VX_API_ENTRY vx_status VX_API_CALL vxMapArrayRange(vx_array arr, vx_size range_start, vx_size range_end, vx_map_id *map_id,
                                                   vx_size *stride, void **ptr, vx_enum usage, vx_enum mem_type, vx_uint32 flags);
VX_API_ENTRY vx_status VX_API_CALL vxUnmapArrayRange(vx_array arr, vx_map_id map_id);
VX_API_ENTRY vx_status VX_API_CALL vxCopyArrayRange(vx_array arr, vx_size range_start, vx_size range_end, vx_size user_stride,
                                                    void *user_ptr, vx_enum usage, vx_enum user_mem_type);
/* OpenVX 1.0 accessors still used by the SFM sample (sfm/SfM.cpp) */
VX_API_ENTRY vx_status VX_API_CALL vxAccessArrayRange(vx_array arr, vx_size start, vx_size end, vx_size *stride, void **ptr, vx_enum usage);
VX_API_ENTRY vx_status VX_API_CALL vxCommitArrayRange(vx_array arr, vx_size start, vx_size end, const void *ptr);
#define vxFormatArrayPointer(ptr, index, stride) (&(((vx_uint8 *)(ptr))[(index) * (stride)]))
#define vxArrayItem(type, ptr, index, stride) (*(type *)(vxFormatArrayPointer((ptr), (index), (stride))))

/* ---- scalar ------------------------------------------------------------------------------- */
/* feature_tracker.cpp:372-380 */
VX_API_ENTRY vx_scalar VX_API_CALL vxCreateScalar(vx_context context, vx_enum data_type, const void *ptr);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseScalar(vx_scalar *scalar);
VX_API_ENTRY vx_status VX_API_CALL vxQueryScalar(vx_scalar scalar, vx_enum attribute, void *ptr, vx_size size);
VX_API_ENTRY vx_status VX_API_CALL vxCopyScalar(vx_scalar scalar, void *user_ptr, vx_enum usage, vx_enum user_mem_type);
VX_API_ENTRY vx_status VX_API_CALL vxReadScalarValue(vx_scalar scalar, void *ptr);
VX_API_ENTRY vx_status VX_API_CALL vxWriteScalarValue(vx_scalar scalar, const void *ptr);

/* ---- delay -------------------------------------------------------------------------------- */
/* feature_tracker.cpp:289,300; main_feature_tracker.cpp:236 (exemplar is cloned, then released by the caller) */
VX_API_ENTRY vx_delay VX_API_CALL vxCreateDelay(vx_context context, vx_reference exemplar, vx_size num_slots);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseDelay(vx_delay *delay);
VX_API_ENTRY vx_status VX_API_CALL vxQueryDelay(vx_delay delay, vx_enum attribute, void *ptr, vx_size size);
/* feature_tracker.cpp:229,323,360,403-406,419 -- index 0, -1, ..., 1-slots; BORROWED reference */
VX_API_ENTRY vx_reference VX_API_CALL vxGetReferenceFromDelay(vx_delay delay, vx_int32 index);
/* feature_tracker.cpp:220-221, main_feature_tracker.cpp:298,392 */
VX_API_ENTRY vx_status VX_API_CALL vxAgeDelay(vx_delay delay);

/* ---- graph -------------------------------------------------------------------------------- */
/* feature_tracker.cpp:352 */
VX_API_ENTRY vx_graph VX_API_CALL vxCreateGraph(vx_context context);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseGraph(vx_graph *graph);
/* feature_tracker.cpp:443: validates parameters and orders nodes by their data dependencies */
VX_API_ENTRY vx_status VX_API_CALL vxVerifyGraph(vx_graph graph);
/* feature_tracker.cpp:224: synchronous */
VX_API_ENTRY vx_status VX_API_CALL vxProcessGraph(vx_graph graph);
/* asynchronous pair (OpenVX 1.1): every graph owns a HIP stream, so independent graphs overlap */
VX_API_ENTRY vx_status VX_API_CALL vxScheduleGraph(vx_graph graph);
VX_API_ENTRY vx_status VX_API_CALL vxWaitGraph(vx_graph graph);
/* nvxio/src/OVX/UtilityOVX.cpp:46-56 */
VX_API_ENTRY vx_status VX_API_CALL vxQueryGraph(vx_graph graph, vx_enum attribute, void *ptr, vx_size size);
/* feature_tracker.cpp:435-436 (NVX_GRAPH_VERIFY_OPTIONS, see NVX/nvx.h) */
VX_API_ENTRY vx_status VX_API_CALL vxSetGraphAttribute(vx_graph graph, vx_enum attribute, const void *ptr, vx_size size);
VX_API_ENTRY vx_bool VX_API_CALL vxIsGraphVerified(vx_graph graph);

/* ---- node --------------------------------------------------------------------------------- */
VX_API_ENTRY vx_status VX_API_CALL vxReleaseNode(vx_node *node);
/* nvxio/src/OVX/UtilityOVX.cpp:58-68 */
VX_API_ENTRY vx_status VX_API_CALL vxQueryNode(vx_node node, vx_enum attribute, void *ptr, vx_size size);
VX_API_ENTRY vx_status VX_API_CALL vxSetNodeAttribute(vx_node node, vx_enum attribute, const void *ptr, vx_size size);
/* feature_tracker.cpp:216-217: rebinding without re-verification; optional parameters accept NULL */
VX_API_ENTRY vx_status VX_API_CALL vxSetParameterByIndex(vx_node node, vx_uint32 index, vx_reference value);

/* feature_tracker.cpp:386 -- RGBX -> U8 (BT.709 luma) or U8 -> U8 */
VX_API_ENTRY vx_node VX_API_CALL vxColorConvertNode(vx_graph graph, vx_image input, vx_image output);
/* feature_tracker.cpp:393 -- input may BE the pyramid's level 0 (no copy is made then) */
VX_API_ENTRY vx_node VX_API_CALL vxGaussianPyramidNode(vx_graph graph, vx_image input, vx_pyramid gaussian);
/* feature_tracker.cpp:402-412 */
VX_API_ENTRY vx_node VX_API_CALL vxOpticalFlowPyrLKNode(vx_graph graph, vx_pyramid old_images, vx_pyramid new_images,
                                                        vx_array old_points, vx_array new_points_estimates,
                                                        vx_array new_points, vx_enum termination, vx_scalar epsilon,
                                                        vx_scalar num_iterations, vx_scalar use_initial_estimate,
                                                        vx_size window_dimension);

/* demos/video_stabilizer/stabilizer.cpp:292-295.  output(x,y) = input(x0/z0, y0/z0) with
 *   x0 = m[0][0] x + m[1][0] y + m[2][0],  y0 = m[0][1] x + m[1][1] y + m[2][1],  z0 = m[0][2] x + m[1][2] y + m[2][2]
 * for the 3x3 FLOAT32 matrix as stored (OpenVX: the matrix is kept column by column -- the layout
 * demos/video_stabilizer/truncate_transform_node.cpp:163-165 writes).  U8 and RGBX; NEAREST_NEIGHBOR and BILINEAR;
 * samples outside the input read the node's constant border value (VX_BORDER_UNDEFINED = constant 0). */
VX_API_ENTRY vx_node VX_API_CALL vxWarpPerspectiveNode(vx_graph graph, vx_image input, vx_matrix matrix, vx_enum type, vx_image output);

/* ---- user kernels (demos/video_stabilizer/{homography_filter,smoother,truncate_transform}_node.cpp) ------------- */
/* Host callbacks executed inside the graph: the runtime drains the graph's HIP stream before calling `func`, so every
 * object the earlier nodes wrote is readable through the ordinary accessors, and what `func` writes is seen by the
 * nodes after it. */
typedef vx_status (VX_CALLBACK *vx_kernel_f)(vx_node node, const vx_reference *parameters, vx_uint32 num);
typedef vx_status (VX_CALLBACK *vx_kernel_initialize_f)(vx_node node, const vx_reference *parameters, vx_uint32 num);
typedef vx_status (VX_CALLBACK *vx_kernel_deinitialize_f)(vx_node node, const vx_reference *parameters, vx_uint32 num);
typedef vx_status (VX_CALLBACK *vx_kernel_validate_f)(vx_node node, const vx_reference parameters[], vx_uint32 num,
                                                      vx_meta_format metas[]);
enum vx_kernel_attribute_e {
    VX_KERNEL_PARAMETERS = 0x80400, VX_KERNEL_NAME, VX_KERNEL_ENUM, VX_KERNEL_LOCAL_DATA_SIZE,
    VX_KERNEL_ATTRIBUTE_PARAMETERS = VX_KERNEL_PARAMETERS, VX_KERNEL_ATTRIBUTE_NAME = VX_KERNEL_NAME,
    VX_KERNEL_ATTRIBUTE_ENUM = VX_KERNEL_ENUM
};
#define VX_KERNEL_BASE(vendor, lib) (((vendor) << 20) | ((lib) << 12))
/* homography_filter_node.cpp:210 */
VX_API_ENTRY vx_status VX_API_CALL vxAllocateUserKernelId(vx_context context, vx_enum *pKernelEnumId);
/* homography_filter_node.cpp:218-225 */
VX_API_ENTRY vx_kernel VX_API_CALL vxAddUserKernel(vx_context context, const vx_char name[VX_MAX_KERNEL_NAME], vx_enum enumeration,
                                                   vx_kernel_f func_ptr, vx_uint32 numParams, vx_kernel_validate_f validate,
                                                   vx_kernel_initialize_f init, vx_kernel_deinitialize_f deinit);
/* homography_filter_node.cpp:234-237 */
VX_API_ENTRY vx_status VX_API_CALL vxAddParameterToKernel(vx_kernel kernel, vx_uint32 index, vx_enum dir, vx_enum data_type, vx_enum state);
VX_API_ENTRY vx_status VX_API_CALL vxFinalizeKernel(vx_kernel kernel);
VX_API_ENTRY vx_status VX_API_CALL vxRemoveKernel(vx_kernel kernel);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseKernel(vx_kernel *kernel);
VX_API_ENTRY vx_status VX_API_CALL vxQueryKernel(vx_kernel kernel, vx_enum attribute, void *ptr, vx_size size);
/* homography_filter_node.cpp:263 -- finalized kernels only */
VX_API_ENTRY vx_kernel VX_API_CALL vxGetKernelByName(vx_context context, const vx_char *name);
VX_API_ENTRY vx_kernel VX_API_CALL vxGetKernelByEnum(vx_context context, vx_enum kernel);
/* homography_filter_node.cpp:267: a node with every parameter unset; vxSetParameterByIndex fills them in */
VX_API_ENTRY vx_node VX_API_CALL vxCreateGenericNode(vx_graph graph, vx_kernel kernel);
/* homography_filter_node.cpp:197-199: the validator describes each output; vxVerifyGraph compares it with the object bound */
VX_API_ENTRY vx_status VX_API_CALL vxSetMetaFormatAttribute(vx_meta_format meta, vx_enum attribute, const void *ptr, vx_size size);
VX_API_ENTRY vx_status VX_API_CALL vxSetMetaFormatFromReference(vx_meta_format meta, vx_reference exemplar);

/* OpenVX 1.0 form of the same, as the SFM sample's scaleCloud kernel uses it (sfm/SfM.cpp:1204-1400): one validator call
 * per input parameter and per output parameter; validators reach the bound object through a vx_parameter. */
typedef vx_status (VX_CALLBACK *vx_kernel_input_validate_f)(vx_node node, vx_uint32 index);
typedef vx_status (VX_CALLBACK *vx_kernel_output_validate_f)(vx_node node, vx_uint32 index, vx_meta_format meta);
enum vx_parameter_attribute_e {
    VX_PARAMETER_INDEX = 0x80500, VX_PARAMETER_DIRECTION, VX_PARAMETER_TYPE, VX_PARAMETER_STATE, VX_PARAMETER_REF,
    VX_PARAMETER_ATTRIBUTE_INDEX = VX_PARAMETER_INDEX, VX_PARAMETER_ATTRIBUTE_DIRECTION = VX_PARAMETER_DIRECTION,
    VX_PARAMETER_ATTRIBUTE_TYPE = VX_PARAMETER_TYPE, VX_PARAMETER_ATTRIBUTE_STATE = VX_PARAMETER_STATE,
    VX_PARAMETER_ATTRIBUTE_REF = VX_PARAMETER_REF
};
VX_API_ENTRY vx_kernel VX_API_CALL vxAddKernel(vx_context context, const vx_char name[VX_MAX_KERNEL_NAME], vx_enum enumeration,
                                               vx_kernel_f func_ptr, vx_uint32 numParams, vx_kernel_input_validate_f input,
                                               vx_kernel_output_validate_f output, vx_kernel_initialize_f init,
                                               vx_kernel_deinitialize_f deinit);
/* SfM.cpp:1211-1235: a +1 reference to the parameter; VX_PARAMETER_REF hands out a +1 reference to the bound object
 * (NULL when nothing is bound), which the caller releases */
VX_API_ENTRY vx_parameter VX_API_CALL vxGetParameterByIndex(vx_node node, vx_uint32 index);
VX_API_ENTRY vx_status VX_API_CALL vxQueryParameter(vx_parameter param, vx_enum attribute, void *ptr, vx_size size);
VX_API_ENTRY vx_status VX_API_CALL vxReleaseParameter(vx_parameter *param);

#ifdef __cplusplus
}
#endif
#endif /* VX_VX_H */
