/*
 * NVX/nvx.h -- the vendor-extension part of the API the VisionWorks samples call for this path:
 * float-coordinate point types, the Harris/FAST "track" detectors, device-pointer mapping and
 * version info.  Written from scratch for MI355X (see VX/vx.h for the conventions); citations are
 * relative to /root/reference/VisionWorks/VisionWorks-1.6-Samples/.
 */
#ifndef NVX_NVX_H
#define NVX_NVX_H

#include <VX/vx.h>
#include <VX/vxu.h>   /* the samples reach the vxu* calls through this header: samples/sfm/SfM.cpp:961-962 calls vxuColorConvert and
                       * vxuGaussianPyramid with <NVX/nvx.h> (via its nvxio headers) as its only OpenVX include */

#ifdef __cplusplus
extern "C" {
#endif

/* item types: numeric values shared with nvxcu_array_item_type_e (nvxio/src/OVX/Private/TypesOVX.cpp:97) */
enum nvx_type_e {
    NVX_TYPE_POINT2F   = VX_TYPE_VENDOR_STRUCT_START + 0,
    NVX_TYPE_POINT3F   = VX_TYPE_VENDOR_STRUCT_START + 1,
    NVX_TYPE_POINT4F   = VX_TYPE_VENDOR_STRUCT_START + 2,
    NVX_TYPE_KEYPOINTF = VX_TYPE_VENDOR_STRUCT_START + 3
};

/* layouts: nvxio/src/OpenGLShaders.hpp:567-584 */
typedef struct _nvx_point2f_t { vx_float32 x, y; } nvx_point2f_t;
typedef struct _nvx_point3f_t { vx_float32 x, y, z; } nvx_point3f_t;
typedef struct _nvx_point4f_t { vx_float32 x, y, z, w; } nvx_point4f_t;
typedef struct _nvx_keypointf_t {
    vx_float32 x, y, strength, scale, orientation;
    vx_int32 tracking_status;        /* 255 = tracked, 0 = lost, 1 = freshly detected */
    vx_float32 error;
} nvx_keypointf_t;

/* graph attribute (demos/feature_tracker/feature_tracker.cpp:435-436): a C string such as "-O3" */
enum nvx_graph_attribute_e { NVX_GRAPH_VERIFY_OPTIONS = 0x80280 };

/* device-pointer mapping for vxMapImagePatch / vxMapArrayRange (the reference's NVX_MEMORY_TYPE_CUDA,
 * nvxio/src/OVX/Private/TypesOVX.cpp:60,93): ptr is the HBM address, stride_y the device pitch */
enum nvx_memory_type_e { NVX_MEMORY_TYPE_HIP = 0x0E100 };

/* usages of the OpenVX 1.0 accessors (vxAccessArrayRange / vxCommitArrayRange) that hand out the HBM address instead of a host
 * copy (the reference's NVX_READ_ONLY_CUDA / NVX_WRITE_ONLY_CUDA, VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp:769,776) */
enum nvx_accessor_e { NVX_READ_ONLY_HIP = 0x80400, NVX_WRITE_ONLY_HIP, NVX_READ_AND_WRITE_HIP };

/* context creation failures reported through vxGetStatus((vx_reference)context)
 * (the reference's NVX_ERROR_NO_CUDA_GPU, nvxio/src/OVX/UtilityOVX.cpp:92-99) */
enum nvx_status_e {
    NVX_ERROR_NO_GPU = -1000, NVX_ERROR_UNSUPPORTED_GPU = -1001,
    /* the reference's spellings, so its callers (nvxio/src/OVX/UtilityOVX.cpp:92-99) compile unmodified */
    NVX_ERROR_NO_CUDA_GPU = NVX_ERROR_NO_GPU, NVX_ERROR_UNSUPPORTED_CUDA_GPU = NVX_ERROR_UNSUPPORTED_GPU
};

/* nvxio/src/OVX/UtilityOVX.cpp:72-88 */
typedef struct _nvx_module_version_t { vx_uint32 major, minor, patch; vx_char suffix[16]; } nvx_module_version_t;
typedef struct _nvx_version_info_t {
    nvx_module_version_t visionworks_version;
    vx_uint32 openvx_major_version, openvx_minor_version, openvx_patch_version;
} nvx_version_info_t;
VX_API_ENTRY void VX_API_CALL nvxGetVersionInfo(nvx_version_info_t *info);

/* feature_tracker.cpp:418-422, demos/video_stabilizer/stabilizer.cpp:299-301.
 * mask, tracked_points and num_corners (VX_TYPE_SIZE scalar) are optional.
 * parameter indices: 0 input, 1 output, 2 mask, 3 tracked_points, 4 k, 5 threshold, 6 cell_size, 7 num_corners */
VX_API_ENTRY vx_node VX_API_CALL nvxHarrisTrackNode(vx_graph graph, vx_image input, vx_array output, vx_image mask,
                                                    vx_array tracked_points, vx_float32 k, vx_float32 threshold,
                                                    vx_uint32 cell_size, vx_scalar num_corners);
/* feature_tracker.cpp:426-430; indices as above with 4 type, 5 threshold */
VX_API_ENTRY vx_node VX_API_CALL nvxFastTrackNode(vx_graph graph, vx_image input, vx_array output, vx_image mask,
                                                  vx_array tracked_points, vx_uint32 type, vx_uint32 threshold,
                                                  vx_uint32 cell_size, vx_scalar num_corners);
/* feature_tracker.cpp:326-330 */
VX_API_ENTRY vx_status VX_API_CALL nvxuHarrisTrack(vx_context context, vx_image input, vx_array output, vx_image mask,
                                                   vx_array tracked_points, vx_float32 k, vx_float32 threshold,
                                                   vx_uint32 cell_size, vx_scalar num_corners);
/* feature_tracker.cpp:334-338 */
VX_API_ENTRY vx_status VX_API_CALL nvxuFastTrack(vx_context context, vx_image input, vx_array output, vx_image mask,
                                                 vx_array tracked_points, vx_uint32 type, vx_uint32 threshold,
                                                 vx_uint32 cell_size, vx_scalar num_corners);

/* demos/video_stabilizer/stabilizer.cpp:264-273, VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp:1576-1578.
 * Homography between two NVX_TYPE_POINT2F arrays (RANSAC); correspondences with a negative coordinate (PyrLK's lost
 * points) are ignored.  `homography`: 3x3 VX_TYPE_FLOAT32, stored so that the consumers' transpose gives x' ~ H x
 * (homography_filter_node.cpp:87-91).  `mask` (optional): VX_TYPE_UINT8 array, one item per correspondence, non-zero =
 * inlier.  Parameter indices: 0 src, 1 dst, 2 homography, 3 method, 4 threshold, 5 maxEstimateIters, 6 maxRefineIters,
 * 7 confidence, 8 outlierRatio, 9 mask. */
enum nvx_find_homography_method_e {
    NVX_FIND_HOMOGRAPHY_METHOD_RANSAC = 0x80300, NVX_FIND_HOMOGRAPHY_METHOD_LMEDS   /* only RANSAC is implemented */
};
VX_API_ENTRY vx_node VX_API_CALL nvxFindHomographyNode(vx_graph graph, vx_array srcPoints, vx_array dstPoints,
                                                       vx_matrix homography, vx_enum method, vx_float32 threshold,
                                                       vx_uint32 maxEstimateIters, vx_uint32 maxRefineIters,
                                                       vx_float32 confidence, vx_float32 outlierRatio, vx_array mask);
VX_API_ENTRY vx_status VX_API_CALL nvxuFindHomography(vx_context context, vx_array srcPoints, vx_array dstPoints,
                                                      vx_matrix homography, vx_enum method, vx_float32 threshold,
                                                      vx_uint32 maxEstimateIters, vx_uint32 maxRefineIters,
                                                      vx_float32 confidence, vx_float32 outlierRatio, vx_array mask);

/* demos/video_stabilizer/stabilizer.cpp:245 */
VX_API_ENTRY vx_node VX_API_CALL nvxCopyImageNode(vx_graph graph, vx_image src, vx_image dst);
VX_API_ENTRY vx_status VX_API_CALL nvxuCopyImage(vx_context context, vx_image src, vx_image dst);

#ifdef __cplusplus
}
#endif
#endif
