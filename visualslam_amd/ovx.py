"""ctypes binding of the OpenVX-subset C API (include/VX/vx.h, VX/vxu.h, NVX/nvx.h) exported by
libvisualslam_amd.so -- plumbing for tests and bench.py; every call goes straight through the C ABI."""
import ctypes as C

import numpy as np

from . import nvxcu

VX_SUCCESS = 0
VX_FAILURE = -1
VX_ERROR_NOT_SUPPORTED = -3
VX_ERROR_INVALID_PARAMETERS = -10
VX_ERROR_INVALID_REFERENCE = -12
VX_ERROR_INVALID_FORMAT = -14
VX_ERROR_INVALID_DIMENSION = -15
VX_ERROR_INVALID_VALUE = -16
VX_ERROR_INVALID_TYPE = -17
VX_ERROR_INVALID_GRAPH = -18
NVX_ERROR_NO_GPU = -1000

VX_TYPE_UINT32, VX_TYPE_FLOAT32, VX_TYPE_SIZE, VX_TYPE_BOOL, VX_TYPE_ENUM = 0x007, 0x00A, 0x00D, 0x010, 0x00C
VX_TYPE_KEYPOINT = 0x021
VX_TYPE_UINT8, VX_TYPE_MATRIX = 0x003, 0x80B
VX_TYPE_IMAGE, VX_TYPE_ARRAY, VX_TYPE_PYRAMID, VX_TYPE_ERROR = 0x80F, 0x80E, 0x809, 0x811
NVX_TYPE_POINT2F, NVX_TYPE_POINT3F, NVX_TYPE_KEYPOINTF = 0x400, 0x401, 0x403
NVX_READ_ONLY_HIP, NVX_WRITE_ONLY_HIP, NVX_READ_AND_WRITE_HIP = 0x80400, 0x80401, 0x80402
VX_DF_IMAGE_U8, VX_DF_IMAGE_RGBX = nvxcu.NVXCU_DF_IMAGE_U8, nvxcu.NVXCU_DF_IMAGE_RGBX
VX_IMAGE_WIDTH, VX_IMAGE_HEIGHT, VX_IMAGE_FORMAT, VX_IMAGE_PLANES = 0x80F00, 0x80F01, 0x80F02, 0x80F03
VX_ARRAY_ITEMTYPE, VX_ARRAY_NUMITEMS, VX_ARRAY_CAPACITY, VX_ARRAY_ITEMSIZE = 0x80E00, 0x80E01, 0x80E02, 0x80E03
VX_PYRAMID_LEVELS = 0x80900
VX_DELAY_TYPE, VX_DELAY_SLOTS = 0x80600, 0x80601
VX_GRAPH_NUMNODES, VX_GRAPH_PERFORMANCE, VX_GRAPH_STATE = 0x80200, 0x80202, 0x80204
VX_NODE_PERFORMANCE = 0x80301
VX_REFERENCE_COUNT, VX_REFERENCE_TYPE = 0x80000, 0x80001
VX_DIRECTIVE_DISABLE_PERFORMANCE = 0x03002
VX_DIRECTIVE_ENABLE_PERFORMANCE = 0x03003
VX_READ_ONLY, VX_WRITE_ONLY, VX_READ_AND_WRITE = 0x11001, 0x11002, 0x11003
VX_MEMORY_TYPE_HOST, NVX_MEMORY_TYPE_HIP = 0x0E001, 0x0E100
VX_TERM_CRITERIA_ITERATIONS, VX_TERM_CRITERIA_EPSILON, VX_TERM_CRITERIA_BOTH = 0x17000, 0x17001, 0x17002
NVX_GRAPH_VERIFY_OPTIONS = 0x80280
VX_SCALE_PYRAMID_HALF = 0.5
K_COLOR_CONVERT, K_GAUSSIAN_PYRAMID, K_PYRLK, K_HARRIS_TRACK, K_FAST_TRACK, K_COPY_IMAGE, K_FIND_HOMOGRAPHY, \
    K_WARP_PERSPECTIVE, K_USER = range(9)
VX_TYPE_KERNEL, VX_TYPE_DELAY, VX_TYPE_SCALAR, VX_TYPE_NODE = 0x804, 0x806, 0x80D, 0x803
VX_INPUT, VX_OUTPUT, VX_BIDIRECTIONAL = 0x04000, 0x04001, 0x04002
VX_PARAMETER_STATE_REQUIRED, VX_PARAMETER_STATE_OPTIONAL = 0x06000, 0x06001
VX_INTERPOLATION_NEAREST_NEIGHBOR, VX_INTERPOLATION_BILINEAR, VX_INTERPOLATION_AREA = 0x14000, 0x14001, 0x14002
VX_BORDER_UNDEFINED, VX_BORDER_CONSTANT, VX_BORDER_REPLICATE = 0x0C000, 0x0C001, 0x0C002
VX_NODE_BORDER = 0x80302
VX_SCALAR_TYPE = 0x80D00
VX_KERNEL_PARAMETERS, VX_KERNEL_NAME, VX_KERNEL_ENUM = 0x80400, 0x80401, 0x80402
VX_ERROR_INVALID_NODE = -19
VX_MATRIX_TYPE, VX_MATRIX_ROWS, VX_MATRIX_COLUMNS, VX_MATRIX_SIZE = 0x80B00, 0x80B01, 0x80B02, 0x80B03
NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, NVX_FIND_HOMOGRAPHY_METHOD_LMEDS = 0x80300, 0x80301


class vx_perf_t(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("tmp", "beg", "end", "sum", "avg", "min", "num", "max")]


class vx_rectangle_t(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("start_x", "start_y", "end_x", "end_y")]


class vx_imagepatch_addressing_t(C.Structure):
    _fields_ = [("dim_x", C.c_uint32), ("dim_y", C.c_uint32), ("stride_x", C.c_int32), ("stride_y", C.c_int32),
                ("scale_x", C.c_uint32), ("scale_y", C.c_uint32), ("step_x", C.c_uint32), ("step_y", C.c_uint32)]


class vx_border_t(C.Structure):
    _fields_ = [("mode", C.c_int32), ("constant_value", C.c_uint8 * 4)]


class vx_pixel_value_t(C.Union):
    _fields_ = [("RGBX", C.c_uint8 * 4), ("U8", C.c_uint8), ("U32", C.c_uint32), ("reserved", C.c_uint8 * 16)]


LOG_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int32, C.c_char_p)
# user-kernel callbacks (VX/vx.h): func / init / deinit (node, parameters, num); validate (node, parameters, num, metas)
KERNEL_F = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32)
VALIDATE_F = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p))
INPUT_VALIDATE_F = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32)                  # OpenVX 1.0 (vxAddKernel)
OUTPUT_VALIDATE_F = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p)
VX_PARAMETER_INDEX, VX_PARAMETER_DIRECTION, VX_PARAMETER_TYPE, VX_PARAMETER_STATE, VX_PARAMETER_REF = range(0x80500, 0x80505)

_SIGS = {
    "vxCreateContext": (C.c_void_p, []),
    "vxReleaseContext": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxGetContext": (C.c_void_p, [C.c_void_p]),
    "vxGetStatus": (C.c_int32, [C.c_void_p]),
    "vxDirective": (C.c_int32, [C.c_void_p, C.c_int32]),
    "vxRegisterLogCallback": (None, [C.c_void_p, LOG_CALLBACK, C.c_int]),
    "vxQueryReference": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxRetainReference": (C.c_int32, [C.c_void_p]),
    "vxReleaseReference": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxCreateImage": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vxReleaseImage": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxQueryImage": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxMapImagePatch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t),
                                    C.POINTER(vx_imagepatch_addressing_t), C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_uint32]),
    "vxUnmapImagePatch": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "vxCopyImagePatch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(vx_imagepatch_addressing_t), C.c_void_p, C.c_int32, C.c_int32]),
    "vxCreatePyramid": (C.c_void_p, [C.c_void_p, C.c_size_t, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vxReleasePyramid": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxQueryPyramid": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxGetPyramidLevel": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "vxCreateArray": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_size_t]),
    "vxReleaseArray": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxQueryArray": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxAddArrayItems": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "vxTruncateArray": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "vxMapArrayRange": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_uint32]),
    "vxUnmapArrayRange": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "vxCopyArrayRange": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int32, C.c_int32]),
    "vxAccessArrayRange": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.c_int32]),
    "vxCommitArrayRange": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "vxCreateScalar": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_void_p]),
    "vxReleaseScalar": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxQueryScalar": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxReadScalarValue": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "vxWriteScalarValue": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "vxCopyScalar": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "vxCreateDelay": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "vxReleaseDelay": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxQueryDelay": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxGetReferenceFromDelay": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "vxAgeDelay": (C.c_int32, [C.c_void_p]),
    "vxCreateGraph": (C.c_void_p, [C.c_void_p]),
    "vxReleaseGraph": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxVerifyGraph": (C.c_int32, [C.c_void_p]),
    "vxProcessGraph": (C.c_int32, [C.c_void_p]),
    "vxScheduleGraph": (C.c_int32, [C.c_void_p]),
    "vxWaitGraph": (C.c_int32, [C.c_void_p]),
    "vxQueryGraph": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxSetGraphAttribute": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxIsGraphVerified": (C.c_int, [C.c_void_p]),
    "vxReleaseNode": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxQueryNode": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxSetParameterByIndex": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "vxColorConvertNode": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vxGaussianPyramidNode": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vxOpticalFlowPyrLKNode": (C.c_void_p, [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "nvxHarrisTrackNode": (C.c_void_p, [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_uint32, C.c_void_p]),
    "nvxFastTrackNode": (C.c_void_p, [C.c_void_p] * 5 + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nvxCopyImageNode": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vxuColorConvert": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vxuGaussianPyramid": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vxuOpticalFlowPyrLK": (C.c_int32, [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "nvxuHarrisTrack": (C.c_int32, [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_uint32, C.c_void_p]),
    "nvxuFastTrack": (C.c_int32, [C.c_void_p] * 5 + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nvxuCopyImage": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vxCreateMatrix": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_size_t, C.c_size_t]),
    "vxReleaseMatrix": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxQueryMatrix": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxCopyMatrix": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "vxReadMatrix": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "vxWriteMatrix": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "vxCreateVirtualArray": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_size_t]),
    "vxCreateVirtualImage": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "nvxFindHomographyNode": (C.c_void_p, [C.c_void_p] * 4 + [C.c_int32, C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]),
    "nvxuFindHomography": (C.c_int32, [C.c_void_p] * 4 + [C.c_int32, C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]),
    "visualslam_amd_graph_order": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "visualslam_amd_graph_fusions": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "vxSetNodeAttribute": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxCreateUniformImage": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "vxCreateImageFromROI": (C.c_void_p, [C.c_void_p, C.POINTER(vx_rectangle_t)]),
    "vxGetValidRegionImage": (C.c_int32, [C.c_void_p, C.POINTER(vx_rectangle_t)]),
    "vxWarpPerspectiveNode": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "vxuWarpPerspective": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "vxAllocateUserKernelId": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "vxAddUserKernel": (C.c_void_p, [C.c_void_p, C.c_char_p, C.c_int32, KERNEL_F, C.c_uint32, VALIDATE_F, KERNEL_F, KERNEL_F]),
    "vxAddKernel": (C.c_void_p, [C.c_void_p, C.c_char_p, C.c_int32, KERNEL_F, C.c_uint32, INPUT_VALIDATE_F, OUTPUT_VALIDATE_F, KERNEL_F, KERNEL_F]),
    "vxGetParameterByIndex": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "vxQueryParameter": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxReleaseParameter": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxAddParameterToKernel": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_int32]),
    "vxFinalizeKernel": (C.c_int32, [C.c_void_p]),
    "vxRemoveKernel": (C.c_int32, [C.c_void_p]),
    "vxReleaseKernel": (C.c_int32, [C.POINTER(C.c_void_p)]),
    "vxQueryKernel": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxGetKernelByName": (C.c_void_p, [C.c_void_p, C.c_char_p]),
    "vxGetKernelByEnum": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "vxCreateGenericNode": (C.c_void_p, [C.c_void_p, C.c_void_p]),
    "vxSetMetaFormatAttribute": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "vxSetMetaFormatFromReference": (C.c_int32, [C.c_void_p, C.c_void_p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = nvxcu.lib()
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status, what=""):
    if status != VX_SUCCESS:
        raise RuntimeError(f"{what} failed with vx_status {status}")


def release(kind, handle):
    """vxRelease<kind>(&handle); returns the status."""
    h = C.c_void_p(handle)
    return getattr(lib(), f"vxRelease{kind}")(C.byref(h))


def query(fn_name, obj, attr, ctype):
    v = ctype()
    check(getattr(lib(), fn_name)(obj, attr, C.byref(v), C.sizeof(v)), fn_name)
    return v.value if hasattr(v, "value") else v


def scalar(ctx, vx_type, value):
    ct = {VX_TYPE_FLOAT32: C.c_float, VX_TYPE_UINT32: C.c_uint32, VX_TYPE_BOOL: C.c_int, VX_TYPE_SIZE: C.c_size_t,
          VX_TYPE_ENUM: C.c_int32}[vx_type]
    v = ct(value)
    return lib().vxCreateScalar(ctx, vx_type, C.byref(v))


def write_image(img, arr):
    """host numpy array (h, w) u8 or (h, w, 4) u8 -> vx_image via vxCopyImagePatch"""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w = arr.shape[:2]
    bpp = 4 if arr.ndim == 3 else 1
    addr = vx_imagepatch_addressing_t(w, h, bpp, w * bpp, 1024, 1024, 1, 1)
    check(lib().vxCopyImagePatch(img, None, 0, C.byref(addr), arr.ctypes.data, VX_WRITE_ONLY, VX_MEMORY_TYPE_HOST), "vxCopyImagePatch(write)")


def read_image(img):
    w = query("vxQueryImage", img, VX_IMAGE_WIDTH, C.c_uint32)
    h = query("vxQueryImage", img, VX_IMAGE_HEIGHT, C.c_uint32)
    fmt = query("vxQueryImage", img, VX_IMAGE_FORMAT, C.c_uint32)
    bpp = 4 if fmt == VX_DF_IMAGE_RGBX else 1
    out = np.empty((h, w, 4) if bpp == 4 else (h, w), np.uint8)
    addr = vx_imagepatch_addressing_t(w, h, bpp, w * bpp, 1024, 1024, 1, 1)
    check(lib().vxCopyImagePatch(img, None, 0, C.byref(addr), out.ctypes.data, VX_READ_ONLY, VX_MEMORY_TYPE_HOST), "vxCopyImagePatch(read)")
    return out


def num_items(arr):
    return query("vxQueryArray", arr, VX_ARRAY_NUMITEMS, C.c_size_t)


def read_array(arr, dtype=None):
    n = num_items(arr)
    isz = query("vxQueryArray", arr, VX_ARRAY_ITEMSIZE, C.c_size_t)
    raw = np.empty(n * isz, np.uint8)
    if n:
        check(lib().vxCopyArrayRange(arr, 0, n, isz, raw.ctypes.data, VX_READ_ONLY, VX_MEMORY_TYPE_HOST), "vxCopyArrayRange")
    if dtype is not None:
        return raw.view(dtype).copy()
    return raw.view(np.float32).reshape(n, 2).copy() if isz == 8 else raw


def add_items(arr, items):
    items = np.ascontiguousarray(items)
    n = len(items)
    isz = items.dtype.itemsize if items.ndim == 1 else items.dtype.itemsize * items.shape[1]
    check(lib().vxAddArrayItems(arr, n, items.ctypes.data, isz), "vxAddArrayItems")


def graph_fusions(graph):
    """what vxVerifyGraph fused (a debug export, not reference API): ColorConvert nodes converted inside the pyramid's first
    launch, detector cell passes riding in the PyrLK launch, virtual grey images made views of a pyramid's level 0,
    FindHomography nodes sharing a launch with a detector node's merge pass"""
    buf = (C.c_int * 4)()
    n = lib().visualslam_amd_graph_fusions(graph, buf, 4)
    assert n == 4, "graph not verified"
    return {"color_convert_in_pyramid": buf[0], "cells_in_pyrlk": buf[1], "virtual_gray_is_level0": buf[2],
            "merge_in_homography": buf[3]}
