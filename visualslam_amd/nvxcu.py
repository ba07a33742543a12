"""ctypes binding of the thin C ABI (include/NVX/nvxcu.h) exported by libvisualslam_amd.so.

This is plumbing for the tests and the benchmark: structs mirror the header field for field and
every call goes straight through the C ABI -- there is no Python or CPU fallback.  Loading fails
loudly when the HIP library has not been built.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VISUALSLAM_AMD_LIB: development override (instrumented builds of the same library)
LIB_PATH = os.environ.get("VISUALSLAM_AMD_LIB") or os.path.join(_HERE, "lib", "libvisualslam_amd.so")


def _fourcc(s):
    a, b, c, d = (ord(ch) for ch in s)
    return a | (b << 8) | (c << 16) | (d << 24)


NVXCU_SUCCESS = 0
NVXCU_FAILURE = -1
NVXCU_ERROR_NO_MEMORY = -8
NVXCU_HARRIS_TRACK_NOALLOC_MAX_CELLS = 65536
NVXCU_ERROR_NOT_SUPPORTED = -3
NVXCU_ERROR_INVALID_PARAMETERS = -10
NVXCU_ERROR_INVALID_FORMAT = -14
NVXCU_ERROR_INVALID_DIMENSION = -15
NVXCU_ERROR_INVALID_TYPE = -17

NVXCU_DF_IMAGE_U8 = _fourcc("U008")
NVXCU_DF_IMAGE_RGBX = _fourcc("RGBA")
NVXCU_DF_IMAGE_RGB = _fourcc("RGB2")

NVXCU_TYPE_KEYPOINT = 0x021
NVXCU_TYPE_POINT2F = 0x400
NVXCU_TYPE_KEYPOINTF = 0x403

NVXCU_PITCH_LINEAR_IMAGE = 0
NVXCU_PITCH_LINEAR_PYRAMID = 0
NVXCU_PLAIN_ARRAY = 0
NVXCU_STREAM_EXEC_TARGET = 0
NVXCU_BORDER_MODE_UNDEFINED, NVXCU_BORDER_MODE_CONSTANT, NVXCU_BORDER_MODE_REPLICATE = 0, 1, 2
NVXCU_COLOR_SPACE_DEFAULT = 3
NVXCU_CHANNEL_RANGE_FULL = 0
NVXCU_TERM_CRITERIA_ITERATIONS, NVXCU_TERM_CRITERIA_EPSILON, NVXCU_TERM_CRITERIA_BOTH = 0, 1, 2
NVXCU_SCALE_PYRAMID_HALF = 0.5

ITEM_SIZE = {NVXCU_TYPE_POINT2F: 8, NVXCU_TYPE_KEYPOINTF: 28, NVXCU_TYPE_KEYPOINT: 28}


class nvxcu_image_t(C.Structure):
    _fields_ = [("image_type", C.c_int), ("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


class nvxcu_image_plane_t(C.Structure):
    _fields_ = [("dev_ptr", C.c_void_p), ("pitch_in_bytes", C.c_uint32)]


class nvxcu_pitch_linear_image_t(C.Structure):
    _fields_ = [("base", nvxcu_image_t), ("planes", nvxcu_image_plane_t * 4)]


class nvxcu_pyramid_t(C.Structure):
    _fields_ = [("pyramid_type", C.c_int), ("num_levels", C.c_uint32), ("scale", C.c_float)]


class nvxcu_pitch_linear_pyramid_t(C.Structure):
    _fields_ = [("base", nvxcu_pyramid_t), ("levels", C.POINTER(nvxcu_pitch_linear_image_t))]


class nvxcu_array_t(C.Structure):
    _fields_ = [("array_type", C.c_int), ("item_type", C.c_int), ("capacity", C.c_uint32)]


class nvxcu_plain_array_t(C.Structure):
    _fields_ = [("base", nvxcu_array_t), ("dev_ptr", C.c_void_p), ("num_items_dev_ptr", C.c_void_p)]


class nvxcu_border_t(C.Structure):
    _fields_ = [("mode", C.c_int), ("constant_value", C.c_uint32)]


class nvxcu_tmp_buf_size_t(C.Structure):
    _fields_ = [("dev_buf_size", C.c_size_t), ("host_buf_size", C.c_size_t)]


class nvxcu_tmp_buf_t(C.Structure):
    _fields_ = [("dev_ptr", C.c_void_p), ("host_ptr", C.c_void_p)]


_lib = None


def lib():
    """Load libvisualslam_amd.so (built by __graft_entry__.build() / visualslam_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libvisualslam_amd.so is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        # PyTorch bundles its own copy of the HIP runtime: load it first, so that this process has ONE runtime
        # (the library then binds to the copy already loaded; two copies each see "no device" from the other's side)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        P = C.c_void_p
        u32, f32 = C.c_uint32, C.c_float
        L.nvxcuColorConvert.argtypes = [P, P, C.c_int, C.c_int, P]
        L.nvxcuColorConvert.restype = C.c_int
        L.nvxcuGaussianPyramid_GetBufSize.argtypes = [u32, u32, u32, f32, P, P]
        L.nvxcuGaussianPyramid_GetBufSize.restype = nvxcu_tmp_buf_size_t
        L.nvxcuGaussianPyramid.argtypes = [P, P, P, P]
        L.nvxcuGaussianPyramid.restype = C.c_int
        for name in ("nvxcuHarrisTrack_GetBufSize", "nvxcuFastTrack_GetBufSize"):
            fn = getattr(L, name)
            fn.argtypes = [u32, u32, C.c_uint32, C.c_int, u32, u32, P]
            fn.restype = nvxcu_tmp_buf_size_t
        L.nvxcuHarrisTrack.argtypes = [P, P, P, P, f32, f32, u32, P, P, P]
        L.nvxcuHarrisTrack.restype = C.c_int
        L.nvxcuHarrisTrack_NoAlloc.argtypes = [P, P, P, P, f32, f32, u32, P, P]
        L.nvxcuHarrisTrack_NoAlloc.restype = C.c_int
        L.nvxcuFastTrack.argtypes = [P, P, P, P, u32, u32, u32, P, P, P]
        L.nvxcuFastTrack.restype = C.c_int
        L.nvxcuOpticalFlowPyrLK_GetBufSize.argtypes = [u32, u32, u32, f32, C.c_int, u32, u32, u32, P, P]
        L.nvxcuOpticalFlowPyrLK_GetBufSize.restype = nvxcu_tmp_buf_size_t
        L.nvxcuOpticalFlowPyrLK.argtypes = [P, P, P, P, C.c_int, f32, u32, u32, u32, P, P, P]
        L.nvxcuOpticalFlowPyrLK.restype = C.c_int
        L.visualslam_amd_sizeof_stream_exec_target.restype = C.c_size_t
        L.visualslam_amd_offsetof_stream.restype = C.c_size_t
        L.visualslam_amd_version.restype = C.c_char_p
        _lib = L
    return _lib


class StreamExecTarget:
    """nvxcu_stream_exec_target_t {base, hipStream_t stream, hipDeviceProp_t dev_prop} as raw bytes.

    dev_prop is left zeroed: no primitive reads it (the *_GetBufSize functions accept NULL)."""

    def __init__(self, stream_handle=0):
        L = lib()
        self._buf = C.create_string_buffer(L.visualslam_amd_sizeof_stream_exec_target())
        C.c_int.from_buffer(self._buf, 0).value = NVXCU_STREAM_EXEC_TARGET
        C.c_void_p.from_buffer(self._buf, L.visualslam_amd_offsetof_stream()).value = stream_handle or None
        self.stream = stream_handle

    @property
    def ptr(self):
        return C.cast(self._buf, C.c_void_p)


def make_image(dev_ptr, pitch, width, height, fmt=NVXCU_DF_IMAGE_U8):
    img = nvxcu_pitch_linear_image_t()
    img.base.image_type = NVXCU_PITCH_LINEAR_IMAGE
    img.base.format = fmt
    img.base.width = width
    img.base.height = height
    img.planes[0].dev_ptr = dev_ptr
    img.planes[0].pitch_in_bytes = pitch
    return img


def make_pyramid(level_images, scale=NVXCU_SCALE_PYRAMID_HALF):
    """level_images: list of nvxcu_pitch_linear_image_t.  Returns (pyramid struct, keep-alive array)."""
    n = len(level_images)
    arr = (nvxcu_pitch_linear_image_t * n)(*level_images)
    pyr = nvxcu_pitch_linear_pyramid_t()
    pyr.base.pyramid_type = NVXCU_PITCH_LINEAR_PYRAMID
    pyr.base.num_levels = n
    pyr.base.scale = scale
    pyr.levels = C.cast(arr, C.POINTER(nvxcu_pitch_linear_image_t))
    pyr._keep = arr
    return pyr


def make_array(dev_ptr, num_items_dev_ptr, capacity, item_type=NVXCU_TYPE_POINT2F):
    a = nvxcu_plain_array_t()
    a.base.array_type = NVXCU_PLAIN_ARRAY
    a.base.item_type = item_type
    a.base.capacity = capacity
    a.dev_ptr = dev_ptr
    a.num_items_dev_ptr = num_items_dev_ptr
    return a


def _ref(x):
    return None if x is None else C.cast(C.byref(x), C.c_void_p)


def color_convert(src, dst, target):
    return lib().nvxcuColorConvert(_ref(src), _ref(dst), NVXCU_COLOR_SPACE_DEFAULT, NVXCU_CHANNEL_RANGE_FULL,
                                   target.ptr)


def gaussian_pyramid(pyr, target, border_mode=NVXCU_BORDER_MODE_UNDEFINED, tmp=None):
    border = nvxcu_border_t(border_mode, 0)
    tmp = tmp or nvxcu_tmp_buf_t(None, None)
    return lib().nvxcuGaussianPyramid(_ref(pyr), _ref(tmp), _ref(border), target.ptr)


def harris_track_buf_size(width, height, item_type, capacity, cell_size):
    return lib().nvxcuHarrisTrack_GetBufSize(width, height, NVXCU_DF_IMAGE_U8, item_type, capacity, cell_size, None)


def fast_track_buf_size(width, height, item_type, capacity, cell_size):
    return lib().nvxcuFastTrack_GetBufSize(width, height, NVXCU_DF_IMAGE_U8, item_type, capacity, cell_size, None)


def harris_track(image, out, mask, tracked, k, threshold, cell_size, num_corners_dev_ptr, tmp, target):
    return lib().nvxcuHarrisTrack(_ref(image), _ref(out), _ref(mask), _ref(tracked), k, threshold, cell_size,
                                  num_corners_dev_ptr, _ref(tmp), target.ptr)


def harris_track_noalloc(image, out, mask, tracked, k, threshold, cell_size, num_corners_dev_ptr, target):
    """nvxcuHarrisTrack_NoAlloc: the cell keys live in a device array inside the library (no tmp_buf)"""
    return lib().nvxcuHarrisTrack_NoAlloc(_ref(image), _ref(out), _ref(mask), _ref(tracked), k, threshold, cell_size,
                                          num_corners_dev_ptr, target.ptr)


def fast_track(image, out, mask, tracked, fast_type, threshold, cell_size, num_corners_dev_ptr, tmp, target):
    return lib().nvxcuFastTrack(_ref(image), _ref(out), _ref(mask), _ref(tracked), fast_type, threshold, cell_size,
                                num_corners_dev_ptr, _ref(tmp), target.ptr)


def optical_flow_pyr_lk(old_pyr, new_pyr, old_pts, new_pts, termination, epsilon, num_iterations, window,
                        target, calc_error=0, border_mode=NVXCU_BORDER_MODE_UNDEFINED, tmp=None):
    border = nvxcu_border_t(border_mode, 0)
    tmp = tmp or nvxcu_tmp_buf_t(None, None)
    return lib().nvxcuOpticalFlowPyrLK(_ref(old_pyr), _ref(new_pyr), _ref(old_pts), _ref(new_pts), termination,
                                       epsilon, num_iterations, window, calc_error, _ref(tmp), _ref(border),
                                       target.ptr)
