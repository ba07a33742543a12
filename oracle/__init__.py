"""ctypes front-end of the CPU oracle (oracle/vxoracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py -- never by the product package.  PARITY UNPINNED (see vxoracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

POINT2F = 0
KEYPOINTF = 1
TERM_ITERATIONS, TERM_EPSILON, TERM_BOTH = 0, 1, 2
SUM_WAVE64, SUM_F64SEQ = 0, 1

keypointf_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("strength", "<f4"), ("scale", "<f4"),
                            ("orientation", "<f4"), ("tracking_status", "<i4"), ("error", "<f4")])
assert keypointf_dtype.itemsize == 28


class _Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("pitch", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


def build(force=False):
    """Compile the oracle with its Makefile (gcc only)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, n))
                            for n in ("libvxoracle.so", "libvxoracle_omp.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"])


_libs = {}


def lib(omp=False):
    name = "libvxoracle_omp.so" if omp else "libvxoracle.so"
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u8p, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
        L.orc_pyramid_dims.argtypes = [u32, u32, u32, f32, C.POINTER(u32), C.POINTER(u32)]
        L.orc_color_convert_rgbx_u8.argtypes = [u8p, u32, u8p, u32, u32, u32]
        L.orc_gaussian_half.argtypes = [u8p, u32, u32, u32, u8p, u32, u32, u32]
        L.orc_harris_response.argtypes = [u8p, u32, u32, u32, f32, C.c_void_p]
        L.orc_fast_strength.argtypes = [u8p, u32, u32, u32, u32, C.c_void_p]
        L.orc_harris_track.argtypes = [u8p, u32, u32, u32, u8p, u32, C.c_int, C.c_void_p, u32,
                                       f32, f32, u32, C.c_void_p, u32, C.POINTER(C.c_size_t), C.c_void_p]
        L.orc_harris_track.restype = u32
        L.orc_fast_track.argtypes = [u8p, u32, u32, u32, u8p, u32, C.c_int, C.c_void_p, u32,
                                     u32, u32, u32, C.c_void_p, u32, C.POINTER(C.c_size_t), C.c_void_p]
        L.orc_fast_track.restype = u32
        L.orc_optical_flow_pyr_lk.argtypes = [C.POINTER(_Image), C.POINTER(_Image), u32, C.c_int,
                                              C.c_void_p, C.c_void_p, u32, C.c_int, f32, u32, u32, C.c_int, C.c_int]
        _libs[name] = L
    return _libs[name]


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def pyramid_dims(width, height, levels, scale=0.5):
    ws = (C.c_uint32 * levels)()
    hs = (C.c_uint32 * levels)()
    lib().orc_pyramid_dims(width, height, levels, scale, ws, hs)
    return [(int(ws[i]), int(hs[i])) for i in range(levels)]


def color_convert(rgbx, omp=False):
    rgbx = _u8(rgbx)
    h, w, c = rgbx.shape
    assert c == 4
    out = np.empty((h, w), np.uint8)
    lib(omp).orc_color_convert_rgbx_u8(rgbx.ctypes.data, w * 4, out.ctypes.data, w, w, h)
    return out


def gaussian_half(src, omp=False):
    src = _u8(src)
    h, w = src.shape
    (_, _), (dw, dh) = pyramid_dims(w, h, 2)
    out = np.empty((dh, dw), np.uint8)
    lib(omp).orc_gaussian_half(src.ctypes.data, w, w, h, out.ctypes.data, dw, dw, dh)
    return out


def gaussian_pyramid(gray, levels, omp=False):
    """Level sizes follow ceil(w * 0.5^i) of the *base* size (feature_tracker_nvxcu.cpp:164-176)."""
    gray = _u8(gray)
    h, w = gray.shape
    dims = pyramid_dims(w, h, levels)
    out = [gray]
    for i in range(1, levels):
        src = out[-1]
        dw, dh = dims[i]
        dst = np.empty((dh, dw), np.uint8)
        lib(omp).orc_gaussian_half(src.ctypes.data, src.shape[1], src.shape[1], src.shape[0],
                                   dst.ctypes.data, dw, dw, dh)
        out.append(dst)
    return out


def harris_response(img, k, omp=False):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.float32)
    lib(omp).orc_harris_response(img.ctypes.data, w, w, h, k, out.ctypes.data)
    return out


def fast_strength(img, type_n, omp=False):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.int32)
    lib(omp).orc_fast_strength(img.ctypes.data, w, w, h, type_n, out.ctypes.data)
    return out


def _pts_buf(item_type, n):
    if item_type == POINT2F:
        return np.zeros((n, 2), np.float32)
    return np.zeros(n, keypointf_dtype)


def _track(fn, img, mask, tracked, item_type, p0, p1, cell, capacity, omp):
    img = _u8(img)
    h, w = img.shape
    mptr, mpitch = None, 0
    if mask is not None:
        mask = _u8(mask)
        assert mask.shape == img.shape
        mptr, mpitch = mask.ctypes.data, w
    tptr, nt = None, 0
    if tracked is not None:
        tracked = np.ascontiguousarray(tracked)
        tptr, nt = tracked.ctypes.data, len(tracked)
    out = _pts_buf(item_type, capacity)
    strength = np.zeros(capacity, np.float32)
    total = C.c_size_t(0)
    n = fn(omp)(img.ctypes.data, w, w, h, mptr, mpitch, item_type, tptr, nt, p0, p1, cell,
                out.ctypes.data, capacity, C.byref(total), strength.ctypes.data)
    return out[:n].copy(), int(total.value), strength[:n].copy()


def harris_track(img, mask=None, tracked=None, k=0.04, threshold=100.0, cell=18, capacity=2000,
                 item_type=POINT2F, omp=False):
    return _track(lambda o: lib(o).orc_harris_track, img, mask, tracked, item_type,
                  C.c_float(k), C.c_float(threshold), cell, capacity, omp)


def fast_track(img, mask=None, tracked=None, type_n=9, threshold=25, cell=18, capacity=2000,
               item_type=POINT2F, omp=False):
    return _track(lambda o: lib(o).orc_fast_track, img, mask, tracked, item_type,
                  int(type_n), int(threshold), cell, capacity, omp)


def _levels(pyr):
    keep = [_u8(l) for l in pyr]
    arr = (_Image * len(keep))()
    for i, l in enumerate(keep):
        arr[i] = _Image(l.ctypes.data, l.shape[1], l.shape[1], l.shape[0])
    return keep, arr


def optical_flow_pyr_lk(prev_pyr, curr_pyr, prev_pts, estimates=None, termination=TERM_BOTH,
                        epsilon=0.01, num_iterations=5, window=10, item_type=POINT2F,
                        sum_mode=SUM_WAVE64, omp=False, calc_error=False):
    kp, ap = _levels(prev_pyr)
    kc, ac = _levels(curr_pyr)
    assert len(kp) == len(kc)
    prev_pts = np.ascontiguousarray(prev_pts)
    out = np.array(prev_pts if estimates is None else estimates, copy=True)
    out = np.ascontiguousarray(out)
    assert out.dtype == prev_pts.dtype and len(out) == len(prev_pts)
    lib(omp).orc_optical_flow_pyr_lk(ap, ac, len(kp), item_type, prev_pts.ctypes.data,
                                     out.ctypes.data, len(prev_pts), termination, epsilon,
                                     num_iterations, window, sum_mode, 1 if calc_error else 0)
    return out


def homography_hypotheses(confidence=0.995, outlier_ratio=0.45, max_estimate_iters=2000):
    f = lib().orc_homography_hypotheses
    f.restype, f.argtypes = C.c_uint32, [C.c_float, C.c_float, C.c_uint32]
    return int(f(confidence, outlier_ratio, max_estimate_iters))


def find_homography(src, dst, threshold=3.0, max_estimate_iters=2000, max_refine_iters=10, confidence=0.995,
                    outlier_ratio=0.45):
    """nvxFindHomographyNode (RANSAC) as the oracle defines it.  src/dst: (n,2) float32.  Returns
    (matrix data as the API hands it out = H transposed, 3x3 float32; mask uint8 (n,); inlier count)."""
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    dst = np.ascontiguousarray(dst, np.float32).reshape(-1, 2)
    n = min(len(src), len(dst))
    H = np.zeros(9, np.float32)
    mask = np.zeros(max(n, 1), np.uint8)
    f = lib().orc_find_homography
    f.restype = C.c_uint32
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                  C.c_void_p, C.c_void_p]
    cnt = f(src.ctypes.data, dst.ctypes.data, n, threshold, max_estimate_iters, max_refine_iters, confidence,
            outlier_ratio, H.ctypes.data, mask.ctypes.data)
    return H.reshape(3, 3), mask[:n], int(cnt)


def warp_perspective(img, matrix_data, out_shape=None, bilinear=True, border=None):
    """vxWarpPerspectiveNode as the oracle defines it.  img: (h,w) uint8 or (h,w,4) uint8; matrix_data: the 3x3
    float32 vx_matrix contents as the API stores them (element [c][r] = M[r][c]); returns the warped image."""
    img = np.ascontiguousarray(img, np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    oh, ow = out_shape if out_shape is not None else (h, w)
    out = np.zeros((oh, ow) + (() if ch == 1 else (ch,)), np.uint8)
    m = np.ascontiguousarray(np.asarray(matrix_data, np.float32).reshape(3, 3).T)     # row-major M
    b = np.zeros(4, np.uint8) if border is None else np.ascontiguousarray(border, np.uint8)
    f = lib().orc_warp_perspective
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                  C.c_void_p, C.c_int, C.c_void_p]
    f(img.ctypes.data, w * ch, w, h, ch, out.ctypes.data, ow * ch, ow, oh, m.ctypes.data, 1 if bilinear else 0, b.ctypes.data)
    return out
