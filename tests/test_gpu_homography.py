"""GPU parity of nvxFindHomographyNode / nvxuFindHomography with the oracle (oracle/vxoracle_homography.c), through the
OpenVX C API as demos/video_stabilizer/stabilizer.cpp:262-273 and SfM.cpp:1563-1580 call it: matrix, inlier mask and
mask item count, bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O
from test_oracle_homography import H_TRUE, scene
from visualslam_amd import ovx

L = ovx.lib()


@pytest.fixture()
def ctx(gpu):
    c = L.vxCreateContext()
    ovx.check(L.vxGetStatus(c), "vxCreateContext")
    yield c
    h = C.c_void_p(c)
    L.vxReleaseContext(C.byref(h))


def _arrays(ctx, src, dst, cap=None):
    cap = cap or max(len(src), len(dst), 1)
    a = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, cap)
    b = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, cap)
    if len(src):
        ovx.add_items(a, src)
    if len(dst):
        ovx.add_items(b, dst)
    return a, b


def _matrix(ctx):
    m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    ovx.check(L.vxGetStatus(m), "vxCreateMatrix")
    return m


def _read_matrix(m):
    out = np.zeros((3, 3), np.float32)
    ovx.check(L.vxCopyMatrix(m, out.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST), "vxCopyMatrix")
    return out


@pytest.mark.parametrize("n,thr,seed,noise,outl", [(4096, 3.0, 0, 0.3, 0.3), (1000, 0.5, 1, 0.1, 0.45), (1000, 3.0, 2, 1.0, 0.6),
                                                    (37, 3.0, 3, 0.2, 0.2), (9000, 2.0, 4, 0.5, 0.35), (150001, 1.5, 5, 0.4, 0.4)])
@pytest.mark.parametrize("large_array_kernels", [False, True])
def test_immediate_matches_oracle(ctx, monkeypatch, n, thr, seed, noise, outl, large_array_kernels):
    # arrays of capacity <= 4096 take the on-chip kernel (points in registers + LDS, 96 registers); the switch forces the any-size kernel
    if large_array_kernels:
        monkeypatch.setenv("VISUALSLAM_AMD_HG_GENERIC", "1")
    src, dst, _ = scene(n=n, noise=noise, outliers=outl, seed=seed)
    a, b = _arrays(ctx, src, dst)
    m = _matrix(ctx)
    mask = L.vxCreateArray(ctx, ovx.VX_TYPE_UINT8, n)
    ovx.check(L.nvxuFindHomography(ctx, a, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, thr, 2000, 10, 0.995, 0.45, mask),
              "nvxuFindHomography")
    Hw, mw, cw = O.find_homography(src, dst, thr, 2000, 10, 0.995, 0.45)
    got = ovx.read_array(mask, np.uint8)
    assert len(got) == n                                             # one item per correspondence (SfM.cpp:1586-1596)
    assert int((got != 0).sum()) == cw
    assert got.tobytes() == mw.tobytes()
    assert _read_matrix(m).tobytes() == Hw.tobytes()                 # identical bits, transposed layout included
    assert cw > 4


@pytest.mark.parametrize("large_array_kernels", [False, True])
def test_refine_and_estimate_limits(ctx, monkeypatch, large_array_kernels):
    if large_array_kernels:
        monkeypatch.setenv("VISUALSLAM_AMD_HG_GENERIC", "1")
    src, dst, _ = scene(n=1500, noise=0.8, seed=7)
    a, b = _arrays(ctx, src, dst)
    for est, ref in ((2000, 0), (5, 10), (1, 1), (2000, 3)):
        m = _matrix(ctx)
        mask = L.vxCreateArray(ctx, ovx.VX_TYPE_UINT8, 1500)
        ovx.check(L.nvxuFindHomography(ctx, a, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, est, ref, 0.995, 0.45, mask))
        Hw, mw, _ = O.find_homography(src, dst, 3.0, est, ref, 0.995, 0.45)
        assert _read_matrix(m).tobytes() == Hw.tobytes() and ovx.read_array(mask, np.uint8).tobytes() == mw.tobytes(), (est, ref)


def test_degenerate_inputs_and_no_mask(ctx):
    src, dst, _ = scene(n=60, seed=2)
    for s, d in ((src[:3], dst[:3]), (src, np.full_like(dst, -1.0)), (np.tile(src[:1], (60, 1)), dst)):
        a, b = _arrays(ctx, s, d, cap=64)
        m = _matrix(ctx)
        mask = L.vxCreateArray(ctx, ovx.VX_TYPE_UINT8, 64)
        ovx.check(L.nvxuFindHomography(ctx, a, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, mask))
        np.testing.assert_array_equal(_read_matrix(m), np.eye(3, dtype=np.float32))
        got = ovx.read_array(mask, np.uint8)
        assert len(got) == min(len(s), len(d)) and not got.any()
    a, b = _arrays(ctx, src, dst)
    m = _matrix(ctx)
    ovx.check(L.nvxuFindHomography(ctx, a, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, None))
    assert _read_matrix(m).tobytes() == O.find_homography(src, dst)[0].tobytes()
    # errors: wrong method, wrong item types, wrong matrix shape
    assert L.nvxuFindHomography(ctx, a, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_LMEDS, 3.0, 2000, 10, 0.995, 0.45, None) != ovx.VX_SUCCESS
    kp = L.vxCreateArray(ctx, ovx.NVX_TYPE_KEYPOINTF, 10)
    assert L.nvxuFindHomography(ctx, kp, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, None) != ovx.VX_SUCCESS
    m2 = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 2, 3)
    assert L.nvxuFindHomography(ctx, a, b, m2, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, None) != ovx.VX_SUCCESS


def test_graph_node_after_pyrlk_like_the_stabilizer(ctx):
    """the stabilizer's sub-graph: PyrLK on delayed pyramids -> FindHomography on (old points, tracked points) with a
    virtual mask array (stabilizer.cpp:254-273); frames related by a known similarity"""
    from oracle.pipeline import OracleTracker
    from visualslam_amd import synth
    from visualslam_amd.thin_tracker import Params
    from visualslam_amd.vx_tracker import VxFeatureTracker
    w, h = 640, 480
    base = synth.base_frame(w, h, 77)
    frames = [synth.to_rgbx(synth.gray_frame(base, t, 77)) for t in range(4)]
    p = Params(pyr_levels=4, lk_win_size=10, array_capacity=1000)
    img = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    ovx.write_image(img, frames[0])
    trk = VxFeatureTracker(ctx, p, w, h, img, None)
    Hm = _matrix(ctx)
    mask = L.vxCreateVirtualArray(trk.g, ovx.VX_TYPE_UINT8, 1000)
    node = L.nvxFindHomographyNode(trk.g, L.vxGetReferenceFromDelay(trk.pts_delay, -1), trk.kp_curr, Hm,
                                   ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, mask)
    ovx.check(L.vxGetStatus(node), "nvxFindHomographyNode")
    ovx.check(L.vxVerifyGraph(trk.g), "vxVerifyGraph")
    ref = OracleTracker(p)
    ref.init(frames[0])
    for t in range(1, 4):
        ovx.write_image(img, frames[t])
        trk.track(img, None)
        old, cur = ref.track(frames[t])
        assert trk.curr().tobytes() == cur.tobytes()
        Hw, mw, cw = O.find_homography(old, cur, 3.0, 2000, 10, 0.995, 0.45)
        assert _read_matrix(Hm).tobytes() == Hw.tobytes(), f"frame {t}"
        assert ovx.read_array(mask, np.uint8).tobytes() == mw.tobytes(), f"frame {t}"
        assert cw > 0.8 * (cur[:, 0] >= 0).sum()                      # a pure translation: nearly every track is an inlier
        H = Hw.T
        assert abs(H[0, 0] - 1) < 0.01 and abs(H[1, 1] - 1) < 0.01 and abs(H[2, 0]) < 1e-4


def test_matrix_object(ctx):
    m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 2)
    assert ovx.query("vxQueryMatrix", m, ovx.VX_MATRIX_COLUMNS, C.c_size_t) == 3
    assert ovx.query("vxQueryMatrix", m, ovx.VX_MATRIX_ROWS, C.c_size_t) == 2
    assert ovx.query("vxQueryMatrix", m, ovx.VX_MATRIX_SIZE, C.c_size_t) == 24
    assert ovx.query("vxQueryMatrix", m, ovx.VX_MATRIX_TYPE, C.c_int32) == ovx.VX_TYPE_FLOAT32
    v = np.arange(6, dtype=np.float32)
    ovx.check(L.vxWriteMatrix(m, v.ctypes.data))
    out = np.zeros(6, np.float32)
    ovx.check(L.vxReadMatrix(m, out.ctypes.data))
    np.testing.assert_array_equal(out, v)
    h = C.c_void_p(m)
    ovx.check(L.vxReleaseMatrix(C.byref(h)))
    assert h.value is None
