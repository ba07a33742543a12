"""CPU tests of the OpenVX-subset runtime's host logic (object model, delays, verification order,
error behaviour) through the C ABI.  No GPU here: the context is created in the runtime's
host-logic-only mode (VISUALSLAM_AMD_ALLOW_NO_GPU=1), where objects exist but nothing can execute --
vxProcessGraph and every vxu/nvxu call fail with NVX_ERROR_NO_GPU (there is no CPU fallback).
Contract under test: SURVEY.md section 8b (ownership, errors, delay semantics)."""
import ctypes as C
import os

import pytest

from visualslam_amd import ovx

L = ovx.lib()


@pytest.fixture
def ctx(monkeypatch):
    monkeypatch.setenv("VISUALSLAM_AMD_ALLOW_NO_GPU", "1")
    c = L.vxCreateContext()
    assert L.vxGetStatus(c) == ovx.VX_SUCCESS
    yield c
    h = C.c_void_p(c)
    assert L.vxReleaseContext(C.byref(h)) == ovx.VX_SUCCESS and h.value is None


def test_context_without_gpu_reports_no_gpu(monkeypatch):
    """nvxio/src/OVX/UtilityOVX.cpp:88-102: a GPU-less context is an object whose status says so"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    monkeypatch.delenv("VISUALSLAM_AMD_ALLOW_NO_GPU", raising=False)
    c = L.vxCreateContext()
    assert c and L.vxGetStatus(c) == ovx.NVX_ERROR_NO_GPU
    img = L.vxCreateImage(c, 8, 8, ovx.VX_DF_IMAGE_U8)
    assert img and L.vxGetStatus(img) == ovx.NVX_ERROR_NO_GPU      # error object, never NULL
    h = C.c_void_p(c)
    assert L.vxReleaseContext(C.byref(h)) == ovx.VX_SUCCESS


def test_nothing_executes_without_gpu(ctx):
    img = L.vxCreateImage(ctx, 32, 32, ovx.VX_DF_IMAGE_RGBX)
    gray = L.vxCreateImage(ctx, 32, 32, ovx.VX_DF_IMAGE_U8)
    assert L.vxuColorConvert(ctx, img, gray) == ovx.NVX_ERROR_NO_GPU
    g = L.vxCreateGraph(ctx)
    n = L.vxColorConvertNode(g, img, gray)
    assert L.vxGetStatus(n) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    assert L.vxProcessGraph(g) == ovx.NVX_ERROR_NO_GPU


def test_create_query_release(ctx):
    img = L.vxCreateImage(ctx, 640, 480, ovx.VX_DF_IMAGE_RGBX)
    assert ovx.query("vxQueryImage", img, ovx.VX_IMAGE_WIDTH, C.c_uint32) == 640
    assert ovx.query("vxQueryImage", img, ovx.VX_IMAGE_HEIGHT, C.c_uint32) == 480
    assert ovx.query("vxQueryImage", img, ovx.VX_IMAGE_FORMAT, C.c_uint32) == ovx.VX_DF_IMAGE_RGBX
    assert ovx.query("vxQueryImage", img, ovx.VX_IMAGE_PLANES, C.c_size_t) == 1
    w = C.c_uint16()
    assert L.vxQueryImage(img, ovx.VX_IMAGE_WIDTH, C.byref(w), 2) == ovx.VX_ERROR_INVALID_PARAMETERS   # wrong size
    arr = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 2000)
    assert ovx.query("vxQueryArray", arr, ovx.VX_ARRAY_CAPACITY, C.c_size_t) == 2000
    assert ovx.query("vxQueryArray", arr, ovx.VX_ARRAY_ITEMSIZE, C.c_size_t) == 8
    assert ovx.query("vxQueryArray", arr, ovx.VX_ARRAY_ITEMTYPE, C.c_int32) == ovx.NVX_TYPE_POINT2F
    assert ovx.num_items(arr) == 0
    kp = L.vxCreateArray(ctx, ovx.NVX_TYPE_KEYPOINTF, 10)
    assert ovx.query("vxQueryArray", kp, ovx.VX_ARRAY_ITEMSIZE, C.c_size_t) == 28
    assert L.vxGetContext(img) == ctx
    # release sets the handle to NULL; releasing NULL is harmless (feature_tracker.cpp:254-270)
    h = C.c_void_p(img)
    assert L.vxReleaseImage(C.byref(h)) == ovx.VX_SUCCESS and h.value is None
    assert L.vxReleaseImage(C.byref(h)) == ovx.VX_SUCCESS
    for kind in ("Array", "Delay", "Node", "Graph", "Pyramid", "Scalar"):
        z = C.c_void_p(None)
        assert getattr(L, f"vxRelease{kind}")(C.byref(z)) == ovx.VX_SUCCESS
    assert L.vxReleaseArray(C.byref(C.c_void_p(kp))) == ovx.VX_SUCCESS
    # wrong handle type is refused
    assert L.vxReleaseImage(C.byref(C.c_void_p(arr))) == ovx.VX_ERROR_INVALID_REFERENCE


def test_error_objects(ctx):
    for bad in (L.vxCreateImage(ctx, 0, 10, ovx.VX_DF_IMAGE_U8), L.vxCreateImage(ctx, 10, 10, 0x12345678),
                L.vxCreateArray(ctx, 0x7777, 10), L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 0),
                L.vxCreatePyramid(ctx, 0, 0.5, 64, 64, ovx.VX_DF_IMAGE_U8),
                L.vxCreatePyramid(ctx, 4, 0.8408964, 64, 64, ovx.VX_DF_IMAGE_U8),
                L.vxCreatePyramid(ctx, 4, 0.5, 64, 64, ovx.VX_DF_IMAGE_RGBX)):
        assert bad is not None and L.vxGetStatus(bad) != ovx.VX_SUCCESS          # NVXIO_CHECK_REFERENCE would throw
        assert ovx.query("vxQueryReference", bad, ovx.VX_REFERENCE_TYPE, C.c_int32) == ovx.VX_TYPE_ERROR
    assert L.vxGetStatus(None) != ovx.VX_SUCCESS
    img = L.vxCreateImage(ctx, 8, 8, ovx.VX_DF_IMAGE_U8)
    g = L.vxCreateGraph(ctx)
    n = L.vxColorConvertNode(g, None, img)                                        # required parameter missing
    assert L.vxGetStatus(n) == ovx.VX_ERROR_INVALID_PARAMETERS
    arr = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 4)
    n = L.vxColorConvertNode(g, arr, img)                                         # wrong object type
    assert L.vxGetStatus(n) == ovx.VX_ERROR_INVALID_REFERENCE


def test_log_callback(ctx):
    seen = []
    cb = ovx.LOG_CALLBACK(lambda c, r, st, msg: seen.append((st, msg.decode())))
    L.vxRegisterLogCallback(ctx, cb, 0)
    L.vxCreateImage(ctx, 0, 0, ovx.VX_DF_IMAGE_U8)
    assert seen and seen[0][0] == ovx.VX_ERROR_INVALID_DIMENSION and "vxCreateImage" in seen[0][1]


def test_pyramid_levels_and_level_ownership(ctx):
    pyr = L.vxCreatePyramid(ctx, 5, 0.5, 1920, 1080, ovx.VX_DF_IMAGE_U8)
    assert ovx.query("vxQueryPyramid", pyr, ovx.VX_PYRAMID_LEVELS, C.c_size_t) == 5
    dims = []
    for i in range(5):
        lv = L.vxGetPyramidLevel(pyr, i)
        dims.append((ovx.query("vxQueryImage", lv, ovx.VX_IMAGE_WIDTH, C.c_uint32),
                     ovx.query("vxQueryImage", lv, ovx.VX_IMAGE_HEIGHT, C.c_uint32)))
        assert ovx.query("vxQueryReference", lv, ovx.VX_REFERENCE_COUNT, C.c_uint32) == 2   # pyramid's + ours (owned)
        assert ovx.release("Image", lv) == ovx.VX_SUCCESS                                     # feature_tracker.cpp:448
    assert dims == [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68)]              # float ceil sizing
    assert L.vxGetStatus(L.vxGetPyramidLevel(pyr, 5)) == ovx.VX_ERROR_INVALID_PARAMETERS
    # a level that outlives its pyramid stays valid
    lv0 = L.vxGetPyramidLevel(pyr, 0)
    assert ovx.release("Pyramid", pyr) == ovx.VX_SUCCESS
    assert ovx.query("vxQueryImage", lv0, ovx.VX_IMAGE_WIDTH, C.c_uint32) == 1920
    assert ovx.release("Image", lv0) == ovx.VX_SUCCESS


def test_delay_slots_keep_their_index(ctx):
    ex = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 16)
    d = L.vxCreateDelay(ctx, ex, 3)
    assert ovx.release("Array", ex) == ovx.VX_SUCCESS                  # exemplar is cloned (feature_tracker.cpp:300-302)
    assert ovx.query("vxQueryDelay", d, ovx.VX_DELAY_SLOTS, C.c_size_t) == 3
    assert ovx.query("vxQueryDelay", d, ovx.VX_DELAY_TYPE, C.c_int32) == ovx.VX_TYPE_ARRAY
    s = [L.vxGetReferenceFromDelay(d, -i) for i in range(3)]
    assert len(set(s)) == 3
    assert all(ovx.query("vxQueryArray", x, ovx.VX_ARRAY_CAPACITY, C.c_size_t) == 16 for x in s)
    assert L.vxGetStatus(L.vxGetReferenceFromDelay(d, 1)) != ovx.VX_SUCCESS
    assert L.vxGetStatus(L.vxGetReferenceFromDelay(d, -3)) != ovx.VX_SUCCESS
    # borrowed: the count stays 1
    assert ovx.query("vxQueryReference", s[0], ovx.VX_REFERENCE_COUNT, C.c_uint32) == 1
    assert L.vxAgeDelay(d) == ovx.VX_SUCCESS
    assert [L.vxGetReferenceFromDelay(d, -i) for i in range(3)] == s    # handles stay bound to their index
    # pyramids and images can be delayed too (feature_tracker.cpp:287-291, main_feature_tracker.cpp:233-238)
    pex = L.vxCreatePyramid(ctx, 3, 0.5, 64, 48, ovx.VX_DF_IMAGE_U8)
    pd = L.vxCreateDelay(ctx, pex, 2)
    ovx.release("Pyramid", pex)
    assert L.vxGetStatus(pd) == ovx.VX_SUCCESS and L.vxAgeDelay(pd) == ovx.VX_SUCCESS
    g = L.vxCreateGraph(ctx)
    assert L.vxGetStatus(L.vxCreateDelay(ctx, g, 2)) == ovx.VX_ERROR_NOT_SUPPORTED
    assert ovx.release("Delay", d) == ovx.VX_SUCCESS


def _ft_graph(ctx, w=320, h=240, levels=4, cap=500, harris=True, order="ft"):
    """the graph of demos/feature_tracker/feature_tracker.cpp:350-449, nodes created in a chosen order"""
    frame = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    pex = L.vxCreatePyramid(ctx, levels, 0.5, w, h, ovx.VX_DF_IMAGE_U8)
    pyr_delay = L.vxCreateDelay(ctx, pex, 2)
    ovx.release("Pyramid", pex)
    aex = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, cap)
    pts_delay = L.vxCreateDelay(ctx, aex, 2)
    ovx.release("Array", aex)
    kp_curr = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, cap)
    g = L.vxCreateGraph(ctx)
    pyr0 = L.vxGetReferenceFromDelay(pyr_delay, 0)
    gray = L.vxGetPyramidLevel(pyr0, 0)
    eps, it, est = ovx.scalar(ctx, ovx.VX_TYPE_FLOAT32, 0.01), ovx.scalar(ctx, ovx.VX_TYPE_UINT32, 5), ovx.scalar(ctx, ovx.VX_TYPE_BOOL, 0)
    mk = {
        "cvt": lambda: L.vxColorConvertNode(g, frame, gray),
        "pyr": lambda: L.vxGaussianPyramidNode(g, gray, pyr0),
        "lk": lambda: L.vxOpticalFlowPyrLKNode(g, L.vxGetReferenceFromDelay(pyr_delay, -1), pyr0,
                                               L.vxGetReferenceFromDelay(pts_delay, -1), L.vxGetReferenceFromDelay(pts_delay, -1),
                                               kp_curr, ovx.VX_TERM_CRITERIA_BOTH, eps, it, est, 10),
        "det": lambda: (L.nvxHarrisTrackNode(g, gray, L.vxGetReferenceFromDelay(pts_delay, 0), None, kp_curr, 0.04, 100.0, 18, None)
                        if harris else
                        L.nvxFastTrackNode(g, gray, L.vxGetReferenceFromDelay(pts_delay, 0), None, kp_curr, 9, 25, 18, None)),
    }
    names = {"ft": ["cvt", "pyr", "lk", "det"], "reversed": ["det", "lk", "pyr", "cvt"], "mixed": ["lk", "cvt", "det", "pyr"]}[order]
    nodes = {n: mk[n]() for n in names}
    for n in nodes.values():
        assert L.vxGetStatus(n) == ovx.VX_SUCCESS
    return dict(g=g, nodes=nodes, frame=frame, gray=gray, pyr_delay=pyr_delay, pts_delay=pts_delay, kp_curr=kp_curr,
                scalars=(eps, it, est))


def _order(g):
    buf = (C.c_int * 8)()
    n = L.visualslam_amd_graph_order(g, buf, 8)
    return list(buf[:n])


@pytest.mark.parametrize("creation", ["ft", "reversed", "mixed"])
@pytest.mark.parametrize("harris", [True, False])
def test_verify_orders_nodes_by_data_dependency(ctx, creation, harris):
    """the order ColorConvert -> GaussianPyramid -> PyrLK -> Harris/FastTrack is written
    nowhere; vxVerifyGraph must derive it from the data (level 0 aliasing included)."""
    ft = _ft_graph(ctx, harris=harris, order=creation)
    opt = b"-O3"
    assert L.vxSetGraphAttribute(ft["g"], ovx.NVX_GRAPH_VERIFY_OPTIONS, opt, len(opt)) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(ft["g"]) == ovx.VX_SUCCESS
    assert L.vxIsGraphVerified(ft["g"]) == 1
    det = ovx.K_HARRIS_TRACK if harris else ovx.K_FAST_TRACK
    assert _order(ft["g"]) == [ovx.K_COLOR_CONVERT, ovx.K_GAUSSIAN_PYRAMID, ovx.K_PYRLK, det]
    assert ovx.query("vxQueryGraph", ft["g"], ovx.VX_GRAPH_NUMNODES, C.c_uint32) == 4


def test_scalars_and_nodes_can_be_released_after_verify(ctx):
    ft = _ft_graph(ctx)
    assert L.vxVerifyGraph(ft["g"]) == ovx.VX_SUCCESS
    for s in ft["scalars"]:
        assert ovx.release("Scalar", s) == ovx.VX_SUCCESS          # feature_tracker.cpp:445-447: node retains
    assert ovx.release("Image", ft["gray"]) == ovx.VX_SUCCESS        # :448
    for n in ft["nodes"].values():
        assert ovx.release("Node", n) == ovx.VX_SUCCESS              # :264-267: the graph keeps its nodes
    assert ovx.query("vxQueryGraph", ft["g"], ovx.VX_GRAPH_NUMNODES, C.c_uint32) == 4
    assert L.vxIsGraphVerified(ft["g"]) == 1
    assert ovx.release("Graph", ft["g"]) == ovx.VX_SUCCESS


def test_set_parameter_by_index_rebinding(ctx):
    ft = _ft_graph(ctx)
    g, nodes = ft["g"], ft["nodes"]
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    frame2 = L.vxCreateImage(ctx, 320, 240, ovx.VX_DF_IMAGE_RGBX)
    mask = L.vxCreateImage(ctx, 320, 240, ovx.VX_DF_IMAGE_U8)
    # feature_tracker.cpp:216-217: new frame + (possibly NULL) mask every frame, no re-verification
    assert L.vxSetParameterByIndex(nodes["cvt"], 0, frame2) == ovx.VX_SUCCESS
    assert L.vxSetParameterByIndex(nodes["det"], 2, mask) == ovx.VX_SUCCESS
    assert L.vxSetParameterByIndex(nodes["det"], 2, None) == ovx.VX_SUCCESS
    assert L.vxIsGraphVerified(g) == 1
    assert L.vxSetParameterByIndex(nodes["cvt"], 0, None) == ovx.VX_ERROR_INVALID_PARAMETERS      # required
    assert L.vxSetParameterByIndex(nodes["cvt"], 0, mask) == ovx.VX_SUCCESS                        # other metadata ...
    assert L.vxIsGraphVerified(g) == 0                                                             # ... needs re-verify
    assert L.vxSetParameterByIndex(nodes["cvt"], 7, frame2) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxSetParameterByIndex(nodes["det"], 1, frame2) == ovx.VX_ERROR_INVALID_REFERENCE      # array expected
    # the old frame was released by the node, the new one retained
    assert ovx.query("vxQueryReference", ft["frame"], ovx.VX_REFERENCE_COUNT, C.c_uint32) == 1
    assert ovx.query("vxQueryReference", mask, ovx.VX_REFERENCE_COUNT, C.c_uint32) == 2


def test_rebinding_an_image_produced_inside_the_graph_unverifies(ctx):
    """keeping the graph verified across vxSetParameterByIndex is only sound for images nobody in the graph writes
    (the camera frame, the mask); an image another node produces -- or an ROI view of one -- changes the order"""
    a, b, c, d = (L.vxCreateImage(ctx, 32, 32, ovx.VX_DF_IMAGE_U8) for _ in range(4))
    g = L.vxCreateGraph(ctx)
    n1 = L.nvxCopyImageNode(g, a, b)          # creation order: consumer first, so a stale order would be wrong
    n0 = L.nvxCopyImageNode(g, c, d)
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    ext = L.vxCreateImage(ctx, 32, 32, ovx.VX_DF_IMAGE_U8)
    assert L.vxSetParameterByIndex(n1, 0, ext) == ovx.VX_SUCCESS and L.vxIsGraphVerified(g) == 1     # external -> external
    assert L.vxSetParameterByIndex(n1, 0, d) == ovx.VX_SUCCESS and L.vxIsGraphVerified(g) == 0       # now reads n0's output
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    assert L.vxSetParameterByIndex(n1, 0, ext) == ovx.VX_SUCCESS and L.vxIsGraphVerified(g) == 0     # dependency removed
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    rect = ovx.vx_rectangle_t(0, 0, 32, 32)
    view = L.vxCreateImageFromROI(d, C.byref(rect))
    assert L.vxGetStatus(view) == ovx.VX_SUCCESS
    assert L.vxSetParameterByIndex(n1, 0, view) == ovx.VX_SUCCESS and L.vxIsGraphVerified(g) == 0    # a view of n0's output
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    assert L.vxSetParameterByIndex(n0, 1, ext) == ovx.VX_SUCCESS and L.vxIsGraphVerified(g) == 0     # an output rebound


def test_pyrlk_node_reads_termination_and_window_from_its_scalars(ctx):
    """indices 5 (termination) and 9 (window_dimension) of vxOpticalFlowPyrLKNode are scalar parameters: rebinding
    them takes effect (verification re-checks the window range), as for every other scalar of the node"""
    ft = _ft_graph(ctx)
    g, lk = ft["g"], ft["nodes"]["lk"]
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    p = L.vxGetParameterByIndex(lk, 9)
    ref = C.c_void_p()
    assert L.vxQueryParameter(p, ovx.VX_PARAMETER_REF, C.byref(ref), C.sizeof(ref)) == ovx.VX_SUCCESS
    assert ovx.query("vxQueryScalar", ref.value, ovx.VX_SCALAR_TYPE, C.c_int32) == ovx.VX_TYPE_SIZE
    big = ovx.scalar(ctx, ovx.VX_TYPE_SIZE, 64)
    assert L.vxSetParameterByIndex(lk, 9, big) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_VALUE            # 3..32 (main_feature_tracker.cpp:103-106)
    ok = ovx.scalar(ctx, ovx.VX_TYPE_SIZE, 21)
    assert L.vxSetParameterByIndex(lk, 9, ok) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS


def test_verify_rejects_bad_graphs(ctx):
    g = L.vxCreateGraph(ctx)
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_GRAPH                     # empty
    a = L.vxCreateImage(ctx, 16, 16, ovx.VX_DF_IMAGE_U8)
    b = L.vxCreateImage(ctx, 16, 16, ovx.VX_DF_IMAGE_U8)
    L.nvxCopyImageNode(g, a, b)
    L.nvxCopyImageNode(g, b, a)
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_GRAPH                     # cycle
    g2 = L.vxCreateGraph(ctx)
    c = L.vxCreateImage(ctx, 16, 16, ovx.VX_DF_IMAGE_U8)
    L.nvxCopyImageNode(g2, a, c)
    L.nvxCopyImageNode(g2, b, c)
    assert L.vxVerifyGraph(g2) == ovx.VX_ERROR_INVALID_GRAPH                    # two writers
    g3 = L.vxCreateGraph(ctx)
    kp = L.vxCreateArray(ctx, ovx.VX_TYPE_KEYPOINT, 10)
    L.nvxHarrisTrackNode(g3, a, kp, None, None, 0.04, 100.0, 18, None)
    assert L.vxVerifyGraph(g3) == ovx.VX_ERROR_NOT_SUPPORTED                    # integer vx_keypoint_t not offered
    g4 = L.vxCreateGraph(ctx)
    p2 = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 10)
    L.nvxHarrisTrackNode(g4, a, p2, None, None, 0.04, 100.0, 0, None)
    assert L.vxVerifyGraph(g4) == ovx.VX_ERROR_INVALID_VALUE                    # cell_size 0


def test_scalar_read_write(ctx):
    s = ovx.scalar(ctx, ovx.VX_TYPE_FLOAT32, 0.25)
    v = C.c_float()
    assert L.vxReadScalarValue(s, C.byref(v)) == 0 and v.value == 0.25
    v.value = 4.5
    assert L.vxWriteScalarValue(s, C.byref(v)) == 0
    w = C.c_float()
    assert L.vxCopyScalar(s, C.byref(w), ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST) == 0 and w.value == 4.5
    assert ovx.query("vxQueryScalar", s, 0x80D00, C.c_int32) == ovx.VX_TYPE_FLOAT32


def test_context_release_reclaims_everything(monkeypatch):
    monkeypatch.setenv("VISUALSLAM_AMD_ALLOW_NO_GPU", "1")
    c = L.vxCreateContext()
    ft = _ft_graph(c)
    assert L.vxVerifyGraph(ft["g"]) == ovx.VX_SUCCESS
    h = C.c_void_p(c)
    assert L.vxReleaseContext(C.byref(h)) == ovx.VX_SUCCESS          # nothing released by hand: no crash, no leak report
    assert L.vxGetStatus(ft["g"]) == ovx.VX_ERROR_INVALID_REFERENCE  # handles are dead afterwards


def test_matrix_object(ctx):
    """vx_matrix as demos/video_stabilizer/stabilizer.cpp:170-176 uses it: 3x3 FLOAT32, written and read whole"""
    import numpy as np
    m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    assert L.vxGetStatus(m) == ovx.VX_SUCCESS
    assert ovx.query("vxQueryReference", m, ovx.VX_REFERENCE_TYPE, C.c_int32) == ovx.VX_TYPE_MATRIX
    assert ovx.query("vxQueryMatrix", m, ovx.VX_MATRIX_TYPE, C.c_int32) == ovx.VX_TYPE_FLOAT32
    assert ovx.query("vxQueryMatrix", m, ovx.VX_MATRIX_ROWS, C.c_size_t) == 3
    assert ovx.query("vxQueryMatrix", m, ovx.VX_MATRIX_COLUMNS, C.c_size_t) == 3
    assert ovx.query("vxQueryMatrix", m, ovx.VX_MATRIX_SIZE, C.c_size_t) == 36
    eye = np.eye(3, dtype=np.float32) * 2 + np.arange(9, dtype=np.float32).reshape(3, 3)
    assert L.vxCopyMatrix(m, eye.ctypes.data, ovx.VX_WRITE_ONLY, ovx.VX_MEMORY_TYPE_HOST) == ovx.VX_SUCCESS
    back = np.zeros((3, 3), np.float32)
    assert L.vxCopyMatrix(m, back.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST) == ovx.VX_SUCCESS
    assert (back == eye).all()
    back[:] = 0
    assert L.vxReadMatrix(m, back.ctypes.data) == ovx.VX_SUCCESS and (back == eye).all()      # OpenVX 1.0 spellings
    assert L.vxWriteMatrix(m, np.zeros(9, np.float32).ctypes.data) == ovx.VX_SUCCESS
    assert L.vxReadMatrix(m, back.ctypes.data) == ovx.VX_SUCCESS and (back == 0).all()
    assert L.vxCopyMatrix(m, None, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxCopyMatrix(m, back.ctypes.data, 0x7777, ovx.VX_MEMORY_TYPE_HOST) == ovx.VX_ERROR_INVALID_PARAMETERS
    img = L.vxCreateImage(ctx, 8, 8, ovx.VX_DF_IMAGE_U8)
    assert L.vxQueryMatrix(img, ovx.VX_MATRIX_ROWS, C.byref(C.c_size_t()), 8) == ovx.VX_ERROR_INVALID_REFERENCE
    for bad in (L.vxCreateMatrix(ctx, ovx.VX_TYPE_BOOL, 3, 3), L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 0, 3),
                L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 0)):
        assert bad is not None and L.vxGetStatus(bad) != ovx.VX_SUCCESS
    h = C.c_void_p(m)
    assert L.vxReleaseMatrix(C.byref(h)) == ovx.VX_SUCCESS and h.value is None
    assert L.vxReleaseMatrix(C.byref(h)) == ovx.VX_SUCCESS
    assert L.vxReleaseMatrix(C.byref(C.c_void_p(img))) == ovx.VX_ERROR_INVALID_REFERENCE


def test_virtual_objects(ctx):
    """virtual images/arrays (stabilizer.cpp:230-262): created against a graph; a 0x0 virtual image takes its
    size from the node that writes it when the graph is verified"""
    g = L.vxCreateGraph(ctx)
    frame = L.vxCreateImage(ctx, 96, 64, ovx.VX_DF_IMAGE_RGBX)
    gray = L.vxCreateVirtualImage(g, 0, 0, ovx.VX_DF_IMAGE_U8)
    assert L.vxGetStatus(gray) == ovx.VX_SUCCESS
    pyr = L.vxCreatePyramid(ctx, 3, 0.5, 96, 64, ovx.VX_DF_IMAGE_U8)
    copy = L.vxCreateImage(ctx, 96, 64, ovx.VX_DF_IMAGE_U8)
    assert L.vxGetStatus(L.vxColorConvertNode(g, frame, gray)) == ovx.VX_SUCCESS
    assert L.vxGetStatus(L.nvxCopyImageNode(g, gray, copy)) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    assert ovx.query("vxQueryImage", gray, ovx.VX_IMAGE_WIDTH, C.c_uint32) == 96
    assert ovx.query("vxQueryImage", gray, ovx.VX_IMAGE_HEIGHT, C.c_uint32) == 64
    assert _order(g) == [ovx.K_COLOR_CONVERT, ovx.K_COPY_IMAGE]
    sized = L.vxCreateVirtualImage(g, 32, 16, ovx.VX_DF_IMAGE_U8)
    assert ovx.query("vxQueryImage", sized, ovx.VX_IMAGE_WIDTH, C.c_uint32) == 32
    assert L.vxGetStatus(L.vxCreateVirtualImage(g, 32, 0, ovx.VX_DF_IMAGE_U8)) == ovx.VX_ERROR_INVALID_DIMENSION
    assert L.vxGetStatus(L.vxCreateVirtualImage(g, 32, 32, 0x12345678)) == ovx.VX_ERROR_INVALID_FORMAT
    assert L.vxCreateVirtualImage(ctx, 32, 32, ovx.VX_DF_IMAGE_U8) is None          # a context is not a graph
    arr = L.vxCreateVirtualArray(g, ovx.VX_TYPE_UINT8, 100)
    assert L.vxGetStatus(arr) == ovx.VX_SUCCESS
    assert ovx.query("vxQueryArray", arr, ovx.VX_ARRAY_CAPACITY, C.c_size_t) == 100
    assert ovx.query("vxQueryArray", arr, ovx.VX_ARRAY_ITEMSIZE, C.c_size_t) == 1
    assert L.vxCreateVirtualArray(ctx, ovx.VX_TYPE_UINT8, 100) is None
    ovx.release("Pyramid", pyr)


def _homography_graph(ctx, method=None, pts_type=None, mask_type=None, mshape=(3, 3), mtype=None, thr=3.0):
    g = L.vxCreateGraph(ctx)
    pts_type = ovx.NVX_TYPE_POINT2F if pts_type is None else pts_type
    a = L.vxCreateArray(ctx, pts_type, 64)
    b = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 64)
    m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32 if mtype is None else mtype, *mshape)
    k = L.vxCreateVirtualArray(g, ovx.VX_TYPE_UINT8 if mask_type is None else mask_type, 64)
    n = L.nvxFindHomographyNode(g, a, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC if method is None else method,
                                thr, 2000, 10, 0.995, 0.45, k)
    return g, n


def test_find_homography_node_verification(ctx):
    """nvxFindHomographyNode with the video stabilizer's arguments (stabilizer.cpp:264-273): metadata is checked
    when the graph is verified, nothing runs without a device"""
    g, n = _homography_graph(ctx)
    assert L.vxGetStatus(n) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    assert _order(g) == [ovx.K_FIND_HOMOGRAPHY]
    assert L.vxProcessGraph(g) == ovx.NVX_ERROR_NO_GPU
    g, _ = _homography_graph(ctx, method=ovx.NVX_FIND_HOMOGRAPHY_METHOD_LMEDS)
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_NOT_SUPPORTED
    g, _ = _homography_graph(ctx, pts_type=ovx.NVX_TYPE_KEYPOINTF)
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_TYPE
    g, _ = _homography_graph(ctx, mask_type=ovx.NVX_TYPE_POINT2F)
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_TYPE
    g, _ = _homography_graph(ctx, mshape=(3, 2))
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_DIMENSION
    g, _ = _homography_graph(ctx, mtype=ovx.VX_TYPE_UINT8)
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_DIMENSION
    g, _ = _homography_graph(ctx, thr=0.0)
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_VALUE
    # required parameters
    g = L.vxCreateGraph(ctx)
    a = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 8)
    m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    n = L.nvxFindHomographyNode(g, a, None, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, None)
    assert L.vxGetStatus(n) == ovx.VX_ERROR_INVALID_PARAMETERS
    n = L.nvxFindHomographyNode(g, a, a, a, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, None)
    assert L.vxGetStatus(n) == ovx.VX_ERROR_INVALID_REFERENCE
    # the immediate form cannot run either
    assert L.nvxuFindHomography(ctx, a, a, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, None) \
        == ovx.NVX_ERROR_NO_GPU


def test_tracker_graph_feeding_homography_orders_after_lk(ctx):
    """BASELINE config C5: the PyrLK matches feed nvxFindHomographyNode in the same graph"""
    ft = _ft_graph(ctx, order="mixed")
    m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    n = L.nvxFindHomographyNode(ft["g"], L.vxGetReferenceFromDelay(ft["pts_delay"], -1), ft["kp_curr"], m,
                                ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, None)
    assert L.vxGetStatus(n) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(ft["g"]) == ovx.VX_SUCCESS
    order = _order(ft["g"])
    assert order.index(ovx.K_FIND_HOMOGRAPHY) > order.index(ovx.K_PYRLK)
    assert order[:3] == [ovx.K_COLOR_CONVERT, ovx.K_GAUSSIAN_PYRAMID, ovx.K_PYRLK]
    # the detector node's merge pass and the homography share a launch: both only read what PyrLK wrote
    assert ovx.graph_fusions(ft["g"])["merge_in_homography"] == 1


def test_homography_and_merge_share_a_launch_only_when_the_later_node_can_move_up(ctx, monkeypatch):
    """vxVerifyGraph pairs a FindHomography node with the detector node whose cell pass rides in the PyrLK launch (the merge
    pass is all that is left of it): whichever comes first in the order launches for both.  No pair when the detector's cell
    pass could not be moved into the PyrLK launch (here: a node after PyrLK writes the detector's mask), and none with
    VISUALSLAM_AMD_NO_FUSION."""
    def graph(mask_written_late):
        ft = _ft_graph(ctx)
        m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
        n = L.nvxFindHomographyNode(ft["g"], L.vxGetReferenceFromDelay(ft["pts_delay"], -1), ft["kp_curr"], m,
                                    ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, None)
        assert L.vxGetStatus(n) == ovx.VX_SUCCESS
        if mask_written_late:
            src = L.vxCreateImage(ctx, 320, 240, ovx.VX_DF_IMAGE_U8)
            msk = L.vxCreateImage(ctx, 320, 240, ovx.VX_DF_IMAGE_U8)
            # created last, the copy is ordered after PyrLK and FindHomography (ties go by creation order); the detector waits for it
            cp = L.nvxCopyImageNode(ft["g"], src, msk)
            assert L.vxGetStatus(cp) == ovx.VX_SUCCESS
            assert L.vxSetParameterByIndex(ft["nodes"]["det"], 2, msk) == ovx.VX_SUCCESS
        assert L.vxVerifyGraph(ft["g"]) == ovx.VX_SUCCESS
        return ft["g"]

    f = ovx.graph_fusions(graph(False))
    assert f["cells_in_pyrlk"] == 1 and f["merge_in_homography"] == 1
    g = graph(True)
    order = _order(g)
    f = ovx.graph_fusions(g)
    assert order.index(ovx.K_COPY_IMAGE) > order.index(ovx.K_PYRLK)
    assert f["cells_in_pyrlk"] == 0 and f["merge_in_homography"] == 0
    monkeypatch.setenv("VISUALSLAM_AMD_NO_FUSION", "1")
    assert ovx.graph_fusions(graph(False))["merge_in_homography"] == 0


# ---- user kernels, generic nodes, delays of matrices, uniform / ROI images (video stabilizer) ----------
_KEEP = []   # ctypes callbacks must outlive the contexts that hold them


def _add_kernel(ctx, name, func, nparams, validate=None, params=()):
    kid = C.c_int32()
    assert L.vxAllocateUserKernelId(ctx, C.byref(kid)) == ovx.VX_SUCCESS
    f = ovx.KERNEL_F(func)
    v = ovx.VALIDATE_F(validate) if validate else C.cast(None, ovx.VALIDATE_F)
    none_f = C.cast(None, ovx.KERNEL_F)
    _KEEP.extend([f, v])
    k = L.vxAddUserKernel(ctx, name.encode(), kid.value, f, nparams, v, none_f, none_f)
    assert L.vxGetStatus(k) == ovx.VX_SUCCESS
    for i, (d, t) in enumerate(params):
        assert L.vxAddParameterToKernel(k, i, d, t, ovx.VX_PARAMETER_STATE_REQUIRED) == ovx.VX_SUCCESS
    return k, kid.value


def test_user_kernel_registration(ctx):
    """the sequence of demos/video_stabilizer/homography_filter_node.cpp:205-256"""
    k, kid = _add_kernel(ctx, "example.test.a", lambda n, p, c: 0, 2,
                         params=[(ovx.VX_INPUT, ovx.VX_TYPE_MATRIX)])
    assert L.vxGetStatus(L.vxGetKernelByName(ctx, b"example.test.a")) != ovx.VX_SUCCESS      # not finalized yet
    assert L.vxFinalizeKernel(k) == ovx.VX_ERROR_INVALID_PARAMETERS                           # parameter 1 undescribed
    g = L.vxCreateGraph(ctx)
    assert L.vxGetStatus(L.vxCreateGenericNode(g, k)) == ovx.VX_ERROR_INVALID_REFERENCE
    assert L.vxAddParameterToKernel(k, 1, ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX, ovx.VX_PARAMETER_STATE_REQUIRED) == ovx.VX_SUCCESS
    assert L.vxAddParameterToKernel(k, 2, ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX, ovx.VX_PARAMETER_STATE_REQUIRED) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxAddParameterToKernel(k, 1, 0x1234, ovx.VX_TYPE_MATRIX, ovx.VX_PARAMETER_STATE_REQUIRED) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxFinalizeKernel(k) == ovx.VX_SUCCESS
    assert L.vxAddParameterToKernel(k, 1, ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX, ovx.VX_PARAMETER_STATE_REQUIRED) == ovx.VX_ERROR_NOT_SUPPORTED
    assert ovx.query("vxQueryKernel", k, ovx.VX_KERNEL_PARAMETERS, C.c_uint32) == 2
    assert ovx.query("vxQueryKernel", k, ovx.VX_KERNEL_ENUM, C.c_int32) == kid
    name = C.create_string_buffer(256)
    assert L.vxQueryKernel(k, ovx.VX_KERNEL_NAME, name, 256) == ovx.VX_SUCCESS and name.value == b"example.test.a"
    assert ovx.release("Kernel", k) == ovx.VX_SUCCESS            # the registry keeps it (homography_filter_node.cpp:247)
    k2 = L.vxGetKernelByName(ctx, b"example.test.a")
    assert k2 == k and L.vxGetStatus(k2) == ovx.VX_SUCCESS
    assert L.vxGetKernelByEnum(ctx, kid) == k
    assert ovx.release("Kernel", k) == ovx.VX_SUCCESS
    # a second kernel gets a different id; a taken name is refused
    _, kid2 = _add_kernel(ctx, "example.test.b", lambda n, p, c: 0, 1, params=[(ovx.VX_INPUT, ovx.VX_TYPE_IMAGE)])
    assert kid2 != kid
    f = ovx.KERNEL_F(lambda n, p, c: 0)
    _KEEP.append(f)
    none_f, none_v = C.cast(None, ovx.KERNEL_F), C.cast(None, ovx.VALIDATE_F)
    assert L.vxGetStatus(L.vxAddUserKernel(ctx, b"example.test.a", kid2 + 1, f, 1, none_v, none_f, none_f)) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxGetStatus(L.vxAddUserKernel(ctx, b"", kid2 + 1, f, 1, none_v, none_f, none_f)) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxGetStatus(L.vxAddUserKernel(ctx, b"x", kid2 + 1, f, 0, none_v, none_f, none_f)) == ovx.VX_ERROR_INVALID_PARAMETERS
    # removal takes it out of the registry
    assert L.vxRemoveKernel(L.vxGetKernelByName(ctx, b"example.test.a")) == ovx.VX_SUCCESS
    assert L.vxGetStatus(L.vxGetKernelByName(ctx, b"example.test.a")) != ovx.VX_SUCCESS


def test_user_node_verification(ctx):
    """vxCreateGenericNode + vxSetParameterByIndex + the validator's meta formats (homography_filter_node.cpp:162-202,259-280)"""
    seen = []
    meta_rows = [3]

    def validate(node, params, num, metas):
        seen.append(("validate", num, params[0], params[1]))
        t, r, c = C.c_int32(ovx.VX_TYPE_FLOAT32), C.c_size_t(meta_rows[0]), C.c_size_t(3)
        L.vxSetMetaFormatAttribute(metas[1], ovx.VX_MATRIX_TYPE, C.byref(t), 4)
        L.vxSetMetaFormatAttribute(metas[1], ovx.VX_MATRIX_ROWS, C.byref(r), 8)
        L.vxSetMetaFormatAttribute(metas[1], ovx.VX_MATRIX_COLUMNS, C.byref(c), 8)
        rows = ovx.query("vxQueryMatrix", params[0], ovx.VX_MATRIX_ROWS, C.c_size_t)
        return ovx.VX_SUCCESS if rows == 3 else ovx.VX_ERROR_INVALID_PARAMETERS

    k, _ = _add_kernel(ctx, "example.test.filter", lambda n, p, c: 0, 2, validate,
                       [(ovx.VX_INPUT, ovx.VX_TYPE_MATRIX), (ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX)])
    assert L.vxFinalizeKernel(k) == ovx.VX_SUCCESS
    g = L.vxCreateGraph(ctx)
    n = L.vxCreateGenericNode(g, k)
    assert L.vxGetStatus(n) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_PARAMETERS and not seen        # required parameters unset
    a, b = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3), L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    img = L.vxCreateImage(ctx, 8, 8, ovx.VX_DF_IMAGE_U8)
    assert L.vxSetParameterByIndex(n, 0, img) == ovx.VX_ERROR_INVALID_REFERENCE      # the kernel says: a matrix
    assert L.vxSetParameterByIndex(n, 0, a) == ovx.VX_SUCCESS
    assert L.vxSetParameterByIndex(n, 1, b) == ovx.VX_SUCCESS
    assert L.vxSetParameterByIndex(n, 2, b) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    assert seen == [("validate", 2, a, b)]
    assert _order(g) == [ovx.K_USER]
    assert L.vxProcessGraph(g) == ovx.NVX_ERROR_NO_GPU                               # graphs need the device, host nodes included
    # the output disagrees with what the validator announced
    meta_rows[0] = 2
    assert L.vxSetParameterByIndex(n, 1, L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_PARAMETERS
    meta_rows[0] = 3
    # the validator's own verdict is the graph's
    assert L.vxSetParameterByIndex(n, 0, L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 2)) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_PARAMETERS


def test_stabilizer_graph_shape_orders_by_data(ctx):
    """the node set of demos/video_stabilizer/stabilizer.cpp:228-316 with Python stand-ins for the three user kernels:
    order follows the data, the delay parameter standing for every slot of the delay"""
    specs = {
        "filter": [(ovx.VX_INPUT, ovx.VX_TYPE_MATRIX), (ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX), (ovx.VX_INPUT, ovx.VX_TYPE_IMAGE), (ovx.VX_INPUT, ovx.VX_TYPE_ARRAY)],
        "smoother": [(ovx.VX_INPUT, ovx.VX_TYPE_DELAY), (ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX)],
        "truncate": [(ovx.VX_INPUT, ovx.VX_TYPE_MATRIX), (ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX), (ovx.VX_INPUT, ovx.VX_TYPE_IMAGE), (ovx.VX_INPUT, ovx.VX_TYPE_SCALAR)],
    }
    kernels = {}
    for name, sp in specs.items():
        k, _ = _add_kernel(ctx, "example.shape." + name, lambda n, p, c: 0, len(sp), None, sp)
        assert L.vxFinalizeKernel(k) == ovx.VX_SUCCESS
        kernels[name] = k
    w, h, cap = 320, 240, 1000
    frame = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    pex = L.vxCreatePyramid(ctx, 4, 0.5, w, h, ovx.VX_DF_IMAGE_U8)
    pyr_delay = L.vxCreateDelay(ctx, pex, 2)
    aex = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, cap)
    pts_delay = L.vxCreateDelay(ctx, aex, 2)
    smoothed = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    mat_delay = L.vxCreateDelay(ctx, smoothed, 11)
    assert L.vxGetStatus(mat_delay) == ovx.VX_SUCCESS
    assert ovx.query("vxQueryDelay", mat_delay, ovx.VX_DELAY_TYPE, C.c_int32) == ovx.VX_TYPE_MATRIX
    frames_delay = L.vxCreateDelay(ctx, frame, 7)
    stabilized = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    g = L.vxCreateGraph(ctx)
    gray = L.vxCreateVirtualImage(g, 0, 0, ovx.VX_DF_IMAGE_U8)
    kp = L.vxCreateVirtualArray(g, ovx.NVX_TYPE_POINT2F, cap)
    mask = L.vxCreateVirtualArray(g, ovx.VX_TYPE_UINT8, cap)
    homography = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    truncated = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    eps, it, est = ovx.scalar(ctx, ovx.VX_TYPE_FLOAT32, 0.01), ovx.scalar(ctx, ovx.VX_TYPE_UINT32, 5), ovx.scalar(ctx, ovx.VX_TYPE_BOOL, 0)
    margin = ovx.scalar(ctx, ovx.VX_TYPE_FLOAT32, 0.05)

    def generic(name, *params):
        n = L.vxCreateGenericNode(g, kernels[name])
        assert L.vxGetStatus(n) == ovx.VX_SUCCESS
        for i, p in enumerate(params):
            assert L.vxSetParameterByIndex(n, i, p) == ovx.VX_SUCCESS
        return n

    # created in an order that is NOT the execution order
    nodes = [
        L.vxWarpPerspectiveNode(g, L.vxGetReferenceFromDelay(frames_delay, -6), truncated, ovx.VX_INTERPOLATION_BILINEAR, stabilized),
        generic("truncate", smoothed, truncated, frame, margin),
        generic("smoother", mat_delay, smoothed),
        generic("filter", homography, L.vxGetReferenceFromDelay(mat_delay, 0), frame, mask),
        L.nvxFindHomographyNode(g, L.vxGetReferenceFromDelay(pts_delay, -1), kp, homography,
                                ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, mask),
        L.nvxHarrisTrackNode(g, gray, L.vxGetReferenceFromDelay(pts_delay, 0), None, kp, 0.04, 100.0, 18, None),
        L.vxOpticalFlowPyrLKNode(g, L.vxGetReferenceFromDelay(pyr_delay, -1), L.vxGetReferenceFromDelay(pyr_delay, 0),
                                 L.vxGetReferenceFromDelay(pts_delay, -1), L.vxGetReferenceFromDelay(pts_delay, -1),
                                 kp, ovx.VX_TERM_CRITERIA_BOTH, eps, it, est, 10),
        L.vxGaussianPyramidNode(g, gray, L.vxGetReferenceFromDelay(pyr_delay, 0)),
        L.nvxCopyImageNode(g, frame, L.vxGetReferenceFromDelay(frames_delay, 0)),
        L.vxColorConvertNode(g, frame, gray),
    ]
    for n in nodes:
        assert L.vxGetStatus(n) == ovx.VX_SUCCESS
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    buf = (C.c_int * 16)()
    order = list(buf[:L.visualslam_amd_graph_order(g, buf, 16)])
    assert len(order) == 10
    pos = {k: [i for i, x in enumerate(order) if x == k] for k in set(order)}
    assert pos[ovx.K_COLOR_CONVERT][0] < pos[ovx.K_GAUSSIAN_PYRAMID][0] < pos[ovx.K_PYRLK][0] < pos[ovx.K_FIND_HOMOGRAPHY][0]
    users = pos[ovx.K_USER]
    assert len(users) == 3 and pos[ovx.K_FIND_HOMOGRAPHY][0] < users[0] and users[2] < pos[ovx.K_WARP_PERSPECTIVE][0]
    assert pos[ovx.K_PYRLK][0] < pos[ovx.K_HARRIS_TRACK][0]
    assert ovx.query("vxQueryImage", gray, ovx.VX_IMAGE_WIDTH, C.c_uint32) == w
    # the stabilizer's frame: ColorConvert inside the pyramid launch (virtual grey = level 0), cell pass inside PyrLK, and the
    # detector's merge pass -- ordered after the three host nodes and the warp -- inside the FindHomography launch
    assert ovx.graph_fusions(g) == {"color_convert_in_pyramid": 1, "cells_in_pyrlk": 1, "virtual_gray_is_level0": 1, "merge_in_homography": 1}


def test_delay_of_matrices(ctx):
    """stabilizer.cpp:357-375,438: slots are clones of the exemplar, written and aged independently"""
    import numpy as np
    ex = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    seed = np.arange(9, dtype=np.float32)
    assert L.vxCopyMatrix(ex, seed.ctypes.data, ovx.VX_WRITE_ONLY, ovx.VX_MEMORY_TYPE_HOST) == 0
    d = L.vxCreateDelay(ctx, ex, 3)
    ovx.release("Matrix", ex)
    slots = [L.vxGetReferenceFromDelay(d, -i) for i in range(3)]
    got = np.zeros(9, np.float32)
    for i, s in enumerate(slots):
        assert L.vxCopyMatrix(s, got.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST) == 0 and (got == seed).all()
        v = np.full(9, float(i), np.float32)
        assert L.vxCopyMatrix(s, v.ctypes.data, ovx.VX_WRITE_ONLY, ovx.VX_MEMORY_TYPE_HOST) == 0
    assert L.vxAgeDelay(d) == ovx.VX_SUCCESS
    vals = []
    for s in slots:                                # what was at 0 is at -1, -1 at -2, the oldest wraps to 0
        L.vxCopyMatrix(s, got.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST)
        vals.append(float(got[0]))
    assert vals == [2.0, 0.0, 1.0]


def test_uniform_and_roi_images(ctx):
    """stabilizer.cpp:399-404 (vxCreateUniformImage), main_video_stabilizer.cpp:225-235 (valid region, ROI views)"""
    px = ovx.vx_pixel_value_t()
    px.RGBX[:] = [1, 2, 3, 4]
    u = L.vxCreateUniformImage(ctx, 64, 32, ovx.VX_DF_IMAGE_RGBX, C.byref(px))
    assert L.vxGetStatus(u) == ovx.VX_SUCCESS and ovx.query("vxQueryImage", u, ovx.VX_IMAGE_WIDTH, C.c_uint32) == 64
    assert L.vxGetStatus(L.vxCreateUniformImage(ctx, 64, 32, ovx.VX_DF_IMAGE_RGBX, None)) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxGetStatus(L.vxCreateUniformImage(ctx, 0, 32, ovx.VX_DF_IMAGE_RGBX, C.byref(px))) == ovx.VX_ERROR_INVALID_DIMENSION
    r = ovx.vx_rectangle_t()
    assert L.vxGetValidRegionImage(u, C.byref(r)) == ovx.VX_SUCCESS
    assert (r.start_x, r.start_y, r.end_x, r.end_y) == (0, 0, 64, 32)
    left = L.vxCreateImageFromROI(u, C.byref(ovx.vx_rectangle_t(0, 0, 32, 32)))
    right = L.vxCreateImageFromROI(u, C.byref(ovx.vx_rectangle_t(32, 0, 64, 32)))
    for v in (left, right):
        assert L.vxGetStatus(v) == ovx.VX_SUCCESS
        assert ovx.query("vxQueryImage", v, ovx.VX_IMAGE_WIDTH, C.c_uint32) == 32
        assert ovx.query("vxQueryImage", v, ovx.VX_IMAGE_FORMAT, C.c_uint32) == ovx.VX_DF_IMAGE_RGBX
    assert ovx.query("vxQueryReference", u, ovx.VX_REFERENCE_COUNT, C.c_uint32) == 3          # the views keep the parent
    assert L.vxGetStatus(L.vxCreateImageFromROI(u, C.byref(ovx.vx_rectangle_t(32, 0, 65, 32)))) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxGetStatus(L.vxCreateImageFromROI(u, C.byref(ovx.vx_rectangle_t(5, 5, 5, 9)))) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert ovx.release("Image", u) == ovx.VX_SUCCESS                                           # views outlive the handle
    assert ovx.query("vxQueryImage", left, ovx.VX_IMAGE_HEIGHT, C.c_uint32) == 32
    # two nodes writing two views of one image cannot be ordered by data: the graph is refused
    g = L.vxCreateGraph(ctx)
    a = L.vxCreateImage(ctx, 32, 32, ovx.VX_DF_IMAGE_RGBX)
    L.nvxCopyImageNode(g, a, left)
    L.nvxCopyImageNode(g, a, L.vxCreateImageFromROI(left, C.byref(ovx.vx_rectangle_t(0, 0, 32, 32))))
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_GRAPH


def test_warp_perspective_node_verification(ctx):
    a = L.vxCreateImage(ctx, 64, 48, ovx.VX_DF_IMAGE_RGBX)
    b = L.vxCreateImage(ctx, 64, 48, ovx.VX_DF_IMAGE_RGBX)
    u8 = L.vxCreateImage(ctx, 64, 48, ovx.VX_DF_IMAGE_U8)
    m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    m23 = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 2, 3)

    def verify(*args):
        g = L.vxCreateGraph(ctx)
        n = L.vxWarpPerspectiveNode(g, *args)
        return n, (L.vxVerifyGraph(g) if L.vxGetStatus(n) == ovx.VX_SUCCESS else L.vxGetStatus(n))

    n, st = verify(a, m, ovx.VX_INTERPOLATION_BILINEAR, b)
    assert st == ovx.VX_SUCCESS
    bd = ovx.vx_border_t(ovx.VX_BORDER_CONSTANT, (C.c_uint8 * 4)(9, 8, 7, 6))
    assert L.vxSetNodeAttribute(n, ovx.VX_NODE_BORDER, C.byref(bd), C.sizeof(bd)) == ovx.VX_SUCCESS
    back = ovx.vx_border_t()
    assert L.vxQueryNode(n, ovx.VX_NODE_BORDER, C.byref(back), C.sizeof(back)) == 0 and list(back.constant_value) == [9, 8, 7, 6]
    bd.mode = ovx.VX_BORDER_REPLICATE
    assert L.vxSetNodeAttribute(n, ovx.VX_NODE_BORDER, C.byref(bd), C.sizeof(bd)) == ovx.VX_ERROR_NOT_SUPPORTED
    assert verify(a, m, ovx.VX_INTERPOLATION_NEAREST_NEIGHBOR, b)[1] == ovx.VX_SUCCESS
    assert verify(a, m, ovx.VX_INTERPOLATION_AREA, b)[1] == ovx.VX_ERROR_NOT_SUPPORTED
    assert verify(a, m, ovx.VX_INTERPOLATION_BILINEAR, u8)[1] == ovx.VX_ERROR_INVALID_FORMAT
    assert verify(a, m23, ovx.VX_INTERPOLATION_BILINEAR, b)[1] == ovx.VX_ERROR_INVALID_DIMENSION
    assert verify(a, None, ovx.VX_INTERPOLATION_BILINEAR, b)[1] == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxuWarpPerspective(ctx, a, m, ovx.VX_INTERPOLATION_BILINEAR, b) == ovx.NVX_ERROR_NO_GPU


def test_objects_shared_by_graphs_on_different_threads(ctx):
    """bench.py drives one graph per camera stream from its own host thread, all rebinding frames of ONE shared ring
    every frame (vxSetParameterByIndex, feature_tracker.cpp:216): reference counts must survive that"""
    import threading
    shared = [L.vxCreateImage(ctx, 64, 48, ovx.VX_DF_IMAGE_RGBX) for _ in range(2)]
    nodes = []
    for _ in range(8):
        g = L.vxCreateGraph(ctx)
        out = L.vxCreateImage(ctx, 64, 48, ovx.VX_DF_IMAGE_U8)
        n = L.vxColorConvertNode(g, shared[0], out)
        assert L.vxGetStatus(n) == ovx.VX_SUCCESS and L.vxVerifyGraph(g) == ovx.VX_SUCCESS
        nodes.append(n)

    def work(n):
        for k in range(20000):
            L.vxSetParameterByIndex(n, 0, shared[k & 1])

    ths = [threading.Thread(target=work, args=(n,)) for n in nodes]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    # every node ends bound to shared[1] (k = 19999): the creator's reference + one per node; shared[0] only the creator's
    assert ovx.query("vxQueryReference", shared[1], ovx.VX_REFERENCE_COUNT, C.c_uint32) == 1 + len(nodes)
    assert ovx.query("vxQueryReference", shared[0], ovx.VX_REFERENCE_COUNT, C.c_uint32) == 1


def test_openvx_1_0_kernel_api(ctx):
    """vxAddKernel with per-parameter validators reaching the bound objects through vx_parameter, as the SFM sample's
    scaleCloud kernel does (sfm/SfM.cpp:1204-1400)"""
    calls = []

    def in_validate(node, index):
        p = L.vxGetParameterByIndex(node, index)
        assert L.vxGetStatus(p) == ovx.VX_SUCCESS
        ref = C.c_void_p()
        assert L.vxQueryParameter(p, ovx.VX_PARAMETER_REF, C.byref(ref), 8) == ovx.VX_SUCCESS
        assert ovx.query("vxQueryParameter", p, ovx.VX_PARAMETER_INDEX, C.c_uint32) == index
        assert ovx.query("vxQueryParameter", p, ovx.VX_PARAMETER_DIRECTION, C.c_int32) == ovx.VX_INPUT
        rows = ovx.query("vxQueryMatrix", ref.value, ovx.VX_MATRIX_ROWS, C.c_size_t)
        cols = ovx.query("vxQueryMatrix", ref.value, ovx.VX_MATRIX_COLUMNS, C.c_size_t)
        calls.append(("in", index, rows, cols))
        assert ovx.release("Matrix", ref.value) == ovx.VX_SUCCESS          # VX_PARAMETER_REF hands out a reference
        assert L.vxReleaseParameter(C.byref(C.c_void_p(p))) == ovx.VX_SUCCESS
        return ovx.VX_SUCCESS if (rows, cols) == (1, 3) else ovx.VX_ERROR_INVALID_PARAMETERS

    def out_validate(node, index, meta):
        calls.append(("out", index))
        t, r, c = C.c_int32(ovx.VX_TYPE_FLOAT32), C.c_size_t(1), C.c_size_t(3)
        L.vxSetMetaFormatAttribute(meta, ovx.VX_MATRIX_TYPE, C.byref(t), 4)
        L.vxSetMetaFormatAttribute(meta, ovx.VX_MATRIX_ROWS, C.byref(r), 8)
        L.vxSetMetaFormatAttribute(meta, ovx.VX_MATRIX_COLUMNS, C.byref(c), 8)
        return ovx.VX_SUCCESS

    f, iv, ov = ovx.KERNEL_F(lambda n, p, c: 0), ovx.INPUT_VALIDATE_F(in_validate), ovx.OUTPUT_VALIDATE_F(out_validate)
    none_f = C.cast(None, ovx.KERNEL_F)
    _KEEP.extend([f, iv, ov])
    kid = C.c_int32()
    assert L.vxAllocateUserKernelId(ctx, C.byref(kid)) == ovx.VX_SUCCESS
    k = L.vxAddKernel(ctx, b"example.test.scale", kid.value, f, 2, iv, ov, none_f, none_f)
    assert L.vxGetStatus(k) == ovx.VX_SUCCESS
    assert L.vxAddParameterToKernel(k, 0, ovx.VX_INPUT, ovx.VX_TYPE_MATRIX, ovx.VX_PARAMETER_STATE_REQUIRED) == 0
    assert L.vxAddParameterToKernel(k, 1, ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX, ovx.VX_PARAMETER_STATE_REQUIRED) == 0
    assert L.vxFinalizeKernel(k) == ovx.VX_SUCCESS
    g = L.vxCreateGraph(ctx)
    n = L.vxCreateGenericNode(g, k)
    a, b = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 1), L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 1)
    assert L.vxSetParameterByIndex(n, 0, a) == 0 and L.vxSetParameterByIndex(n, 1, b) == 0
    refs_before = ovx.query("vxQueryReference", a, ovx.VX_REFERENCE_COUNT, C.c_uint32)
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    assert calls == [("in", 0, 1, 3), ("out", 1)]
    assert ovx.query("vxQueryReference", a, ovx.VX_REFERENCE_COUNT, C.c_uint32) == refs_before
    assert ovx.query("vxQueryReference", n, ovx.VX_REFERENCE_COUNT, C.c_uint32) == 2       # the parameter object let go of the node
    assert L.vxSetParameterByIndex(n, 0, L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)) == 0
    assert L.vxVerifyGraph(g) == ovx.VX_ERROR_INVALID_PARAMETERS
    assert L.vxGetStatus(L.vxGetParameterByIndex(n, 2)) == ovx.VX_ERROR_INVALID_PARAMETERS


def _sfm_graph(ctx, virtual=True, second_pyramid=False, w=320, h=240, unsized=False):
    """front-end nodes of nvx::SfM::createMainGraph (VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp:970-1003,1077-1078) and of the
    stabilizer (video_stabilizer/stabilizer.cpp:237-260): a (virtual) grey image that the pyramid is built FROM"""
    frame = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    pex = L.vxCreatePyramid(ctx, 4, 0.5, w, h, ovx.VX_DF_IMAGE_U8)
    pyr_delay = L.vxCreateDelay(ctx, pex, 2)
    aex = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 500)
    pts_delay = L.vxCreateDelay(ctx, aex, 2)
    kp = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 500)
    g = L.vxCreateGraph(ctx)
    gw, gh = (0, 0) if unsized else (w, h)
    gray = L.vxCreateVirtualImage(g, gw, gh, ovx.VX_DF_IMAGE_U8) if virtual else L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_U8)
    pyr0 = L.vxGetReferenceFromDelay(pyr_delay, 0)
    eps, it, est = ovx.scalar(ctx, ovx.VX_TYPE_FLOAT32, 0.01), ovx.scalar(ctx, ovx.VX_TYPE_UINT32, 5), ovx.scalar(ctx, ovx.VX_TYPE_BOOL, 0)
    nodes = [L.nvxHarrisTrackNode(g, gray, L.vxGetReferenceFromDelay(pts_delay, 0), None, kp, 0.01, 5.0, 18, None),
             L.vxOpticalFlowPyrLKNode(g, L.vxGetReferenceFromDelay(pyr_delay, -1), pyr0, L.vxGetReferenceFromDelay(pts_delay, -1),
                                      L.vxGetReferenceFromDelay(pts_delay, -1), kp, ovx.VX_TERM_CRITERIA_BOTH, eps, it, est, 10),
             L.vxGaussianPyramidNode(g, gray, pyr0),
             L.vxColorConvertNode(g, frame, gray)]
    if second_pyramid:
        nodes.append(L.vxGaussianPyramidNode(g, gray, pex))
    for n in nodes:
        assert L.vxGetStatus(n) == ovx.VX_SUCCESS
    return g, gray


def test_virtual_grey_image_becomes_level_0_of_the_pyramid_it_feeds(ctx, monkeypatch):
    """A virtual image is one the application never accesses (OpenVX): the runtime gives the stabilizer's and the SFM sample's
    virtual grey frame no memory of its own but makes it a view of level 0 of the pyramid built from it, so their graphs run
    with the feature tracker's fusions.  A real image, or a grey image feeding two pyramids, keeps its own memory."""
    order = [ovx.K_COLOR_CONVERT, ovx.K_GAUSSIAN_PYRAMID, ovx.K_PYRLK, ovx.K_HARRIS_TRACK]
    for unsized in (False, True):           # the stabilizer's virtual image is created 0 x 0 (stabilizer.cpp:237)
        g, gray = _sfm_graph(ctx, unsized=unsized)
        assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS and _order(g) == order
        assert ovx.graph_fusions(g) == {"color_convert_in_pyramid": 1, "cells_in_pyrlk": 1, "virtual_gray_is_level0": 1, "merge_in_homography": 0}
        assert ovx.query("vxQueryImage", gray, ovx.VX_IMAGE_WIDTH, C.c_uint32) == 320
        # a second verification starts over and arrives at the same graph
        assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS and _order(g) == order
        assert ovx.graph_fusions(g)["virtual_gray_is_level0"] == 1
    g, _ = _sfm_graph(ctx, virtual=False)
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS and _order(g) == order
    assert ovx.graph_fusions(g) == {"color_convert_in_pyramid": 0, "cells_in_pyrlk": 1, "virtual_gray_is_level0": 0, "merge_in_homography": 0}
    g, _ = _sfm_graph(ctx, second_pyramid=True)
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS
    assert ovx.graph_fusions(g)["virtual_gray_is_level0"] == 0 and ovx.graph_fusions(g)["color_convert_in_pyramid"] == 0
    monkeypatch.setenv("VISUALSLAM_AMD_NO_FUSION", "1")
    g, _ = _sfm_graph(ctx)
    assert L.vxVerifyGraph(g) == ovx.VX_SUCCESS and _order(g) == order
    assert ovx.graph_fusions(g) == {"color_convert_in_pyramid": 0, "cells_in_pyrlk": 0, "virtual_gray_is_level0": 0, "merge_in_homography": 0}
