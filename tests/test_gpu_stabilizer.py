"""GPU tests of what the video stabilizer adds to the tracker graph, through the OpenVX C API:
vxWarpPerspectiveNode / vxuWarpPerspective against the oracle (bit for bit), uniform and ROI images, user kernels
(host callbacks) running between device nodes of one graph, and the reference's own demo -- main_video_stabilizer.cpp,
stabilizer.cpp and its three Eigen user-kernel nodes compiled unmodified into oracle/_ref/nvx_demo_video_stabilizer --
end to end."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O
from test_oracle_homography import scene
from visualslam_amd import ovx, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ovx.lib()
_KEEP = []


@pytest.fixture
def ctx(gpu):
    c = L.vxCreateContext()
    assert L.vxGetStatus(c) == ovx.VX_SUCCESS
    yield c
    h = C.c_void_p(c)
    assert L.vxReleaseContext(C.byref(h)) == ovx.VX_SUCCESS


def _matrix(ctx, M):
    """M: the 3x3 map (x0,y0,z0) = M (x,y,1); stored as OpenVX keeps it, column by column"""
    m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    data = np.ascontiguousarray(np.asarray(M, np.float32).T)
    ovx.check(L.vxCopyMatrix(m, data.ctypes.data, ovx.VX_WRITE_ONLY, ovx.VX_MEMORY_TYPE_HOST))
    return m, data


def _maps(w, h):
    cx, cy, a, s = (w - 1) / 2, (h - 1) / 2, np.deg2rad(5.0), 1.05
    rot = np.array([[s * np.cos(a), -s * np.sin(a), 0], [s * np.sin(a), s * np.cos(a), 0], [0, 0, 1]])
    T = lambda x, y: np.array([[1, 0, x], [0, 1, y], [0, 0, 1.0]])
    return {
        "identity": np.eye(3),
        "shift": T(3.25, -1.5),
        "rotate": T(cx, cy) @ rot @ T(-cx, -cy),
        "perspective": np.array([[1.02, 0.03, -4.0], [-0.02, 0.97, 6.5], [1.5e-4, -2.0e-4, 1.0]]),
        "horizon": np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, -1.0 / (h / 2), 1.0]]),   # z0 = 0 along y = h/2
        "far away": T(10 * w, 0),
    }


@pytest.mark.parametrize("fmt", ["u8", "rgbx"])
@pytest.mark.parametrize("interp", ["bilinear", "nearest"])
def test_warp_perspective_matches_oracle(ctx, fmt, interp):
    w, h = 331, 203                                                    # odd sizes: ragged tiles
    rng = np.random.default_rng(5)
    img_np = rng.integers(0, 256, (h, w) if fmt == "u8" else (h, w, 4), dtype=np.uint8)
    vfmt = ovx.VX_DF_IMAGE_U8 if fmt == "u8" else ovx.VX_DF_IMAGE_RGBX
    src = L.vxCreateImage(ctx, w, h, vfmt)
    ovx.write_image(src, img_np)
    vint = ovx.VX_INTERPOLATION_BILINEAR if interp == "bilinear" else ovx.VX_INTERPOLATION_NEAREST_NEIGHBOR
    for name, M in _maps(w, h).items():
        for ow, oh in ((w, h), (200, 260)):
            dst = L.vxCreateImage(ctx, ow, oh, vfmt)
            m, data = _matrix(ctx, M)
            ovx.check(L.vxuWarpPerspective(ctx, src, m, vint, dst), "vxuWarpPerspective")
            want = O.warp_perspective(img_np, data, (oh, ow), bilinear=interp == "bilinear")
            got = ovx.read_image(dst)
            assert got.tobytes() == want.tobytes(), (name, ow, oh, int((got != want).sum()))
            ovx.release("Image", dst)
            ovx.release("Matrix", m)
    # sanity of the convention (truncate_transform_node.cpp:163-165): a shift by whole pixels moves the picture
    m, data = _matrix(ctx, [[1, 0, 7], [0, 1, 2], [0, 0, 1]])
    dst = L.vxCreateImage(ctx, w, h, vfmt)
    ovx.check(L.vxuWarpPerspective(ctx, src, m, vint, dst))
    got = ovx.read_image(dst)
    np.testing.assert_array_equal(got[:h - 2, :w - 7], img_np[2:, 7:])
    assert not got[h - 2:].any() and not got[:, w - 7:].any()          # outside the input: the (zero) border


def test_warp_node_border_value_and_roi_views(ctx):
    w, h = 256, 128
    rng = np.random.default_rng(6)
    big = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    parent = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    ovx.write_image(parent, big)
    view = L.vxCreateImageFromROI(parent, C.byref(ovx.vx_rectangle_t(37, 11, 37 + 150, 11 + 90)))
    assert L.vxGetStatus(view) == ovx.VX_SUCCESS
    np.testing.assert_array_equal(ovx.read_image(view), big[11:101, 37:187])
    out = L.vxCreateImage(ctx, 150, 90, ovx.VX_DF_IMAGE_RGBX)
    M = np.array([[0.9, 0.1, 5.0], [-0.1, 0.95, 3.0], [1e-4, 2e-4, 1.0]])
    m, data = _matrix(ctx, M)
    g = L.vxCreateGraph(ctx)
    n = L.vxWarpPerspectiveNode(g, view, m, ovx.VX_INTERPOLATION_BILINEAR, out)
    bd = ovx.vx_border_t(ovx.VX_BORDER_CONSTANT, (C.c_uint8 * 4)(9, 80, 7, 255))
    ovx.check(L.vxSetNodeAttribute(n, ovx.VX_NODE_BORDER, C.byref(bd), C.sizeof(bd)))
    ovx.check(L.vxProcessGraph(g), "vxProcessGraph")
    want = O.warp_perspective(big[11:101, 37:187], data, (90, 150), border=[9, 80, 7, 255])
    got = ovx.read_image(out)
    assert (got == np.array([9, 80, 7, 255], np.uint8)).all(axis=2).any()       # some pixels do fall outside
    assert got.tobytes() == want.tobytes()
    # U8 view at an odd offset (unaligned rows), written INTO by a node: only the view's pixels change
    gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
    p8 = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_U8)
    ovx.write_image(p8, gray)
    v8 = L.vxCreateImageFromROI(p8, C.byref(ovx.vx_rectangle_t(3, 5, 3 + 101, 5 + 67)))
    src8 = L.vxCreateImage(ctx, 101, 67, ovx.VX_DF_IMAGE_U8)
    s_np = rng.integers(0, 256, (67, 101), dtype=np.uint8)
    ovx.write_image(src8, s_np)
    m2, d2 = _matrix(ctx, [[1, 0, 0.5], [0, 1, 0.25], [0, 0, 1]])
    ovx.check(L.vxuWarpPerspective(ctx, src8, m2, ovx.VX_INTERPOLATION_BILINEAR, v8))
    want8 = gray.copy()
    want8[5:72, 3:104] = O.warp_perspective(s_np, d2)
    np.testing.assert_array_equal(ovx.read_image(p8), want8)


def test_uniform_image_and_side_by_side_views(ctx):
    """stabilizer.cpp:399-411 (black uniform image copied into delay slots), main_video_stabilizer.cpp:225-266 (two
    views of one display image filled by nvxuCopyImage)"""
    w, h = 96, 40
    px = ovx.vx_pixel_value_t()
    px.RGBX[:] = [10, 20, 30, 40]
    u = L.vxCreateUniformImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX, C.byref(px))
    assert (ovx.read_image(u) == np.array([10, 20, 30, 40], np.uint8)).all()
    px8 = ovx.vx_pixel_value_t()
    px8.U8 = 77
    u8 = L.vxCreateUniformImage(ctx, w, h, ovx.VX_DF_IMAGE_U8, C.byref(px8))
    assert (ovx.read_image(u8) == 77).all()
    demo = L.vxCreateImage(ctx, 2 * w, h, ovx.VX_DF_IMAGE_RGBX)
    left = L.vxCreateImageFromROI(demo, C.byref(ovx.vx_rectangle_t(0, 0, w, h)))
    right = L.vxCreateImageFromROI(demo, C.byref(ovx.vx_rectangle_t(w, 0, 2 * w, h)))
    pic = np.random.default_rng(2).integers(0, 256, (h, w, 4), dtype=np.uint8)
    a = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    ovx.write_image(a, pic)
    ovx.check(L.nvxuCopyImage(ctx, a, right))
    ovx.check(L.nvxuCopyImage(ctx, u, left))
    got = ovx.read_image(demo)
    np.testing.assert_array_equal(got[:, w:], pic)
    assert (got[:, :w] == np.array([10, 20, 30, 40], np.uint8)).all()
    # a delay of images whose exemplar is the frame (stabilizer.cpp:447): slots are fresh images of that shape
    d = L.vxCreateDelay(ctx, a, 3)
    ovx.check(L.nvxuCopyImage(ctx, a, L.vxGetReferenceFromDelay(d, 0)))
    ovx.check(L.vxAgeDelay(d))
    np.testing.assert_array_equal(ovx.read_image(L.vxGetReferenceFromDelay(d, -1)), pic)
    assert not ovx.read_image(L.vxGetReferenceFromDelay(d, 0)).any()


def test_user_kernel_runs_between_device_nodes(ctx):
    """One graph: nvxFindHomographyNode (device) -> a user kernel (host: reads the matrix and the inlier mask through
    the ordinary accessors, writes another matrix) -> vxWarpPerspectiveNode (device, uses that matrix).  What the
    callback sees and what the last node produces must be what the oracle computes for the same data."""
    src, dst, _ = scene(n=800, noise=0.3, outliers=0.3, seed=11)
    a = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 1000)
    b = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 1000)
    ovx.add_items(a, src)
    ovx.add_items(b, dst)
    w, h = 160, 120
    pic = np.random.default_rng(4).integers(0, 256, (h, w, 4), dtype=np.uint8)
    img = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    out = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    ovx.write_image(img, pic)
    seen = {}

    def func(node, params, num):
        Hm = np.zeros((3, 3), np.float32)
        st = L.vxCopyMatrix(params[0], Hm.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST)
        n_mask = ovx.num_items(params[1])
        seen["H"] = Hm.copy()
        seen["mask"] = ovx.read_array(params[1], np.uint8)
        seen["n"] = n_mask
        # something a host kernel would do: a damped version of the estimated motion, in the stored layout
        M = (0.5 * (Hm.T.astype(np.float64) / 400.0 + np.eye(3))).astype(np.float32)
        M[2, 2] = 1.0
        seen["M"] = np.ascontiguousarray(M.T)
        st |= L.vxCopyMatrix(params[2], seen["M"].ctypes.data, ovx.VX_WRITE_ONLY, ovx.VX_MEMORY_TYPE_HOST)
        return st

    def validate(node, params, num, metas):
        t, r, c = C.c_int32(ovx.VX_TYPE_FLOAT32), C.c_size_t(3), C.c_size_t(3)
        for attr, v, sz in ((ovx.VX_MATRIX_TYPE, t, 4), (ovx.VX_MATRIX_ROWS, r, 8), (ovx.VX_MATRIX_COLUMNS, c, 8)):
            L.vxSetMetaFormatAttribute(metas[2], attr, C.byref(v), sz)
        return ovx.VX_SUCCESS

    kid = C.c_int32()
    ovx.check(L.vxAllocateUserKernelId(ctx, C.byref(kid)))
    f, v, none_f = ovx.KERNEL_F(func), ovx.VALIDATE_F(validate), C.cast(None, ovx.KERNEL_F)
    _KEEP.extend([f, v])
    k = L.vxAddUserKernel(ctx, b"example.test.damp", kid.value, f, 3, v, none_f, none_f)
    for i, (d, t) in enumerate([(ovx.VX_INPUT, ovx.VX_TYPE_MATRIX), (ovx.VX_INPUT, ovx.VX_TYPE_ARRAY), (ovx.VX_OUTPUT, ovx.VX_TYPE_MATRIX)]):
        ovx.check(L.vxAddParameterToKernel(k, i, d, t, ovx.VX_PARAMETER_STATE_REQUIRED))
    ovx.check(L.vxFinalizeKernel(k))

    g = L.vxCreateGraph(ctx)
    Hmat = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    M2 = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
    mask = L.vxCreateVirtualArray(g, ovx.VX_TYPE_UINT8, 1000)
    # created last-first: the order has to come from the data
    nw = L.vxWarpPerspectiveNode(g, img, M2, ovx.VX_INTERPOLATION_BILINEAR, out)
    nu = L.vxCreateGenericNode(g, k)
    for i, p in enumerate((Hmat, mask, M2)):
        ovx.check(L.vxSetParameterByIndex(nu, i, p))
    nh = L.nvxFindHomographyNode(g, a, b, Hmat, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, mask)
    for n in (nw, nu, nh):
        assert L.vxGetStatus(n) == ovx.VX_SUCCESS
    ovx.check(L.vxDirective(ctx, ovx.VX_DIRECTIVE_ENABLE_PERFORMANCE))
    for rep in range(2):                                               # twice: the second run must not see stale mirrors
        seen.clear()
        ovx.check(L.vxProcessGraph(g), "vxProcessGraph")
        Hw, mw, cw = O.find_homography(src, dst)
        assert seen["n"] == len(src) and seen["mask"].tobytes() == mw.tobytes()
        assert seen["H"].tobytes() == Hw.tobytes()
        want = O.warp_perspective(pic, seen["M"])
        assert ovx.read_image(out).tobytes() == want.tobytes()
        back = np.zeros((3, 3), np.float32)
        ovx.check(L.vxCopyMatrix(M2, back.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST))
        assert back.tobytes() == seen["M"].tobytes()                   # the host-written matrix survives the run
        ovx.check(L.vxCopyMatrix(Hmat, back.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST))
        assert back.tobytes() == Hw.tobytes()
    for n in (nh, nu, nw):
        perf = ovx.query("vxQueryNode", n, ovx.VX_NODE_PERFORMANCE, ovx.vx_perf_t)
        assert perf.num == 2 and perf.sum > 0
    # a failing host kernel fails the graph with its status
    seen.clear()
    f2 = ovx.KERNEL_F(lambda node, params, num: ovx.VX_ERROR_INVALID_VALUE)
    _KEEP.append(f2)
    k2 = L.vxAddUserKernel(ctx, b"example.test.fail", kid.value + 1, f2, 1, C.cast(None, ovx.VALIDATE_F), none_f, none_f)
    ovx.check(L.vxAddParameterToKernel(k2, 0, ovx.VX_INPUT, ovx.VX_TYPE_MATRIX, ovx.VX_PARAMETER_STATE_REQUIRED))
    ovx.check(L.vxFinalizeKernel(k2))
    g2 = L.vxCreateGraph(ctx)
    n2 = L.vxCreateGenericNode(g2, k2)
    ovx.check(L.vxSetParameterByIndex(n2, 0, Hmat))
    assert L.vxProcessGraph(g2) == ovx.VX_ERROR_INVALID_VALUE


def _shift_estimate(a, b):
    """translation between two gray images by phase correlation (sub-pixel by the peak's centroid)"""
    fa, fb = np.fft.rfft2(a), np.fft.rfft2(b)
    r = fa * np.conj(fb)
    r /= np.abs(r) + 1e-9
    c = np.fft.irfft2(r, s=a.shape)
    py, px = np.unravel_index(np.argmax(c), c.shape)
    ys = (np.arange(-2, 3) + py) % c.shape[0]
    xs = (np.arange(-2, 3) + px) % c.shape[1]
    win = np.maximum(c[np.ix_(ys, xs)], 0)
    dy = (win.sum(1) * np.arange(-2, 3)).sum() / win.sum() + py
    dx = (win.sum(0) * np.arange(-2, 3)).sum() / win.sum() + px
    if dy > c.shape[0] / 2: dy -= c.shape[0]
    if dx > c.shape[1] / 2: dx -= c.shape[1]
    return dx, dy


def test_reference_video_stabilizer_demo_compiled_unmodified_runs(gpu, tmp_path):
    """oracle/_ref/nvx_demo_video_stabilizer: the reference's demo -- its main, nvx::VideoStabilizer (stabilizer.cpp:
    ColorConvert, CopyImage, GaussianPyramid, PyrLK, FindHomography, three Eigen user kernels, WarpPerspective,
    HarrisTrack in ONE graph with four delays) -- running on libvisualslam_amd.  The input jitters by random whole-pixel
    translations; what the demo displays must be [the input, delayed | the stabilized stream]: the left half bit for bit,
    the right half by its residual frame-to-frame motion."""
    exe = os.path.join(ROOT, "oracle", "_ref", "nvx_demo_video_stabilizer")
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref/nvx_demo_video_stabilizer is missing: it is built by __graft_entry__.build() where /root/reference exists and travels to the GPU box with the tree -- the boundary test must not vanish silently")
    (tmp_path / "bin").mkdir()
    (tmp_path / "data").mkdir()
    run_exe = tmp_path / "bin" / "nvx_demo_video_stabilizer"
    shutil.copy(exe, run_exe)
    w, h, n, smooth = 640, 480, 40, 5
    rng = np.random.default_rng(21)
    base = synth.base_frame(w + 32, h + 32, 77)
    jit = rng.integers(-4, 5, (n, 2))
    frames = [synth.to_rgbx(np.ascontiguousarray(base[16 + dy:16 + dy + h, 16 + dx:16 + dx + w])) for dx, dy in jit]
    fpath, ipath = tmp_path / "frames.rgbx", tmp_path / "shown.bin"
    with open(fpath, "wb") as f:
        for fr in frames:
            f.write(fr.tobytes())
    shown = n - 1
    env = dict(os.environ, VSA_HEADLESS_FRAMES=str(shown), VSA_HEADLESS_IMAGES=str(ipath),
               LD_LIBRARY_PATH=os.path.join(ROOT, "visualslam_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    cmd = [str(run_exe), f"--source=rawrgbx://{fpath}?width={w}&height={h}", "-n", str(smooth), "--crop=0.05", "--nvxio_fps_limit=1000"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=170, env=env)
    if res.returncode != 0 and "nvxio_fps_limit" in (res.stdout + res.stderr):
        res = subprocess.run(cmd[:-1], capture_output=True, text=True, timeout=170, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for line in ("Graph Time", "Find Homography time", "Homography Filter time", "Matrices Smoothing time",
                 "Truncate Stab Transform time", "Warp Perspective time", "Feature Track time"):
        assert line in res.stdout, line                                   # stabilizer.cpp:319-355
    raw = np.fromfile(ipath, np.uint8)
    rec = 8 + 2 * w * h * 4
    assert raw.size == shown * rec
    imgs = []
    for i in range(shown):
        hdr = raw[i * rec:i * rec + 8].view(np.uint32)
        assert tuple(hdr) == (2 * w, h)
        imgs.append(raw[i * rec + 8:(i + 1) * rec].reshape(h, 2 * w, 4))
    # the left half is the input delayed by smooth + 1 frames (main_video_stabilizer.cpp:190-197,262-266), black before
    lag = smooth + 1
    for i, im in enumerate(imgs):
        t = i + 1 - lag                     # display i follows process(frame i)... the delay was aged once more since
        left = im[:, :w]
        if t < 0:
            assert not left[..., :3].any()
        else:
            np.testing.assert_array_equal(left, frames[t])
    # the right half is the stabilized stream: far less frame-to-frame motion than the input has
    crop = (slice(80, h - 80), slice(80, w - 80))
    gray = lambda x: x[..., 1].astype(np.float64)
    first = 2 * lag + 2                      # both the smoothing window and the frame delay are full from here on
    raw_motion, stab_motion = [], []
    for i in range(first, shown - 1):
        raw_motion.append(np.hypot(*_shift_estimate(gray(imgs[i][:, :w])[crop], gray(imgs[i + 1][:, :w])[crop])))
        stab_motion.append(np.hypot(*_shift_estimate(gray(imgs[i][:, w:])[crop], gray(imgs[i + 1][:, w:])[crop])))
    assert np.mean(raw_motion) > 2.0
    assert np.mean(stab_motion) < 0.5 * np.mean(raw_motion), (np.mean(stab_motion), np.mean(raw_motion))
    assert imgs[-1][:, w:, :3].any()


def test_warp_perspective_edge_cases(ctx):
    """tiny and degenerate inputs: a 1x1 source, single rows/columns, a singular and an all-zero matrix (every pixel takes
    the border), infinities in the matrix -- always the oracle's bytes, never a fault"""
    rng = np.random.default_rng(12)
    cases = [((1, 1), (5, 7)), ((1, 9), (9, 1)), ((13, 1), (4, 4)), ((3, 2), (64, 5)), ((2, 300), (300, 2))]
    mats = [np.eye(3), np.zeros((3, 3)), np.array([[1, 2, 3], [2, 4, 6], [0, 0, 0.0]]),
            np.array([[0.5, 0, 0], [0, 0.5, 0], [0, 0, 1.0]]), np.array([[1, 0, np.inf], [0, 1, 0], [0, 0, 1.0]]),
            np.array([[1, 0, 0], [0, 1, 0], [1e30, 1e30, 1.0]]), np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1.0]])]
    for fmt, vfmt in (("u8", ovx.VX_DF_IMAGE_U8), ("rgbx", ovx.VX_DF_IMAGE_RGBX)):
        for (sh, sw), (oh, ow) in cases:
            img_np = rng.integers(0, 256, (sh, sw) if fmt == "u8" else (sh, sw, 4), dtype=np.uint8)
            src = L.vxCreateImage(ctx, sw, sh, vfmt)
            dst = L.vxCreateImage(ctx, ow, oh, vfmt)
            ovx.write_image(src, img_np)
            for M in mats:
                with np.errstate(all="ignore"):
                    m, data = _matrix(ctx, M)
                for vint, bil in ((ovx.VX_INTERPOLATION_BILINEAR, True), (ovx.VX_INTERPOLATION_NEAREST_NEIGHBOR, False)):
                    ovx.check(L.vxuWarpPerspective(ctx, src, m, vint, dst))
                    want = O.warp_perspective(img_np, data, (oh, ow), bilinear=bil)
                    assert ovx.read_image(dst).tobytes() == want.tobytes(), (fmt, sh, sw, oh, ow, M.tolist(), bil)
                ovx.release("Matrix", m)
            ovx.release("Image", src)
            ovx.release("Image", dst)
