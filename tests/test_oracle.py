"""CPU tests of the oracle (oracle/vxoracle.c): behaviour the reference callers pin, independent
numpy restatements of the integer stages, analytic properties, and the frozen golden fixture
(tests/golden/front_end_v2.npz, made by tests/golden/make_golden.py).

PARITY UNPINNED: the reference holds no golden vector for this path (closed libvisionworks); these
tests pin the oracle itself."""
import os
import zlib

import numpy as np
import pytest

import oracle as O
from oracle.pipeline import OracleTracker
from visualslam_amd import synth
from visualslam_amd.thin_tracker import Params

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "front_end_v2.npz"))


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


# ---- facts pinned by the reference callers ---------------------------------------------------
def test_pyramid_level_sizes_follow_float_ceil():
    # SURVEY 8a / feature_tracker_nvxcu.cpp:164-176
    assert O.pyramid_dims(640, 480, 6) == [(640, 480), (320, 240), (160, 120), (80, 60), (40, 30), (20, 15)]
    assert O.pyramid_dims(1920, 1080, 5) == [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68)]
    assert O.pyramid_dims(3840, 2160, 7)[-2:] == [(120, 68), (60, 34)]
    assert O.pyramid_dims(5, 7, 4) == [(5, 7), (3, 4), (2, 2), (1, 1)]
    from visualslam_amd.devmem import pyramid_dims
    for w, h, n in [(641, 479, 8), (1, 1, 3), (1023, 7, 5)]:
        assert pyramid_dims(w, h, n) == O.pyramid_dims(w, h, n)


def test_item_layouts():
    # nvxio/src/OpenGLShaders.hpp:567-597
    assert O.keypointf_dtype.itemsize == 28
    assert [O.keypointf_dtype.fields[n][1] for n in ("x", "y", "strength", "scale", "orientation",
                                                     "tracking_status", "error")] == [0, 4, 8, 12, 16, 20, 24]


# ---- colour convert ---------------------------------------------------------------------------
def test_color_convert_known_values():
    px = np.array([[[255, 255, 255, 0], [255, 0, 0, 9], [0, 255, 0, 9], [0, 0, 255, 9], [0, 0, 0, 255],
                    [10, 20, 30, 40]]], np.uint8)
    got = O.color_convert(px)[0]
    assert got.tolist() == [255, 54, 182, 19, 0, (54 * 10 + 183 * 20 + 19 * 30 + 128) >> 8]
    g = np.arange(256, dtype=np.uint8).reshape(16, 16)
    np.testing.assert_array_equal(O.color_convert(synth.to_rgbx(g)), g)   # weights sum to 256
    rgbx = synth.color_rgbx(97, 31, 3)
    y = 0.2126 * rgbx[..., 0] + 0.7152 * rgbx[..., 1] + 0.0722 * rgbx[..., 2]
    assert np.abs(O.color_convert(rgbx).astype(float) - y).max() <= 1.0   # BT.709 to fixed-point accuracy
    np.testing.assert_array_equal(O.color_convert(rgbx, omp=True), O.color_convert(rgbx))


# ---- Gaussian pyramid --------------------------------------------------------------------------
def _np_gaussian_half(src):
    h, w = src.shape
    dw, dh = (w + 1) // 2, (h + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(src.astype(np.int64), 2, mode="edge")
    hs = sum(k[i] * p[:, i:i + w] for i in range(5))
    vs = sum(k[j] * hs[j:j + h, :] for j in range(5))
    return ((vs[0:2 * dh:2, 0:2 * dw:2] + 128) >> 8).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(64, 48), (65, 33), (5, 7), (1, 1), (2, 9), (131, 35)])
def test_gaussian_half_matches_numpy(w, h):
    rng = np.random.default_rng(w * 100 + h)
    src = rng.integers(0, 256, (h, w), dtype=np.uint8)
    np.testing.assert_array_equal(O.gaussian_half(src), _np_gaussian_half(src))


def test_gaussian_pyramid_properties():
    flat = np.full((37, 53), 201, np.uint8)
    for lvl in O.gaussian_pyramid(flat, 5):
        assert (lvl == 201).all()                       # weights sum to 256, replicate border
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (120, 160), dtype=np.uint8)
    pyr = O.gaussian_pyramid(img, 4)
    assert pyr[0] is not None and np.array_equal(pyr[0], img)   # level 0 is the input itself
    assert [p.shape for p in pyr] == [(120, 160), (60, 80), (30, 40), (15, 20)]
    for a, b in zip(O.gaussian_pyramid(img, 4, omp=True), pyr):
        np.testing.assert_array_equal(a, b)


# ---- Harris ------------------------------------------------------------------------------------
def _np_harris(img, k):
    from scipy.ndimage import correlate
    f = img.astype(np.int64)
    sx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    ix, iy = correlate(f, sx, mode="nearest"), correlate(f, sx.T, mode="nearest")
    box = np.ones((3, 3), np.int64)
    a, b, c = (correlate(v, box, mode="constant") for v in (ix * ix, ix * iy, iy * iy))
    S = np.float32(1.0) / np.float32(144.0)
    A, B, C = (v.astype(np.float32) * S for v in (a, b, c))
    R = (A * C - B * B) - np.float32(k) * (A + C) * (A + C)
    out = np.full(img.shape, -np.inf, np.float32)
    out[2:-2, 2:-2] = R[2:-2, 2:-2]
    return out


def test_harris_response_matches_numpy():
    img = synth.gray_frame(synth.base_frame(96, 64, 5), 0, 5)
    for k in (0.04, 0.01):
        np.testing.assert_array_equal(O.harris_response(img, k), _np_harris(img, k))
    assert np.isneginf(O.harris_response(np.zeros((4, 9), np.uint8), 0.04)).all()   # no 5x5 support


def test_harris_track_checkerboard_and_flat():
    img = np.zeros((72, 72), np.uint8)
    img[:36, :36] = 255
    img[36:, 36:] = 255
    pts, total, st = O.harris_track(img, None, None, 0.04, 100.0, 72, 10, item_type=O.KEYPOINTF)
    assert total == 1 and len(pts) == 1
    assert abs(pts["x"][0] - 35.5) <= 1.0 and abs(pts["y"][0] - 35.5) <= 1.0
    assert pts["tracking_status"][0] == 1 and pts["strength"][0] == st[0] > 100.0
    none, total, _ = O.harris_track(np.full((60, 60), 77, np.uint8), None, None, 0.04, 100.0, 18, 10)
    assert total == 0 and len(none) == 0


def test_harris_track_cells_mask_merge_order():
    w, h, cs = 200, 150, 18
    img = synth.gray_frame(synth.base_frame(w, h, 8), 0, 8)
    pts, total, _ = O.harris_track(img, None, None, 0.04, 100.0, cs, 1000)
    cells = (pts[:, 1] // cs).astype(int) * ((w + cs - 1) // cs) + (pts[:, 0] // cs).astype(int)
    assert total == len(pts) and len(np.unique(cells)) == len(cells)       # one corner per cell
    assert (np.diff(cells) > 0).all()                                       # cell raster order
    mask = np.zeros((h, w), np.uint8)
    mask[:, 100:] = 1
    mp, _, _ = O.harris_track(img, mask, None, 0.04, 100.0, cs, 1000)
    assert len(mp) and (mp[:, 0] >= 100).all()                              # mask != 0 gating
    tracked = np.array([[-1, -1], [50.2, 40.7], [199.0, 149.0], [50.9, 40.1], [200.0, 10.0], [10.5, -0.1]], np.float32)
    out, total, _ = O.harris_track(img, None, tracked, 0.04, 100.0, cs, 1000)
    np.testing.assert_array_equal(out[:3], tracked[[1, 2, 3]])              # survivors first, input order
    new = out[3:]
    occupied = {(int(50.2) // cs, int(40.7) // cs), (199 // cs, 149 // cs)}
    assert not {(int(x) // cs, int(y) // cs) for x, y in new} & occupied    # occupied cells emit nothing
    assert total == len(out)
    cut, total2, _ = O.harris_track(img, None, tracked, 0.04, 100.0, cs, 5)
    assert total2 == total and len(cut) == 5
    np.testing.assert_array_equal(cut, out[:5])                             # truncation keeps the order
    kp = np.zeros(2, O.keypointf_dtype)
    kp["x"], kp["y"], kp["tracking_status"], kp["strength"] = [30, 60], [30, 60], [0, 255], [7, 9]
    ko, _, _ = O.harris_track(img, None, kp, 0.04, 100.0, cs, 1000, item_type=O.KEYPOINTF)
    assert ko[0]["strength"] == 9 and ko[0]["tracking_status"] == 255      # status 0 = lost, dropped
    np.testing.assert_array_equal(O.harris_track(img, None, None, 0.04, 100.0, cs, 1000, omp=True)[0], pts)


# ---- FAST --------------------------------------------------------------------------------------
def _py_fast_strength(img, n):
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3]
    h, w = img.shape
    out = np.full((h, w), -1, np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            ip = int(img[y, x])
            ring = [int(img[y + dy[i], x + dx[i]]) for i in range(16)]
            best = -1
            for t in range(0, 255):   # largest t for which the segment test still passes
                br = [v > ip + t for v in ring]
                dk = [v < ip - t for v in ring]
                ok = any(all(b[(s + i) % 16] for i in range(n)) for b in (br, dk) for s in range(16))
                if not ok:
                    break
                best = t
            out[y, x] = best
    return out


def test_fast_strength_matches_segment_test_definition():
    img = synth.gray_frame(synth.base_frame(24, 20, 13, blobs_per_mpix=60000), 0, 13)
    for n in (9, 12):
        np.testing.assert_array_equal(O.fast_strength(img, n), _py_fast_strength(img, n))


def test_fast_track_corner_of_square():
    img = np.full((40, 40), 20, np.uint8)
    img[10:30, 10:30] = 220
    pts, total, st = O.fast_track(img, None, None, 9, 50, 40, 10, item_type=O.KEYPOINTF)
    assert total == 1
    assert (int(pts["x"][0]), int(pts["y"][0])) in {(10, 10), (29, 10), (10, 29), (29, 29), (11, 11), (28, 11), (11, 28), (28, 28)}
    assert st[0] >= 50
    assert O.fast_track(img, None, None, 9, 250, 40, 10)[1] == 0


# ---- PyrLK -------------------------------------------------------------------------------------
def _pair(w, h, levels, dx, dy, seed=3):
    base = synth.base_frame(w, h, seed)
    f0 = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    f1 = np.clip(np.rint(synth.shift_bilinear(base, dx, dy)), 0, 255).astype(np.uint8)
    return O.gaussian_pyramid(f0, levels), O.gaussian_pyramid(f1, levels)


@pytest.mark.parametrize("win,levels,shift", [(6, 4, (1.37, -0.62)), (10, 4, (-2.5, 3.25)), (21, 5, (5.3, -3.7))])
def test_pyrlk_recovers_translation(win, levels, shift):
    p0, p1 = _pair(320, 240, levels, *shift)
    rng = np.random.default_rng(win)
    pts = np.stack([rng.uniform(40, 280, 300), rng.uniform(40, 200, 300)], 1).astype(np.float32)
    out = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_BOTH, 0.01, 10, win)
    ok = out[:, 0] >= 0
    assert ok.mean() > 0.9
    assert np.median(np.abs(out[ok] - pts[ok] - np.array(shift, np.float32))) < 0.05   # SURVEY 8c analytic check
    alt = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_BOTH, 0.01, 10, win, sum_mode=O.SUM_F64SEQ)
    np.testing.assert_array_equal(alt[:, 0] >= 0, ok)
    assert np.abs(alt[ok] - out[ok]).max() <= 0.01
    np.testing.assert_array_equal(O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_BOTH, 0.01, 10, win, omp=True), out)


def test_pyrlk_contract_count_order_lost_encoding():
    p0, p1 = _pair(160, 120, 3, 0.5, 0.25)
    pts = np.array([[-1, -1], [80, 60], [160.5, 10], [3, 118.9], [80.25, 60.75], [0, 0]], np.float32)
    out = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_BOTH, 0.01, 5, 7)
    assert out.shape == pts.shape                                           # same count, same order
    np.testing.assert_array_equal(out[0], [-1, -1])                          # lost stays lost
    np.testing.assert_array_equal(out[2], [-1, -1])                          # outside the image
    assert (out[[1, 4]] >= 0).all()
    lost = out[:, 0] < 0
    assert (out[lost] == -1).all()                                          # lost POINT2F is exactly (-1,-1)
    kp = np.zeros(3, O.keypointf_dtype)
    kp["x"], kp["y"] = [80, 40, 500], [60, 30, 10]
    kp["strength"], kp["tracking_status"], kp["error"] = [5, 6, 7], [255, 0, 1], 9.0
    ko = O.optical_flow_pyr_lk(p0, p1, kp, None, O.TERM_BOTH, 0.01, 5, 7, item_type=O.KEYPOINTF)
    assert ko["tracking_status"].tolist() == [255, 0, 0]
    np.testing.assert_array_equal(ko["strength"], kp["strength"])
    assert (ko["error"] == 0).all()
    # calcError: mean |I - J| at the final position -- ~0 between identical frames, small after a good track,
    # the same in both summation orders up to rounding, 0 for lost points
    ke = O.optical_flow_pyr_lk(p0, p1, kp, None, O.TERM_BOTH, 0.01, 5, 7, item_type=O.KEYPOINTF, calc_error=True)
    for f in ("x", "y", "tracking_status", "strength"):
        np.testing.assert_array_equal(ke[f], ko[f])
    assert 0 < ke["error"][0] < 6.0 and (ke["error"][1:] == 0).all()
    k64 = O.optical_flow_pyr_lk(p0, p1, kp, None, O.TERM_BOTH, 0.01, 5, 7, item_type=O.KEYPOINTF, calc_error=True, sum_mode=O.SUM_F64SEQ)
    assert abs(float(k64["error"][0]) - float(ke["error"][0])) < 0.02
    ks = O.optical_flow_pyr_lk(p0, p0, kp, None, O.TERM_BOTH, 0.01, 5, 7, item_type=O.KEYPOINTF, calc_error=True)
    assert ks["tracking_status"][0] == 255 and ks["error"][0] < 0.05
    flat = [np.full((30 >> l, 40 >> l), 50, np.uint8) for l in range(2)]
    np.testing.assert_array_equal(O.optical_flow_pyr_lk(flat, flat, np.array([[20, 15]], np.float32), window=5), [[-1, -1]])
    same = O.optical_flow_pyr_lk(p0, p0, pts[[1, 4]], None, O.TERM_BOTH, 0.01, 5, 7)
    assert np.abs(same - pts[[1, 4]]).max() < 1e-3                          # identical frames: no motion


def test_pyrlk_termination_modes_and_estimates():
    p0, p1 = _pair(200, 150, 3, 2.0, -1.5)
    pts = np.array([[100, 75], [60.5, 40.5], [150, 100]], np.float32)
    one = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_ITERATIONS, 0.01, 1, 9)
    many = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_ITERATIONS, 0.01, 30, 9)
    eps = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_EPSILON, 1e-4, 0, 9)
    target = pts + np.array([2.0, -1.5], np.float32)
    assert np.abs(many - target).max() < 0.1 and np.abs(eps - target).max() < 0.1
    assert np.abs(one - target).max() >= np.abs(many - target).max()
    single0, single1 = [p0[0]], [p1[0]]
    est = pts + np.array([2.0, -1.0], np.float32)
    with_est = O.optical_flow_pyr_lk(single0, single1, pts, est, O.TERM_BOTH, 0.01, 10, 9)
    assert np.abs(with_est - target).max() < 0.1                             # estimate seeds the search


# ---- PyrLK: an independent second source --------------------------------------------------------
def _np_pyrlk(p0, p1, pts, win, iters=5, eps=0.01):
    """Pyramidal LK restated independently of oracle/vxoracle.c, in SURVEY.md section 8a's ORIGINAL formulation (the
    one the oracle had before its round-1 policy move): Scharr 3x3 / 32 derivatives of the INTEGER
    previous image, then derivatives and image interpolated bilinearly at the window's sub-pixel sample positions
    (scipy.ndimage.map_coordinates, order 1), float64 sums in raster order, 2x2 solve in float64.  It shares no code,
    no summation order and no resampling order with the oracle; only the frozen policies are the same: window offsets
    -win/2 .. win-1-win/2, replicate border (the image extended by its edge pixels), a flat level is skipped (lost at
    level 0), a window centre more than one window outside a level or a final position outside the image loses the
    point, |eta|^2 < eps^2 or the iteration count stops a level, g <- 2 (g + v)."""
    from scipy.ndimage import correlate, map_coordinates
    half, levels = win // 2, len(p0)
    pad = win + 12
    kx = np.array([[-3, 0, 3], [-10, 0, 10], [-3, 0, 3]], np.float64) / 32.0
    prep = []
    for a, b in zip(p0, p1):
        A = np.pad(a.astype(np.float64), pad, mode="edge")
        B = np.pad(b.astype(np.float64), pad, mode="edge")
        prep.append((A, correlate(A, kx, mode="nearest"), correlate(A, kx.T, mode="nearest"), B))
    oy, ox = np.meshgrid(np.arange(win) - half, np.arange(win) - half, indexing="ij")
    h0, w0 = p0[0].shape
    out = np.full((len(pts), 2), -1.0, np.float32)
    for i, (px, py) in enumerate(pts.astype(np.float64)):
        if not (0 <= px <= w0 - 1 and 0 <= py <= h0 - 1):
            continue
        g, v, ok = np.zeros(2), np.zeros(2), True
        for L in range(levels - 1, -1, -1):
            A, Ax, Ay, B = prep[L]
            lh, lw = p0[L].shape
            u = np.array([px, py]) / (1 << L)
            v = np.zeros(2)
            at = [oy + u[1] + pad, ox + u[0] + pad]
            I, Ix, Iy = (map_coordinates(X, at, order=1, mode="nearest") for X in (A, Ax, Ay))
            gxx, gxy, gyy = (Ix * Ix).sum(), (Ix * Iy).sum(), (Iy * Iy).sum()
            det = gxx * gyy - gxy * gxy
            min_eig = ((gxx + gyy) - np.sqrt((gxx - gyy) ** 2 + 4 * gxy * gxy)) / (2 * win * win)
            if not (min_eig >= 1e-4) or not (det >= np.finfo(np.float32).eps):
                if L == 0:
                    ok = False
                g = 2 * g
                continue
            for _ in range(iters):
                c = u + g + v
                if not (-win < c[0] < lw + win and -win < c[1] < lh + win):
                    ok = False
                    break
                J = map_coordinates(B, [oy + c[1] + pad, ox + c[0] + pad], order=1, mode="nearest")
                d = I - J
                b1, b2 = (d * Ix).sum(), (d * Iy).sum()
                eta = np.array([gyy * b1 - gxy * b2, gxx * b2 - gxy * b1]) / det
                v = v + eta
                if eta @ eta < eps * eps:
                    break
            if not ok:
                break
            if L > 0:
                g = 2 * (g + v)
        if ok:
            r = np.array([px, py]) + g + v
            if 0 <= r[0] <= w0 - 1 and 0 <= r[1] <= h0 - 1:
                out[i] = r
    return out


@pytest.mark.parametrize("win", [6, 10, 21])
def test_pyrlk_against_independent_numpy_restatement_on_the_golden_pair(win):
    """Second source for the one float-heavy kernel (VERDICT r1 'Next round' 1b).  The oracle's PyrLK changed its
    formulation once in round 1 (template resampled, then differentiated; y-first lerps); this keeps the original
    formulation alive, independently implemented, as the guard: positions within 0.01 px, identical lost flags."""
    g0 = O.color_convert(GOLD["lena_rgbx"])
    p0, p1 = O.gaussian_pyramid(g0, 3), O.gaussian_pyramid(GOLD["lk_g1"], 3)
    pts = GOLD["lk_pts"]
    want = _np_pyrlk(p0, p1, pts, win)
    for mode in (O.SUM_WAVE64, O.SUM_F64SEQ):
        got = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_BOTH, 0.01, 5, win, sum_mode=mode)
        np.testing.assert_array_equal(got[:, 0] >= 0, want[:, 0] >= 0)
        ok = want[:, 0] >= 0
        assert ok.sum() > 100
        d = np.abs(got[ok] - want[ok]).max(axis=1)
        assert d.max() <= 0.01, (win, mode, float(d.max()))


def test_pyrlk_against_independent_numpy_restatement_larger_motion():
    """the same guard away from the fixture: 5 levels, 21x21, a (5.3, -3.7) px shift, points anywhere in the frame
    (windows that poke out of the upper levels included)"""
    p0, p1 = _pair(320, 240, 5, 5.3, -3.7)
    rng = np.random.default_rng(21)
    pts = np.stack([rng.uniform(0, 319, 250), rng.uniform(0, 239, 250)], 1).astype(np.float32)
    want = _np_pyrlk(p0, p1, pts, 21)
    got = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_BOTH, 0.01, 5, 21)
    np.testing.assert_array_equal(got[:, 0] >= 0, want[:, 0] >= 0)
    ok = want[:, 0] >= 0
    assert ok.sum() > 200 and np.abs(got[ok] - want[ok]).max() <= 0.01


# ---- frozen golden fixture ---------------------------------------------------------------------
@pytest.mark.parametrize("name", ["lena", "baboon"])
def test_golden_front_end(name):
    rgbx = GOLD[f"{name}_rgbx"]
    g = O.color_convert(rgbx)
    assert crc(g) == GOLD[f"{name}_gray_crc"]
    pyr = O.gaussian_pyramid(g, 4)
    assert [crc(l) for l in pyr] == GOLD[f"{name}_pyr_crc"].tolist()
    np.testing.assert_array_equal(pyr[3], GOLD[f"{name}_pyr_l3"])
    mask = np.zeros(g.shape, np.uint8)
    mask[30:90, 40:120] = 255
    for tag, m in (("all", None), ("center", mask)):
        hp, ht, _ = O.harris_track(g, m, None, 0.04, 100.0, 18, 200, item_type=O.KEYPOINTF)
        want = GOLD[f"{name}_harris_{tag}"]
        assert ht == GOLD[f"{name}_harris_{tag}_total"]
        assert hp.tobytes() == want.tobytes()
    fp, ft, _ = O.fast_track(g, None, None, 9, 25, 18, 200, item_type=O.KEYPOINTF)
    assert ft == GOLD[f"{name}_fast9_25_total"] and fp.tobytes() == GOLD[f"{name}_fast9_25"].tobytes()


def test_golden_lk():
    g0 = O.color_convert(GOLD["lena_rgbx"])
    p0, p1 = O.gaussian_pyramid(g0, 3), O.gaussian_pyramid(GOLD["lk_g1"], 3)
    pts = GOLD["lk_pts"]
    for win in (6, 10, 21):
        out = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_BOTH, 0.01, 5, win)
        want = GOLD[f"lk_win{win}"]
        np.testing.assert_array_equal(out[:, 0] >= 0, want[:, 0] >= 0)
        assert np.abs(out - want).max() <= 0.01 and out.tobytes() == want.tobytes()
        ok = want[:-4, 0] >= 0
        assert np.median(np.abs(want[:-4][ok] - pts[:-4][ok] - np.array([1.37, -0.62]))) < 0.1


def test_golden_sequence():
    base = synth.base_frame(320, 240, 99)
    trk = OracleTracker(Params(pyr_levels=4, lk_win_size=10, array_capacity=400))
    trk.init(synth.to_rgbx(synth.gray_frame(base, 0, 99)))
    rec = []
    for t in range(1, 12):
        old, cur = trk.track(synth.to_rgbx(synth.gray_frame(base, t, 99)))
        rec.append((len(old), int((cur[:, 0] >= 0).sum()), len(trk.prev_pts), crc(cur), crc(trk.prev_pts)))
    np.testing.assert_array_equal(np.array(rec, np.int64), GOLD["seq_record"])
