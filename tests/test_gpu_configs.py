"""GPU parity at the WORKLOAD of each BASELINE.json config, through the OpenVX C API, against the oracle:

  configs[0]  the feature_tracker graph with the demo's ini values on the 640x480 sequence that stands in for the missing
              bundled video (warps of the reference's lena.jpg / baboon.jpg);
  configs[1]  tests/test_gpu_thin_layer.py::test_pyrlk_full_size_config and the graph below;
  configs[2]  4K / 7 levels / FAST track / 10 000 features as one graph;
  configs[3]  8 independent 1080p camera streams driven concurrently by native host threads on one GPU;
  configs[4]  the 1080p graph with nvxFindHomographyNode on the PyrLK matches, as bench.py --homography builds it.

Bars as everywhere: pyramids bit-exact (implied by identical corners), corner sets identical, LK positions within
0.01 px with the same lost flags -- and in fact bit-identical to the oracle's canonical summation order.
PARITY UNPINNED (oracle/vxoracle.h): the oracle restates closed-source behaviour."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O
from oracle.pipeline import OracleTracker
from visualslam_amd import ovx, synth
from visualslam_amd.thin_tracker import Params
from visualslam_amd.vx_tracker import VxFeatureTracker

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ovx.lib()


@pytest.fixture
def ctx(gpu):
    c = L.vxCreateContext()
    assert L.vxGetStatus(c) == ovx.VX_SUCCESS
    yield c
    h = C.c_void_p(c)
    assert L.vxReleaseContext(C.byref(h)) == ovx.VX_SUCCESS


def _check_frame(trk, ref, frame_np, t):
    old, cur = ref.track(frame_np)
    np.testing.assert_array_equal(trk.prev(), old, err_msg=f"frame {t}: previous features")
    got = trk.curr()
    assert got.shape == cur.shape, f"frame {t}"
    np.testing.assert_array_equal(got[:, 0] >= 0, cur[:, 0] >= 0, err_msg=f"frame {t}: lost flags")
    ok = cur[:, 0] >= 0
    if ok.any():
        assert np.abs(got[ok] - cur[ok]).max() <= 0.01, f"frame {t}"           # north_star tolerance ...
    assert got.tobytes() == cur.tobytes(), f"frame {t}"                         # ... and in fact the same bits
    np.testing.assert_array_equal(trk.merged(), ref.prev_pts, err_msg=f"frame {t}: merged features")
    return old, cur


def test_config0_demo_ini_on_the_lena_baboon_sequence(ctx):
    """data/feature_tracker_demo_config.ini:1-13 (6 levels, 5 iterations, window 6, capacity 2000, cell 18, Harris
    k 0.04 / threshold 2000) on 640x480 frames made from the reference's still images; the scene cut in the middle
    loses most tracks and the detector restores the set (feature_tracker_user_guide.md:32-33)"""
    w, h, n = 640, 480, 12
    src = np.load(os.path.join(ROOT, "tests", "golden", "c1_sources.npz"))
    frames = synth.c1_sequence(src, n, w, h)
    p = Params(pyr_levels=6, lk_num_iters=5, lk_win_size=6, array_capacity=2000, detector_cell_size=18,
               use_harris_detector=True, harris_k=0.04, harris_thresh=2000.0)
    frame = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
    ovx.write_image(frame, frames[0])
    trk = VxFeatureTracker(ctx, p, w, h, frame, None)
    ref = OracleTracker(p)
    ref.init(frames[0])
    np.testing.assert_array_equal(trk.merged(), ref.prev_pts)
    assert 300 < len(ref.prev_pts) <= 2000
    kept = []
    for t in range(1, n):
        ovx.write_image(frame, frames[t])
        trk.track(frame, None)
        old, cur = _check_frame(trk, ref, frames[t], t)
        kept.append(float((cur[:, 0] >= 0).mean()))
    cut = (n + 1) // 2
    assert min(kept[: cut - 1]) > 0.95 and kept[cut - 1] < 0.7 and min(kept[cut:]) > 0.95      # the cut is visible
    # the colour frames exercise the luma weights (R, G and B differ)
    assert (frames[0][..., 0] != frames[0][..., 1]).mean() > 0.5


def _ring(ctx, frames):
    ring = []
    for f in frames:
        img = L.vxCreateImage(ctx, f.shape[1], f.shape[0], ovx.VX_DF_IMAGE_RGBX)
        ovx.write_image(img, f)
        ring.append(img)
    return ring


def test_config3_eight_1080p_streams_driven_concurrently(ctx):
    """configs[3] on one GPU: 8 camera streams at the headline workload (1080p, 5 levels, 4096 points, 21x21, Harris
    cell 18), each with its own graph and HIP stream, driven by 8 native host threads (vsa_drive_streams) that make the
    per-frame calls of feature_tracker.cpp:216-224 concurrently; every stream ends exactly where the oracle ends"""
    w, h, n_streams, n_frames, ring_len = 1920, 1080, 8, 3, 4
    p = Params(pyr_levels=5, lk_num_iters=5, lk_win_size=21, array_capacity=4096, detector_cell_size=18,
               use_harris_detector=True, harris_k=0.04, harris_thresh=100.0)
    base = synth.base_frame(w, h, 1234)
    frames = [synth.to_rgbx(synth.gray_frame(base, t, 1234)) for t in range(ring_len)]
    ring = _ring(ctx, frames)
    phases = [(s * 3) % ring_len for s in range(n_streams)]
    trackers = [VxFeatureTracker(ctx, p, w, h, ring[ph], None) for ph in phases]
    drv = C.CDLL(os.path.join(ROOT, "visualslam_amd", "lib", "libvsa_stream_driver.so"))
    drv.vsa_drive_streams.restype = C.c_int
    drv.vsa_drive_streams.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    arr = lambda xs: (C.c_void_p * len(xs))(*xs)    # noqa: E731
    st = drv.vsa_drive_streams(n_streams, arr([t.g for t in trackers]), arr([t.cvt for t in trackers]), arr([t.det for t in trackers]),
                               arr([t.pyr_delay for t in trackers]), arr([t.pts_delay for t in trackers]), arr(ring), ring_len,
                               (C.c_int * n_streams)(*phases), 0, n_frames)
    assert st == ovx.VX_SUCCESS
    done = {}
    for trk, ph in zip(trackers, phases):
        if ph not in done:                      # streams with the same phase saw the same frames: one oracle run serves them
            ref = OracleTracker(p, omp=True)
            ref.init(frames[ph], None)
            for k in range(n_frames):
                old, cur = ref.track(frames[(ph + 1 + k) % ring_len], None)
            done[ph] = (old, cur, ref.prev_pts)
        old, cur, merged = done[ph]
        assert len(old) == 4096
        np.testing.assert_array_equal(trk.prev(), old)
        got = trk.curr()
        np.testing.assert_array_equal(got[:, 0] >= 0, cur[:, 0] >= 0)
        assert np.abs(got - cur).max() <= 0.01 and got.tobytes() == cur.tobytes()
        np.testing.assert_array_equal(trk.merged(), merged)
        assert (cur[:, 0] >= 0).mean() > 0.9


def test_config4_1080p_graph_with_homography_node(ctx):
    """configs[4] as bench.py --homography builds it: the 1080p graph plus nvxFindHomographyNode (RANSAC, the
    stabilizer's arguments, stabilizer.cpp:264-273) on (previous points, PyrLK matches); 2 frames, H and mask bit for bit"""
    w, h = 1920, 1080
    p = Params(pyr_levels=5, lk_num_iters=5, lk_win_size=21, array_capacity=4096, detector_cell_size=18,
               use_harris_detector=True, harris_k=0.04, harris_thresh=100.0)
    base = synth.base_frame(w, h, 1234)
    frames = [synth.to_rgbx(synth.gray_frame(base, t, 1234)) for t in range(3)]
    ring = _ring(ctx, frames)
    trk = VxFeatureTracker(ctx, p, w, h, ring[0], None, homography=True)
    ref = OracleTracker(p, omp=True)
    ref.init(frames[0])
    for t in (1, 2):
        trk.track(ring[t], None)
        old, cur = _check_frame(trk, ref, frames[t], t)
        Hw, mw, cw = O.find_homography(old, cur, 3.0, 2000, 10, 0.995, 0.45)
        Hg, mg = trk.homography()
        assert Hg.tobytes() == Hw.tobytes(), f"frame {t}"
        assert len(mg) == len(old) == 4096 and mg.tobytes() == mw.tobytes(), f"frame {t}"
        assert cw > 0.9 * (cur[:, 0] >= 0).sum()


def test_config4_homography_graphs_of_many_streams_at_once(ctx):
    """configs[4] under contention: 12 camera streams (own graph and HIP stream each, a Python thread per stream: ctypes drops
    the GIL across vxProcessGraph) run the tracker graph with the FindHomography node for 10 frames at the same time; EVERY
    frame's H and inlier mask of every stream is the oracle's.  The hypothesis workgroups of one launch sit on different XCDs and
    hand their results to the refining workgroup past the L2s (homography.hip): a stale read would pick another hypothesis."""
    import threading
    w, h, n_streams, n_frames, ring_len = 800, 600, 12, 10, 5
    p = Params(pyr_levels=4, lk_num_iters=5, lk_win_size=21, array_capacity=2000, detector_cell_size=18,
               use_harris_detector=True, harris_k=0.04, harris_thresh=100.0)
    base = synth.base_frame(w, h, 777)
    frames = [synth.to_rgbx(synth.gray_frame(base, t, 777)) for t in range(ring_len)]
    ring = _ring(ctx, frames)
    phases = [s % ring_len for s in range(n_streams)]
    trackers = [VxFeatureTracker(ctx, p, w, h, ring[ph], None, homography=True, homography_first=bool(s & 1)) for s, ph in enumerate(phases)]
    got = [[] for _ in range(n_streams)]
    errors = []
    start = threading.Barrier(n_streams)

    def run(s):
        try:
            start.wait()
            for k in range(n_frames):
                trackers[s].track(ring[(phases[s] + 1 + k) % ring_len], None)
                Hg, mg = trackers[s].homography()
                got[s].append((Hg.copy(), mg.copy(), trackers[s].merged().copy()))
        except Exception as e:      # noqa: BLE001
            errors.append((s, repr(e)))

    ths = [threading.Thread(target=run, args=(s,)) for s in range(n_streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    want = {}
    for ph in sorted(set(phases)):
        ref = OracleTracker(p, omp=True)
        ref.init(frames[ph], None)
        seq = []
        for k in range(n_frames):
            old, cur = ref.track(frames[(ph + 1 + k) % ring_len], None)
            Hw, mw, _ = O.find_homography(old, cur, 3.0, 2000, 10, 0.995, 0.45)
            seq.append((Hw, mw, ref.prev_pts.copy()))
        want[ph] = seq
    for s, ph in enumerate(phases):
        for k in range(n_frames):
            assert got[s][k][0].tobytes() == want[ph][k][0].tobytes(), f"stream {s} frame {k}: H"
            assert got[s][k][1].tobytes() == want[ph][k][1].tobytes(), f"stream {s} frame {k}: inlier mask"
            assert got[s][k][2].tobytes() == want[ph][k][2].tobytes(), f"stream {s} frame {k}: merged features"


def test_config2_4k_fast_graph(ctx):
    """configs[2] as one graph: 3840x2160, 7 levels, FAST-9 (threshold 25) track, 10 000 features, 21x21 PyrLK"""
    w, h = 3840, 2160
    p = Params(pyr_levels=7, lk_num_iters=5, lk_win_size=21, array_capacity=10000, detector_cell_size=18,
               use_harris_detector=False, fast_type=9, fast_thresh=25)
    base = synth.base_frame(w, h, 4321)
    frames = [synth.to_rgbx(synth.gray_frame(base, t, 4321)) for t in range(3)]
    ring = _ring(ctx, frames)
    trk = VxFeatureTracker(ctx, p, w, h, ring[0], None)
    ref = OracleTracker(p, omp=True)
    ref.init(frames[0])
    np.testing.assert_array_equal(trk.merged(), ref.prev_pts)
    assert len(ref.prev_pts) == 10000
    for t in (1, 2):
        trk.track(ring[t], None)
        _check_frame(trk, ref, frames[t], t)


def test_concurrent_tracker_init_in_one_context(ctx):
    """one context per process, one host thread per camera stream.  FeatureTracker::init() makes
    immediate-mode calls (vxuColorConvert, vxuGaussianPyramid, nvxuHarrisTrack; feature_tracker.cpp:317-342); run
    from many threads at once they share the context's stream, pinned slots and scratch and must not disturb each
    other.  Frame sizes differ so that the shared detector scratch has to grow while other threads use it."""
    sizes = [(640, 480), (320, 240), (800, 600), (1280, 720), (200, 120), (1920, 1080)] * 2
    p = Params(pyr_levels=3, lk_win_size=10, array_capacity=3000)
    frames, imgs = [], []
    for i, (w, h) in enumerate(sizes):
        f = synth.to_rgbx(synth.gray_frame(synth.base_frame(w, h, 50 + i), 0, 50 + i))
        img = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_RGBX)
        ovx.write_image(img, f)
        frames.append(f)
        imgs.append(img)
    trackers, errors = [None] * len(sizes), []
    start = threading.Barrier(len(sizes))

    def work(i):
        try:
            start.wait()
            trackers[i] = VxFeatureTracker(ctx, p, sizes[i][0], sizes[i][1], imgs[i], None)
        except Exception as e:      # noqa: BLE001
            errors.append((i, repr(e)))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(sizes))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for i, trk in enumerate(trackers):
        ref = OracleTracker(p)
        ref.init(frames[i])
        np.testing.assert_array_equal(trk.merged(), ref.prev_pts, err_msg=f"stream {i} ({sizes[i]})")


def test_homography_node_first_run_with_poisoned_memory(ctx):
    """the node's scratch (ticket word + hypotheses) is allocated at its first execution; whatever the allocation held
    before must not matter.  Fill and free a large block first so that the node's hipMalloc is likely to reuse
    non-zero memory, then require the very first execution of a fresh node to be right."""
    import torch
    from test_oracle_homography import scene
    junk = torch.full((64 << 20,), 0xA5, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()
    src, dst, _ = scene(n=1500, noise=0.4, outliers=0.3, seed=11)
    a = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 1500)
    b = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, 1500)
    ovx.add_items(a, src)
    ovx.add_items(b, dst)
    Hw, mw, _ = O.find_homography(src, dst, 3.0, 2000, 10, 0.995, 0.45)
    for _ in range(3):                               # three fresh graphs, each run once, then once more
        g = L.vxCreateGraph(ctx)
        m = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
        mask = L.vxCreateArray(ctx, ovx.VX_TYPE_UINT8, 1500)
        node = L.nvxFindHomographyNode(g, a, b, m, ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, mask)
        ovx.check(L.vxGetStatus(node), "nvxFindHomographyNode")
        for run in range(2):
            ovx.check(L.vxProcessGraph(g), "vxProcessGraph")
            out = np.zeros((3, 3), np.float32)
            ovx.check(L.vxCopyMatrix(m, out.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST))
            assert out.tobytes() == Hw.tobytes(), f"run {run}"
            assert ovx.read_array(mask, np.uint8).tobytes() == mw.tobytes(), f"run {run}"


def test_blocking_sync_option_runs_the_same_frames(gpu):
    """VISUALSLAM_AMD_BLOCKING_SYNC=1 (vxCreateContext sets hipDeviceScheduleBlockingSync: the host waits for a graph asleep
    instead of polling) changes how the host waits, not what the graph computes: tools/soak.py --
    the headline workload through one graph against the oracle's frame loop, every frame byte for byte -- in a process of its
    own, because the flag is process-wide."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VISUALSLAM_AMD_BLOCKING_SYNC="1")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "6"], env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0 and "SOAK_OK 6 frames identical to the oracle" in res.stdout, res.stdout[-600:] + res.stderr[-600:]
