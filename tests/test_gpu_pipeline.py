"""GPU parity of the whole per-frame path through the thin C ABI, frame after frame, against the
oracle tracker (call sequence of demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:255-321)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.pipeline import OracleTracker
from visualslam_amd import nvxcu, synth
from visualslam_amd.devmem import DeviceImage
from visualslam_amd.thin_tracker import Params, ThinFeatureTracker


def _frames(w, h, n, seed):
    base = synth.base_frame(w, h, seed)
    return [synth.to_rgbx(synth.gray_frame(base, t, seed)) for t in range(n)]


@pytest.mark.parametrize("harris,win,levels", [(True, 6, 6), (False, 10, 4), (True, 21, 5)])
def test_sequence_matches_oracle(gpu, harris, win, levels):
    w, h, n = 640, 480, 6
    p = Params(pyr_levels=levels, lk_win_size=win, use_harris_detector=harris, array_capacity=2000,
               harris_thresh=2000.0)
    frames = _frames(w, h, n, seed=42 + win)
    mask = np.full((h, w), 255, np.uint8)
    mask[:, : w // 8] = 0
    dmask = DeviceImage(w, h).upload(mask)
    dev = [DeviceImage(w, h, nvxcu.NVXCU_DF_IMAGE_RGBX).upload(f) for f in frames]

    trk = ThinFeatureTracker(p, w, h)
    ref = OracleTracker(p)
    trk.init(dev[0].desc, dmask.desc)
    ref.init(frames[0], mask)
    trk.synchronize()
    np.testing.assert_array_equal(trk.curr_pts.download(), ref.prev_pts)
    for t in range(1, n):
        trk.track(dev[t].desc, dmask.desc)
        trk.synchronize()
        old, cur = ref.track(frames[t], mask)
        np.testing.assert_array_equal(trk.prev_features(), old, err_msg=f"frame {t} prev")
        got = trk.curr_features()
        np.testing.assert_array_equal(got[:, 0] >= 0, cur[:, 0] >= 0, err_msg=f"frame {t} status")
        assert np.abs(got - cur).max() <= 0.01, f"frame {t}"
        assert got.tobytes() == cur.tobytes(), f"frame {t} not bit-identical"
        np.testing.assert_array_equal(trk.curr_pts.download(), ref.prev_pts, err_msg=f"frame {t} merged")
        assert len(cur) > 500 and (cur[:, 0] >= 0).mean() > 0.8


@pytest.mark.parametrize("name,w,h,levels,harris,cap,win", [
    ("C2 1080p / 5 levels / Harris / 4096 features / 21x21", 1920, 1080, 5, True, 4096, 21),
    ("C3 4K / 7 levels / FAST-9 t=25 / 10000 features / 21x21", 3840, 2160, 7, False, 10000, 21)])
def test_baseline_configs_full_size(gpu, name, w, h, levels, harris, cap, win):
    """BASELINE.json configs[1] and configs[2] at their full sizes, two tracked frames, bit for bit
    (the OpenMP build of the oracle keeps this to a few seconds)."""
    p = Params(pyr_levels=levels, lk_win_size=win, use_harris_detector=harris, array_capacity=cap,
               harris_thresh=100.0, fast_type=9, fast_thresh=25)
    frames = _frames(w, h, 3, seed=1234)
    dev = [DeviceImage(w, h, nvxcu.NVXCU_DF_IMAGE_RGBX).upload(f) for f in frames]
    trk = ThinFeatureTracker(p, w, h)
    ref = OracleTracker(p, omp=True)
    trk.init(dev[0].desc)
    ref.init(frames[0])
    for t in (1, 2):
        trk.track(dev[t].desc)
        trk.synchronize()
        old, cur = ref.track(frames[t])
        assert trk.prev_features().tobytes() == old.tobytes(), f"{name}: frame {t} prev"
        assert trk.curr_features().tobytes() == cur.tobytes(), f"{name}: frame {t} tracked"
        assert trk.curr_pts.download().tobytes() == ref.prev_pts.tobytes(), f"{name}: frame {t} merged"
        assert len(cur) == cap and (cur[:, 0] >= 0).mean() > 0.9


def test_u8_input_and_no_mask(gpu):
    w, h = 320, 240
    p = Params(pyr_levels=3, lk_win_size=7, array_capacity=500)
    base = synth.base_frame(w, h, 5)
    grays = [synth.gray_frame(base, t, 5) for t in range(3)]
    dev = [DeviceImage(w, h).upload(g) for g in grays]
    trk = ThinFeatureTracker(p, w, h)
    ref = OracleTracker(p)
    trk.init(dev[0].desc)
    ref.init(synth.to_rgbx(grays[0]))
    for t in (1, 2):
        trk.track(dev[t].desc)
        old, cur = ref.track(synth.to_rgbx(grays[t]))
        trk.synchronize()
        assert trk.curr_features().tobytes() == cur.tobytes()
        np.testing.assert_array_equal(trk.curr_pts.download(), ref.prev_pts)
