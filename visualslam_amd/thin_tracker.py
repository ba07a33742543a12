"""Host-side mirror of nvxcu::FeatureTracker (demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp)
on top of the thin C ABI: same init()/track() call sequence, same buffers, one HIP stream per
camera stream, no host synchronisation inside track().  Used by the tests and by bench.py.
"""
from dataclasses import dataclass

import torch

from . import nvxcu
from .devmem import DeviceArray, DeviceImage, DevicePyramid, Scratch


@dataclass
class Params:
    """defaults: feature_tracker_nvxcu.cpp:487-506"""
    pyr_levels: int = 6
    lk_num_iters: int = 5
    lk_win_size: int = 10
    array_capacity: int = 2000
    detector_cell_size: int = 18
    use_harris_detector: bool = True
    harris_k: float = 0.04
    harris_thresh: float = 100.0
    fast_type: int = 9
    fast_thresh: int = 25


class ThinFeatureTracker:
    def __init__(self, params, width, height, device="cuda:0", stream=None):
        self.p, self.width, self.height, self.device = params, width, height, device
        self.stream = stream or torch.cuda.current_stream(device)
        self.target = nvxcu.StreamExecTarget(self.stream.cuda_stream)
        p = params
        # createDataObjects(): feature_tracker_nvxcu.cpp:387-450
        self.prev_pyr = DevicePyramid(width, height, p.pyr_levels, device)
        self.curr_pyr = DevicePyramid(width, height, p.pyr_levels, device)
        self.prev_pts = DeviceArray(p.array_capacity, device=device)
        self.curr_pts = DeviceArray(p.array_capacity, device=device)
        self.kp_curr_list = DeviceArray(p.array_capacity, device=device)
        size_fn = nvxcu.harris_track_buf_size if p.use_harris_detector else nvxcu.fast_track_buf_size
        sz = size_fn(width, height, nvxcu.NVXCU_TYPE_POINT2F, p.array_capacity, p.detector_cell_size)
        self.keypoints_buf = Scratch(sz.dev_buf_size, device)
        self.num_corners = torch.zeros(1, dtype=torch.int64, device=device)
        self.lk_events = None   # set to a list to collect (start, end) HIP event pairs around PyrLK

    def _check(self, st, what):
        if st != nvxcu.NVXCU_SUCCESS:
            raise RuntimeError(f"{what} failed with nvxcu status {st}")

    def _detect(self, tracked):
        p = self.p
        gray = self.curr_pyr.levels[0].desc
        if p.use_harris_detector:
            st = nvxcu.harris_track(gray, self.curr_pts.desc, self.mask, tracked, p.harris_k, p.harris_thresh,
                                    p.detector_cell_size, self.num_corners.data_ptr(), self.keypoints_buf.desc,
                                    self.target)
        else:
            st = nvxcu.fast_track(gray, self.curr_pts.desc, self.mask, tracked, p.fast_type, p.fast_thresh,
                                  p.detector_cell_size, self.num_corners.data_ptr(), self.keypoints_buf.desc,
                                  self.target)
        self._check(st, "corner track")

    def _front(self, frame):
        if frame.base.format == nvxcu.NVXCU_DF_IMAGE_RGBX or frame.base.format == nvxcu.NVXCU_DF_IMAGE_U8:
            self._check(nvxcu.color_convert(frame, self.curr_pyr.levels[0].desc, self.target), "color convert")
        else:
            raise ValueError("frame must be RGBX or U8")
        self._check(nvxcu.gaussian_pyramid(self.curr_pyr.desc, self.target), "gaussian pyramid")

    def init(self, frame, mask=None):
        """processFirstFrame(): feature_tracker_nvxcu.cpp:457-484 (frame/mask: nvxcu_pitch_linear_image_t)"""
        self.mask = mask
        with torch.cuda.stream(self.stream):
            self._front(frame)
            self._detect(None)

    def track(self, frame, mask=None):
        """track(): feature_tracker_nvxcu.cpp:255-321, minus the final stream synchronise"""
        self.mask = mask
        p = self.p
        self.curr_pts, self.prev_pts = self.prev_pts, self.curr_pts
        self.curr_pyr, self.prev_pyr = self.prev_pyr, self.curr_pyr
        with torch.cuda.stream(self.stream):
            self._front(frame)
            # initial estimate = previous positions (count + payload), :288-295
            self.kp_curr_list.count.copy_(self.prev_pts.count, non_blocking=True)
            self.kp_curr_list.buf.copy_(self.prev_pts.buf, non_blocking=True)
            if self.lk_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
            st = nvxcu.optical_flow_pyr_lk(self.prev_pyr.desc, self.curr_pyr.desc, self.prev_pts.desc,
                                           self.kp_curr_list.desc, nvxcu.NVXCU_TERM_CRITERIA_BOTH, 0.01,
                                           p.lk_num_iters, p.lk_win_size, self.target)
            if self.lk_events is not None:
                e1.record(self.stream)
                self.lk_events.append((e0, e1))
            self._check(st, "optical flow")
            self._detect(self.kp_curr_list.desc)

    def synchronize(self):
        self.stream.synchronize()

    def prev_features(self):
        return self.prev_pts.download()

    def curr_features(self):
        return self.kp_curr_list.download()
