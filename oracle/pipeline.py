"""Oracle-side statement of the feature_tracker graph (colour convert -> Gaussian pyramid -> PyrLK ->
Harris/FAST track), frame by frame, as demos/feature_tracker/feature_tracker.cpp:317-342,386-430
wires it.  Test infrastructure."""
from . import (TERM_BOTH, color_convert, fast_track, gaussian_pyramid, harris_track,
               optical_flow_pyr_lk)


class OracleTracker:
    def __init__(self, params, omp=False):
        self.p, self.omp = params, omp
        self.prev_pyr = self.prev_pts = self.curr = None

    def _detect(self, gray, mask, tracked):
        p = self.p
        if p.use_harris_detector:
            pts, _, _ = harris_track(gray, mask, tracked, p.harris_k, p.harris_thresh, p.detector_cell_size,
                                       p.array_capacity, omp=self.omp)
        else:
            pts, _, _ = fast_track(gray, mask, tracked, p.fast_type, p.fast_thresh, p.detector_cell_size,
                                     p.array_capacity, omp=self.omp)
        return pts

    def init(self, rgbx, mask=None):
        gray = color_convert(rgbx, omp=self.omp)
        self.prev_pyr = gaussian_pyramid(gray, self.p.pyr_levels, omp=self.omp)
        self.prev_pts = self._detect(gray, mask, None)

    def track(self, rgbx, mask=None):
        """returns (old points, their tracked positions); afterwards prev_pts = merged set"""
        gray = color_convert(rgbx, omp=self.omp)
        pyr = gaussian_pyramid(gray, self.p.pyr_levels, omp=self.omp)
        old = self.prev_pts
        cur = optical_flow_pyr_lk(self.prev_pyr, pyr, old, None, TERM_BOTH, 0.01, self.p.lk_num_iters,
                                    self.p.lk_win_size, omp=self.omp)
        self.prev_pts = self._detect(gray, mask, cur)
        self.prev_pyr = pyr
        self.curr = cur
        return old, cur
