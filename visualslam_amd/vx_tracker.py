"""Host-side mirror of nvx::FeatureTracker (demos/feature_tracker/feature_tracker.cpp) on the OpenVX
node/graph C API of libvisualslam_amd: the same objects, the same graph, the same per-frame call
sequence (vxSetParameterByIndex x2, vxAgeDelay x2, vxProcessGraph).  Used by the tests and bench.py."""
from . import ovx

L = ovx.lib()


class VxFeatureTracker:
    """call-for-call transcription of FeatureTrackerImpl (feature_tracker.cpp:127-449)"""

    def __init__(self, ctx, p, w, h, first_frame, mask, homography=False, detect_on_copy=False, homography_first=False):
        """homography=True appends the video stabilizer's / SFM sample's next stage to the same graph
        (demos/video_stabilizer/stabilizer.cpp:264-273): nvxFindHomographyNode on the PyrLK matches.
        detect_on_copy=True (tests): the detector reads a copy of the grey frame that an nvxCopyImageNode created AFTER the
        PyrLK node makes, i.e. a graph in which the detector's input is not ready when PyrLK starts.
        homography_first=True creates the FindHomography node BEFORE the detector node, the stabilizer's creation order
        (stabilizer.cpp:266,299): it then precedes the detector node in the execution order"""
        self.ctx, self.p = ctx, p
        pex = L.vxCreatePyramid(ctx, p.pyr_levels, 0.5, w, h, ovx.VX_DF_IMAGE_U8)
        self.pyr_delay = L.vxCreateDelay(ctx, pex, 2)
        ovx.release("Pyramid", pex)
        aex = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, p.array_capacity)
        self.pts_delay = L.vxCreateDelay(ctx, aex, 2)
        ovx.release("Array", aex)
        self.kp_curr = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, p.array_capacity)
        # createMainGraph
        self.g = L.vxCreateGraph(ctx)
        pyr0 = L.vxGetReferenceFromDelay(self.pyr_delay, 0)
        gray = L.vxGetPyramidLevel(pyr0, 0)
        eps, it, est = ovx.scalar(ctx, ovx.VX_TYPE_FLOAT32, 0.01), ovx.scalar(ctx, ovx.VX_TYPE_UINT32, p.lk_num_iters), ovx.scalar(ctx, ovx.VX_TYPE_BOOL, 0)
        self.cvt = L.vxColorConvertNode(self.g, first_frame, gray)
        self.pyr = L.vxGaussianPyramidNode(self.g, gray, pyr0)
        self.lk = L.vxOpticalFlowPyrLKNode(self.g, L.vxGetReferenceFromDelay(self.pyr_delay, -1), pyr0,
                                           L.vxGetReferenceFromDelay(self.pts_delay, -1), L.vxGetReferenceFromDelay(self.pts_delay, -1),
                                           self.kp_curr, ovx.VX_TERM_CRITERIA_BOTH, eps, it, est, p.lk_win_size)
        out = L.vxGetReferenceFromDelay(self.pts_delay, 0)
        self.hg = None
        if homography and homography_first:
            self._add_homography(ctx, p)
        det_in = gray
        if detect_on_copy:
            det_in = self.det_copy = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_U8)
            self.copy = L.nvxCopyImageNode(self.g, gray, det_in)
            assert L.vxGetStatus(self.copy) == ovx.VX_SUCCESS
        if p.use_harris_detector:
            self.det = L.nvxHarrisTrackNode(self.g, det_in, out, mask, self.kp_curr, p.harris_k, p.harris_thresh, p.detector_cell_size, None)
        else:
            self.det = L.nvxFastTrackNode(self.g, det_in, out, mask, self.kp_curr, p.fast_type, p.fast_thresh, p.detector_cell_size, None)
        if not homography:
            self.hg = self.hg_matrix = self.hg_mask = None
        if homography and not homography_first:
            self._add_homography(ctx, p)
        for n in (self.cvt, self.pyr, self.lk, self.det):
            assert L.vxGetStatus(n) == ovx.VX_SUCCESS
        opt = b"-O3"
        ovx.check(L.vxSetGraphAttribute(self.g, ovx.NVX_GRAPH_VERIFY_OPTIONS, opt, len(opt)))
        ovx.check(L.vxVerifyGraph(self.g), "vxVerifyGraph")
        for s in (eps, it, est):
            ovx.release("Scalar", s)
        ovx.release("Image", gray)
        # processFirstFrame
        w_, h_ = w, h
        fgray = L.vxCreateImage(ctx, w_, h_, ovx.VX_DF_IMAGE_U8)
        ovx.check(L.vxuColorConvert(ctx, first_frame, fgray), "vxuColorConvert")
        ovx.check(L.vxuGaussianPyramid(ctx, fgray, L.vxGetReferenceFromDelay(self.pyr_delay, 0)), "vxuGaussianPyramid")
        if p.use_harris_detector:
            ovx.check(L.nvxuHarrisTrack(ctx, fgray, L.vxGetReferenceFromDelay(self.pts_delay, 0), mask, None,
                                        p.harris_k, p.harris_thresh, p.detector_cell_size, None), "nvxuHarrisTrack")
        else:
            ovx.check(L.nvxuFastTrack(ctx, fgray, L.vxGetReferenceFromDelay(self.pts_delay, 0), mask, None,
                                      p.fast_type, p.fast_thresh, p.detector_cell_size, None), "nvxuFastTrack")
        ovx.release("Image", fgray)

    def _add_homography(self, ctx, p):
        self.hg_matrix = L.vxCreateMatrix(ctx, ovx.VX_TYPE_FLOAT32, 3, 3)
        self.hg_mask = L.vxCreateArray(ctx, ovx.VX_TYPE_UINT8, p.array_capacity)
        self.hg = L.nvxFindHomographyNode(self.g, L.vxGetReferenceFromDelay(self.pts_delay, -1), self.kp_curr, self.hg_matrix,
                                          ovx.NVX_FIND_HOMOGRAPHY_METHOD_RANSAC, 3.0, 2000, 10, 0.995, 0.45, self.hg_mask)
        assert L.vxGetStatus(self.hg) == ovx.VX_SUCCESS

    def track(self, frame, mask, schedule_only=False):
        ovx.check(L.vxSetParameterByIndex(self.cvt, 0, frame))
        ovx.check(L.vxSetParameterByIndex(self.det, 2, mask))
        ovx.check(L.vxAgeDelay(self.pyr_delay))
        ovx.check(L.vxAgeDelay(self.pts_delay))
        if schedule_only:
            ovx.check(L.vxScheduleGraph(self.g), "vxScheduleGraph")
        else:
            ovx.check(L.vxProcessGraph(self.g), "vxProcessGraph")

    def prev(self):
        return ovx.read_array(L.vxGetReferenceFromDelay(self.pts_delay, -1))

    def curr(self):
        return ovx.read_array(self.kp_curr)

    def merged(self):
        return ovx.read_array(L.vxGetReferenceFromDelay(self.pts_delay, 0))

    def homography(self):
        """(H as the matrix object stores it: element [c][r] = H[r][c], inlier mask)"""
        import numpy as np
        m = np.zeros((3, 3), np.float32)
        ovx.check(L.vxCopyMatrix(self.hg_matrix, m.ctypes.data, ovx.VX_READ_ONLY, ovx.VX_MEMORY_TYPE_HOST))
        return m, ovx.read_array(self.hg_mask)

    def wait(self):
        ovx.check(L.vxWaitGraph(self.g), "vxWaitGraph")


class VxSfmFrontEnd:
    """The feature-tracking front end of the SFM sample's wrapper, nvx::SfM (VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp),
    call for call: createDataObjects (:900-957), processFirstFrame (:956-968), the front-end nodes of createMainGraph (:970-1003,
    1077-1078) and track() (:623-704).  It differs from the feature tracker's graph where the reference's does:
      * the grey frame is a VIRTUAL image and the pyramid is built FROM it (:977-988) instead of the frame being converted straight
        into level 0;
      * the delays are aged BEFORE the frame is validated and the node parameters are set (:625-627);
      * the mask parameter is set only when a mask is given (:658) -- the node keeps the mask it was created with otherwise;
      * Harris only, k / threshold / cell from sfm_config.ini (0.01 / 5.0 / 18), window 10, 6 levels.
    The nvxSfm* nodes between PyrLK and HarrisTrack belong to the closed libvisionworks_sfm and are not part of this path."""

    def __init__(self, ctx, p, w, h, first_frame, mask):
        self.ctx, self.p = ctx, p
        pex = L.vxCreatePyramid(ctx, p.pyr_levels, 0.5, w, h, ovx.VX_DF_IMAGE_U8)
        self.pyr_delay = L.vxCreateDelay(ctx, pex, 2)
        ovx.release("Pyramid", pex)
        aex = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, p.array_capacity)
        self.pts_delay = L.vxCreateDelay(ctx, aex, 2)
        ovx.release("Array", aex)
        self.kp_tracked = L.vxCreateArray(ctx, ovx.NVX_TYPE_POINT2F, p.array_capacity)
        # createMainGraph(firstFrame, mask)
        self.g = L.vxCreateGraph(ctx)
        gray = L.vxCreateVirtualImage(self.g, w, h, ovx.VX_DF_IMAGE_U8)
        assert L.vxGetStatus(gray) == ovx.VX_SUCCESS
        self.cvt = L.vxColorConvertNode(self.g, first_frame, gray)
        pyr0 = L.vxGetReferenceFromDelay(self.pyr_delay, 0)
        self.pyr = L.vxGaussianPyramidNode(self.g, gray, pyr0)
        eps, it, est = ovx.scalar(ctx, ovx.VX_TYPE_FLOAT32, 0.01), ovx.scalar(ctx, ovx.VX_TYPE_UINT32, p.lk_num_iters), ovx.scalar(ctx, ovx.VX_TYPE_BOOL, 0)
        self.lk = L.vxOpticalFlowPyrLKNode(self.g, L.vxGetReferenceFromDelay(self.pyr_delay, -1), pyr0,
                                           L.vxGetReferenceFromDelay(self.pts_delay, -1), L.vxGetReferenceFromDelay(self.pts_delay, -1),
                                           self.kp_tracked, ovx.VX_TERM_CRITERIA_BOTH, eps, it, est, p.lk_win_size)
        self.det = L.nvxHarrisTrackNode(self.g, gray, L.vxGetReferenceFromDelay(self.pts_delay, 0), mask, self.kp_tracked,
                                        p.harris_k, p.harris_thresh, p.detector_cell_size, None)
        for n in (self.cvt, self.pyr, self.lk, self.det):
            assert L.vxGetStatus(n) == ovx.VX_SUCCESS
        ovx.check(L.vxVerifyGraph(self.g), "vxVerifyGraph")
        for s in (eps, it, est):
            ovx.release("Scalar", s)
        ovx.release("Image", gray)
        # processFirstFrame(firstFrame, mask)
        fgray = L.vxCreateImage(ctx, w, h, ovx.VX_DF_IMAGE_U8)
        ovx.check(L.vxuColorConvert(ctx, first_frame, fgray), "vxuColorConvert")
        ovx.check(L.vxuGaussianPyramid(ctx, fgray, L.vxGetReferenceFromDelay(self.pyr_delay, 0)), "vxuGaussianPyramid")
        ovx.check(L.nvxuHarrisTrack(ctx, fgray, L.vxGetReferenceFromDelay(self.pts_delay, 0), mask, None,
                                    p.harris_k, p.harris_thresh, p.detector_cell_size, None), "nvxuHarrisTrack")
        ovx.release("Image", fgray)

    def track(self, frame, mask):
        ovx.check(L.vxAgeDelay(self.pyr_delay))                     # SfM.cpp:625-627: aged first
        ovx.check(L.vxAgeDelay(self.pts_delay))
        if mask:
            ovx.check(L.vxSetParameterByIndex(self.det, 2, mask))   # :658: only when there is one
        ovx.check(L.vxSetParameterByIndex(self.cvt, 0, frame))      # :663
        n = ovx.num_items(L.vxGetReferenceFromDelay(self.pts_delay, -1))    # :665-666 (feeds the closed RANSAC sampler)
        ovx.check(L.vxProcessGraph(self.g), "vxProcessGraph")       # :704
        return n

    def prev(self):
        return ovx.read_array(L.vxGetReferenceFromDelay(self.pts_delay, -1))

    def curr(self):
        return ovx.read_array(self.kp_tracked)

    def merged(self):
        return ovx.read_array(L.vxGetReferenceFromDelay(self.pts_delay, 0))
