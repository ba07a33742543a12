"""CPU checks of the oracle's nvxFindHomography restatement (oracle/vxoracle_homography.c): analytic properties
and an independent numpy evaluation of its outputs.  PARITY UNPINNED: the reference holds call sites only
(demos/video_stabilizer/stabilizer.cpp:264-273, SfM.cpp:1576-1578, homography_filter_node.cpp:36-159)."""
import numpy as np

import oracle as O

H_TRUE = np.array([[1.01, 0.02, 5.0], [-0.015, 0.99, -3.0], [1e-5, -2e-5, 1.0]])


def scene(n=1200, noise=0.3, outliers=0.3, lost=0.05, seed=0, H=H_TRUE):
    rng = np.random.default_rng(seed)
    src = np.stack([rng.uniform(0, 1920, n), rng.uniform(0, 1080, n)], 1).astype(np.float32)
    p = np.c_[src, np.ones(n)] @ H.T
    dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    dst += rng.normal(0, noise, dst.shape).astype(np.float32)
    out = rng.random(n) < outliers
    dst[out] = np.stack([rng.uniform(0, 1920, out.sum()), rng.uniform(0, 1080, out.sum())], 1)
    gone = rng.random(n) < lost
    dst[gone] = -1.0                                   # PyrLK's lost points
    return src, dst, ~out & ~gone


def reproj_err(Hdata, src, dst):
    H = Hdata.T.astype(np.float64)                     # the API hands out the transposed matrix
    p = np.c_[src.astype(np.float64), np.ones(len(src))] @ H.T
    return np.hypot(p[:, 0] / p[:, 2] - dst[:, 0], p[:, 1] / p[:, 2] - dst[:, 1])


def test_hypothesis_count_is_the_a_priori_ransac_count():
    assert O.homography_hypotheses(0.995, 0.45, 2000) == 56          # stabilizer.cpp:268-272 parameters
    assert O.homography_hypotheses(0.995, 0.45, 20) == 20            # capped by the estimate-iteration limit
    assert O.homography_hypotheses(0.99, 0.0, 2000) == 1
    assert O.homography_hypotheses(0.999, 0.8, 100000) == int(np.ceil(np.log(1 - 0.999) / np.log(1 - 0.2 ** 4)))


def test_recovers_homography_among_outliers_and_lost_points():
    src, dst, good = scene()
    Hd, mask, cnt = O.find_homography(src, dst, threshold=3.0)
    assert cnt == int((mask != 0).sum()) and set(np.unique(mask)) <= {0, 255}
    assert (mask[good] != 0).mean() > 0.97                           # nearly every true inlier found
    assert (mask[dst[:, 0] < 0] == 0).all()                          # lost points never count
    H = Hd.T / Hd.T[2, 2]
    tol = np.array([[2e-3, 2e-3, 0.6], [2e-3, 2e-3, 0.6], [2e-6, 2e-6, 1e-9]])
    assert (np.abs(H - H_TRUE) <= tol).all(), H - H_TRUE
    # the mask is exactly "reprojection error <= threshold" of the returned matrix (away from the boundary)
    e = reproj_err(Hd, src, dst)
    usable = (dst >= 0).all(1)
    assert (mask[usable & (e < 2.99)] != 0).all() and (mask[usable & (e > 3.01)] == 0).all()


def test_exact_data_gives_all_inliers_and_exact_matrix():
    src, dst, good = scene(n=300, noise=0.0, outliers=0.0, lost=0.0, seed=3)
    Hd, mask, cnt = O.find_homography(src, dst, threshold=0.5)       # SfM.cpp:1566 threshold
    usable = (dst >= 0).all(1)                                       # a negative coordinate reads as a lost point
    assert cnt == int(usable.sum()) >= 290 and (mask[usable] == 255).all() and not mask[~usable].any()
    np.testing.assert_allclose(Hd.T, H_TRUE, rtol=2e-4, atol=2e-3)


def test_refinement_never_loses_inliers():
    src, dst, _ = scene(seed=5, noise=0.8)
    _, _, c0 = O.find_homography(src, dst, max_refine_iters=0)
    _, _, c10 = O.find_homography(src, dst, max_refine_iters=10)
    assert c10 >= c0 > 0


def test_degenerate_inputs_give_identity_and_empty_mask():
    src, dst, _ = scene(n=50, seed=2)
    for s, d in ((src[:3], dst[:3]),                                 # fewer than 4 correspondences
                 (src, np.full_like(dst, -1.0)),                     # everything lost
                 (np.tile(src[:1], (50, 1)), dst)):                  # all source points coincide
        Hd, mask, cnt = O.find_homography(s, d)
        assert cnt == 0 and not mask.any()
        np.testing.assert_array_equal(Hd, np.eye(3, dtype=np.float32))


def test_deterministic_and_prefix_independent():
    src, dst, _ = scene(seed=9)
    a = O.find_homography(src, dst)
    b = O.find_homography(src.copy(), dst.copy())
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2] == b[2]
