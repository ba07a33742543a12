"""nvxcuHarrisTrack_NoAlloc (include/NVX/nvxcu.h): the entry point that allocates nothing -- its cell keys live in a
file-scope device array of the library -- against nvxcuHarrisTrack with a caller-owned tmp_buf AND against the oracle,
through the C ABI.  Bit-identical outputs are required (corner sets, strengths, counts), on two calls in a row on the same
GPU, the second with a smaller input (the array keeps the first call's keys), then a third that grows again; and once
captured into a HIP graph and replayed (the callers the entry point is for).
Collected last (test_zz_*): `pytest -x` must not hide the other tests behind a failure here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O
from visualslam_amd import nvxcu, synth
from visualslam_amd.devmem import DeviceArray, DeviceImage, Scratch, current_target

OK = nvxcu.NVXCU_SUCCESS


def _both(gray, mask, tracked, item_type, cell, cap, k=0.04, thr=100.0):
    import torch
    h, w = gray.shape
    img = DeviceImage(w, h).upload(gray)
    mimg = DeviceImage(w, h).upload(mask) if mask is not None else None
    tarr = DeviceArray(max(len(tracked), 1), item_type).upload(tracked) if tracked is not None else None
    res = []
    for noalloc in (False, True):
        out = DeviceArray(cap, item_type)
        ncorn = torch.zeros(1, dtype=torch.int64, device="cuda:0")
        args = (img.desc, out.desc, mimg.desc if mimg else None, tarr.desc if tarr else None, k, thr, cell, ncorn.data_ptr())
        if noalloc:
            st = nvxcu.harris_track_noalloc(*args, current_target())
        else:
            tmp = Scratch(nvxcu.harris_track_buf_size(w, h, item_type, cap, cell).dev_buf_size)
            st = nvxcu.harris_track(*args, tmp.desc, current_target())
        assert st == OK
        torch.cuda.synchronize()
        res.append((out.buf[: out.num_items() * out.item_size].cpu().numpy().tobytes(), out.num_items(), int(ncorn.item())))
    return res


def _as_items(tracked, item_type):
    return tracked          # every case with tracked points uses POINT2F items


def _scene(w, h, seed):
    return synth.gray_frame(synth.base_frame(w, h, seed), 0, seed)


def test_noalloc_matches_harris_track_on_consecutive_calls(gpu):
    rng = np.random.default_rng(3)
    # (w, h, cell, capacity, item type, with mask, tracked points): large, then smaller, then larger again
    cases = [(1920, 1080, 18, 4096, nvxcu.NVXCU_TYPE_POINT2F, False, 3000),
             (320, 240, 18, 600, nvxcu.NVXCU_TYPE_POINT2F, True, 200),
             (640, 480, 6, 9000, nvxcu.NVXCU_TYPE_KEYPOINTF, False, 0),
             (64, 48, 1, 4000, nvxcu.NVXCU_TYPE_POINT2F, False, 50)]
    for w, h, cell, cap, it, with_mask, nt in cases:
        gray = _scene(w, h, seed=w + cell)
        mask = None
        if with_mask:
            mask = np.zeros((h, w), np.uint8)
            mask[h // 4: 3 * h // 4, w // 4: 3 * w // 4] = 255
        tracked = None
        if nt:
            tracked = np.stack([rng.uniform(-3, w + 3, nt), rng.uniform(-3, h + 3, nt)], 1).astype(np.float32)
            tracked[::7] = (-1.0, -1.0)
        ref, got = _both(gray, mask, tracked, it, cell, cap)
        assert ref[1] > 0 and ref[2] >= ref[1]
        assert got[1:] == ref[1:], (w, h, cell)
        assert got[0] == ref[0], (w, h, cell)
        # ... and both are the oracle's list: a bug shared by the two entry points' common path would not hide
        want, want_total, _ = O.harris_track(gray, mask, _as_items(tracked, it), 0.04, 100.0, cell, cap,
                                             item_type=O.KEYPOINTF if it == nvxcu.NVXCU_TYPE_KEYPOINTF else O.POINT2F)
        assert (got[1], got[2]) == (len(want), want_total), (w, h, cell)
        assert got[0] == want.tobytes(), (w, h, cell)


def test_noalloc_rejects_grids_beyond_its_array(gpu):
    w, h = 640, 480          # cell 1: 307 200 cells > 65 536
    img = DeviceImage(w, h)
    out = DeviceArray(100)
    st = nvxcu.harris_track_noalloc(img.desc, out.desc, None, None, 0.04, 100.0, 1, None, current_target())
    assert st == nvxcu.NVXCU_ERROR_NO_MEMORY
    # the largest grid it takes: 256 x 256 cells of 2 px
    gray = _scene(512, 512, seed=4)
    ref, got = _both(gray, None, None, nvxcu.NVXCU_TYPE_POINT2F, 2, 70000)
    assert got == ref
    # same argument checks as nvxcuHarrisTrack
    assert nvxcu.harris_track_noalloc(img.desc, out.desc, None, None, 0.04, 100.0, 0, None, current_target()) == \
        nvxcu.NVXCU_ERROR_INVALID_PARAMETERS


def test_noalloc_captured_into_a_hip_graph_and_replayed(gpu):
    """The callers this entry point exists for: code under stream capture cannot allocate.  One plain call first (it looks up
    the array's address on this GPU, as the header asks), then the call is captured into a HIP graph -- nothing may allocate or
    synchronise during capture -- and the graph is replayed on two different frames written into the same input buffer."""
    import torch
    w, h, cell, cap = 640, 480, 18, 1500
    scenes = [_scene(w, h, seed=s) for s in (21, 22, 23)]
    img = DeviceImage(w, h).upload(scenes[0])
    out = DeviceArray(cap, nvxcu.NVXCU_TYPE_POINT2F)
    ncorn = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    args = (img.desc, out.desc, None, None, 0.04, 100.0, cell, ncorn.data_ptr())
    assert nvxcu.harris_track_noalloc(*args, current_target()) == OK           # warm-up outside capture
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        assert nvxcu.harris_track_noalloc(*args, current_target()) == OK
    for gray in scenes[1:]:
        img.upload(gray)
        out.count.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want, total, _ = O.harris_track(gray, None, None, 0.04, 100.0, cell, cap)
        n = out.num_items()
        assert n == len(want) > 100 and int(ncorn.item()) == total
        assert out.buf[: n * out.item_size].cpu().numpy().tobytes() == want.tobytes()


def test_noalloc_refuses_a_stream_of_another_device(gpu):
    """with one GPU the mismatch cannot be staged; what can be checked is that the stream's own device passes and that a
    handle that is no stream is an error, not a fault"""
    import torch
    if torch.cuda.device_count() < 2:
        img, out = DeviceImage(64, 48), DeviceArray(10)
        assert nvxcu.harris_track_noalloc(img.desc, out.desc, None, None, 0.04, 100.0, 18, None, current_target()) == OK
        torch.cuda.synchronize()
        return
    with torch.cuda.device(1):
        other = nvxcu.StreamExecTarget(torch.cuda.Stream(device="cuda:1").cuda_stream)
    img, out = DeviceImage(64, 48), DeviceArray(10)
    assert nvxcu.harris_track_noalloc(img.desc, out.desc, None, None, 0.04, 100.0, 18, None, other) == nvxcu.NVXCU_ERROR_INVALID_PARAMETERS
