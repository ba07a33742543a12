"""GPU: the HIP path against the committed golden fixture (tests/golden/front_end_v2.npz), called
through the thin C ABI.  Nothing here needs /root/reference."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O
from visualslam_amd import nvxcu
from visualslam_amd.devmem import DeviceArray, DeviceImage, DevicePyramid, Scratch, current_target

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "front_end_v2.npz"))


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def _front(rgbx, levels):
    import torch
    h, w = rgbx.shape[:2]
    src = DeviceImage(w, h, nvxcu.NVXCU_DF_IMAGE_RGBX).upload(rgbx)
    pyr = DevicePyramid(w, h, levels)
    t = current_target()
    assert nvxcu.color_convert(src.desc, pyr.levels[0].desc, t) == 0
    assert nvxcu.gaussian_pyramid(pyr.desc, t) == 0
    torch.cuda.synchronize()
    return pyr


@pytest.mark.parametrize("name", ["lena", "baboon"])
def test_golden_front_end_on_gpu(gpu, name):
    import torch
    pyr = _front(GOLD[f"{name}_rgbx"], 4)
    lv = pyr.download()
    assert crc(lv[0]) == GOLD[f"{name}_gray_crc"]
    assert [crc(l) for l in lv] == GOLD[f"{name}_pyr_crc"].tolist()
    np.testing.assert_array_equal(lv[3], GOLD[f"{name}_pyr_l3"])
    h, w = lv[0].shape
    mask = np.zeros((h, w), np.uint8)
    mask[30:90, 40:120] = 255
    dmask = DeviceImage(w, h).upload(mask)
    tmp = Scratch(nvxcu.harris_track_buf_size(w, h, nvxcu.NVXCU_TYPE_KEYPOINTF, 200, 18).dev_buf_size)
    t = current_target()
    ncorn = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    for tag, m in (("all", None), ("center", dmask.desc)):
        out = DeviceArray(200, nvxcu.NVXCU_TYPE_KEYPOINTF)
        assert nvxcu.harris_track(pyr.levels[0].desc, out.desc, m, None, 0.04, 100.0, 18, ncorn.data_ptr(), tmp.desc, t) == 0
        torch.cuda.synchronize()
        assert int(ncorn.item()) == GOLD[f"{name}_harris_{tag}_total"]
        assert out.download(O.keypointf_dtype).tobytes() == GOLD[f"{name}_harris_{tag}"].tobytes()
    out = DeviceArray(200, nvxcu.NVXCU_TYPE_KEYPOINTF)
    assert nvxcu.fast_track(pyr.levels[0].desc, out.desc, None, None, 9, 25, 18, ncorn.data_ptr(), tmp.desc, t) == 0
    torch.cuda.synchronize()
    assert int(ncorn.item()) == GOLD[f"{name}_fast9_25_total"]
    assert out.download(O.keypointf_dtype).tobytes() == GOLD[f"{name}_fast9_25"].tobytes()


def test_golden_lk_on_gpu(gpu):
    import torch
    p0 = _front(GOLD["lena_rgbx"], 3)
    g1 = GOLD["lk_g1"]
    p1 = DevicePyramid(g1.shape[1], g1.shape[0], 3)
    p1.levels[0].upload(g1)
    t = current_target()
    assert nvxcu.gaussian_pyramid(p1.desc, t) == 0
    pts = GOLD["lk_pts"]
    for win in (6, 10, 21):
        old = DeviceArray(len(pts)).upload(pts)
        new = DeviceArray(len(pts)).upload(pts)
        assert nvxcu.optical_flow_pyr_lk(p0.desc, p1.desc, old.desc, new.desc, nvxcu.NVXCU_TERM_CRITERIA_BOTH,
                                         0.01, 5, win, t) == 0
        torch.cuda.synchronize()
        got, want = new.download(), GOLD[f"lk_win{win}"]
        np.testing.assert_array_equal(got[:, 0] >= 0, want[:, 0] >= 0)      # status flags
        assert np.abs(got - want).max() <= 0.01                              # north_star tolerance
        assert got.tobytes() == want.tobytes()                               # and bit-identical
