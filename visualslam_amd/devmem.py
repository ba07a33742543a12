"""Device-memory plumbing for the tests and the benchmark (torch is used only to own HBM buffers
and streams -- the compute goes through the C ABI in nvxcu.py).

Mirrors the allocation helpers of the thin-layer demo
(demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:102-179): pitch-linear images, a pyramid
whose level i is ceil(base * 0.5^i), plain arrays with a device-resident u32 item count.
"""
import math

import numpy as np
import torch

from . import nvxcu

PITCH_ALIGN = 256  # what hipMallocPitch would give


def _pitch(row_bytes):
    return (row_bytes + PITCH_ALIGN - 1) // PITCH_ALIGN * PITCH_ALIGN


class DeviceImage:
    def __init__(self, width, height, fmt=nvxcu.NVXCU_DF_IMAGE_U8, device="cuda:0", pitch=None):
        self.width, self.height, self.format = width, height, fmt
        self.bpp = 4 if fmt == nvxcu.NVXCU_DF_IMAGE_RGBX else 1
        self.pitch = pitch or _pitch(width * self.bpp)
        self.buf = torch.zeros((height, self.pitch), dtype=torch.uint8, device=device)
        self.desc = nvxcu.make_image(self.buf.data_ptr(), self.pitch, width, height, fmt)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint8).reshape(self.height, self.width * self.bpp)
        self.buf[:, : self.width * self.bpp].copy_(torch.from_numpy(arr))
        return self

    def download(self):
        a = self.buf[:, : self.width * self.bpp].cpu().numpy()
        return a.reshape(self.height, self.width, 4) if self.bpp == 4 else a


def pyramid_dims(width, height, levels, scale=0.5):
    """float32 ceil sizing, as feature_tracker_nvxcu.cpp:164-176"""
    dims, cur = [], np.float32(1.0)
    w, h = width, height
    for _ in range(levels):
        dims.append((int(w), int(h)))
        cur = np.float32(cur * np.float32(scale))
        w = int(math.ceil(float(np.float32(width) * cur)))
        h = int(math.ceil(float(np.float32(height) * cur)))
    return dims


class DevicePyramid:
    def __init__(self, width, height, levels, device="cuda:0", pitch_pad=None):
        """pitch_pad: if given, level pitch = width + pitch_pad (to exercise unaligned rows)"""
        self.levels = [DeviceImage(w, h, device=device, pitch=(w + pitch_pad) if pitch_pad is not None else None)
                       for (w, h) in pyramid_dims(width, height, levels)]
        self.desc = nvxcu.make_pyramid([l.desc for l in self.levels])

    def upload(self, arrays):
        for l, a in zip(self.levels, arrays):
            l.upload(a)
        return self

    def download(self):
        return [l.download() for l in self.levels]


class DeviceArray:
    def __init__(self, capacity, item_type=nvxcu.NVXCU_TYPE_POINT2F, device="cuda:0"):
        self.capacity, self.item_type = capacity, item_type
        self.item_size = nvxcu.ITEM_SIZE[item_type]
        self.buf = torch.zeros(max(capacity, 1) * self.item_size, dtype=torch.uint8, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self.desc = nvxcu.make_array(self.buf.data_ptr(), self.count.data_ptr(), capacity, item_type)

    def upload(self, items):
        items = np.ascontiguousarray(items)
        n = len(items)
        assert n <= self.capacity
        raw = items.view(np.uint8).reshape(-1)
        assert raw.size == n * self.item_size
        if n:
            self.buf[: raw.size].copy_(torch.from_numpy(raw.copy()))
        self.count.fill_(n)
        return self

    def num_items(self):
        return int(self.count.item())

    def download(self, dtype=None):
        n = self.num_items()
        raw = self.buf[: n * self.item_size].cpu().numpy()
        if self.item_type == nvxcu.NVXCU_TYPE_POINT2F:
            return raw.view(np.float32).reshape(n, 2).copy()
        return raw.view(dtype).copy() if dtype is not None else raw.copy()


class Scratch:
    def __init__(self, size, device="cuda:0"):
        self.buf = torch.zeros(max(int(size), 8), dtype=torch.uint8, device=device)
        self.desc = nvxcu.nvxcu_tmp_buf_t(self.buf.data_ptr(), None)


def current_target():
    return nvxcu.StreamExecTarget(torch.cuda.current_stream().cuda_stream)
