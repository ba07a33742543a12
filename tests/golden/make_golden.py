#!/usr/bin/env python3
"""Generates tests/golden/front_end_v2.npz (run once, in the build container, with /root/reference mounted).

Inputs are DATA files the reference ships (lena.jpg / baboon.jpg, 512x512 RGB, decoded with PIL and
cropped to 160x120 to keep the fixture small) plus a seeded synthetic frame.  Outputs are what the
oracle (oracle/vxoracle.c) produced when this fixture was frozen.  The reference holds no golden
vector for this path (its arithmetic is in closed libvisionworks) -> PARITY UNPINNED: this fixture
pins the ORACLE against drift, and is what the HIP path is compared with on the GPU box, where
/root/reference does not exist.
"""
import os
import sys
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402
from visualslam_amd import synth  # noqa: E402

DATA = "/root/reference/VisionWorks/VisionWorks-1.6-Samples/data"


def crop_rgbx(name, x0, y0, w=160, h=120):
    im = np.array(Image.open(os.path.join(DATA, name)).convert("RGB"))[y0:y0 + h, x0:x0 + w]
    return np.concatenate([im, np.full((h, w, 1), 255, np.uint8)], axis=2)


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def main():
    out = {}
    lena = crop_rgbx("lena.jpg", 200, 200)
    baboon = crop_rgbx("baboon.jpg", 150, 20)
    out["lena_rgbx"], out["baboon_rgbx"] = lena, baboon
    for name, rgbx in (("lena", lena), ("baboon", baboon)):
        g = O.color_convert(rgbx)
        pyr = O.gaussian_pyramid(g, 4)
        out[f"{name}_gray_crc"] = crc(g)
        out[f"{name}_pyr_crc"] = np.array([crc(l) for l in pyr], np.uint32)
        out[f"{name}_pyr_l3"] = pyr[3]
        mask = np.zeros(g.shape, np.uint8)
        mask[30:90, 40:120] = 255                      # mask_center.png-style centre mask
        for tag, m in (("all", None), ("center", mask)):
            hp, ht, _ = O.harris_track(g, m, None, 0.04, 100.0, 18, 200, item_type=O.KEYPOINTF)
            out[f"{name}_harris_{tag}"] = hp
            out[f"{name}_harris_{tag}_total"] = np.int64(ht)
        fp, ft, _ = O.fast_track(g, None, None, 9, 25, 18, 200, item_type=O.KEYPOINTF)
        out[f"{name}_fast9_25"] = fp
        out[f"{name}_fast9_25_total"] = np.int64(ft)
    # LK: lena crop vs sub-pixel translated copy (+1.37, -0.62), windows 6/10/21, 3 levels
    g0 = O.color_convert(lena)
    g1 = np.clip(np.rint(synth.shift_bilinear(g0.astype(np.float32), 1.37, -0.62)), 0, 255).astype(np.uint8)
    p0, p1 = O.gaussian_pyramid(g0, 3), O.gaussian_pyramid(g1, 3)
    pts, _, _ = O.harris_track(g0, None, None, 0.04, 100.0, 9, 150)
    pts = np.concatenate([pts, np.array([[-1, -1], [0, 0], [159, 119], [80.5, 60.25]], np.float32)])
    out["lk_g1"] = g1
    out["lk_pts"] = pts
    for win in (6, 10, 21):
        out[f"lk_win{win}"] = O.optical_flow_pyr_lk(p0, p1, pts, None, O.TERM_BOTH, 0.01, 5, win)
    # 12-frame synthetic sequence through the whole graph: per-frame counts and position hashes
    from oracle.pipeline import OracleTracker
    from visualslam_amd.thin_tracker import Params
    base = synth.base_frame(320, 240, 99)
    trk = OracleTracker(Params(pyr_levels=4, lk_win_size=10, array_capacity=400))
    trk.init(synth.to_rgbx(synth.gray_frame(base, 0, 99)))
    rec = []
    for t in range(1, 12):
        old, cur = trk.track(synth.to_rgbx(synth.gray_frame(base, t, 99)))
        rec.append((len(old), int((cur[:, 0] >= 0).sum()), len(trk.prev_pts), crc(cur), crc(trk.prev_pts)))
    out["seq_record"] = np.array(rec, np.int64)
    np.savez_compressed(os.path.join(HERE, "front_end_v2.npz"), **out)
    print("wrote", os.path.join(HERE, "front_end_v2.npz"), os.path.getsize(os.path.join(HERE, "front_end_v2.npz")), "bytes")


if __name__ == "__main__":
    main()
