#!/usr/bin/env python3
"""Generates tests/golden/c1_sources.npz (run once, in the build container, with /root/reference mounted).

BASELINE configs[0] names the reference's bundled 640x480 sample video, which is not in the repository
(/root/reference/.MISSING_LARGE_BLOBS:1) and could not be decoded here anyway.  SURVEY.md section 8d prescribes
frames synthesised from warps of the still images the reference does ship.  This fixture holds the DATA those frames
are made from: a 512x384 crop of data/lena.jpg and of data/baboon.jpg (512x512 RGB, decoded with PIL), box-filtered
to 320x240 RGB to keep the fixture small.  visualslam_amd/synth.py:c1_sequence() turns them into the 640x480 RGBX
sequence (2x bilinear upsampling + a slowly rotating, zooming, translating view; a scene cut from lena to baboon in
the middle).  No oracle output is stored here: the GPU tests compare the HIP path with the oracle on these frames.
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = "/root/reference/VisionWorks/VisionWorks-1.6-Samples/data"


def source(name, y0):
    im = Image.open(os.path.join(DATA, name)).convert("RGB").crop((0, y0, 512, y0 + 384))
    return np.array(im.resize((320, 240), Image.BOX), np.uint8)


def main():
    out = {"lena_rgb": source("lena.jpg", 64), "baboon_rgb": source("baboon.jpg", 40)}
    path = os.path.join(HERE, "c1_sources.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
