"""Seeded synthetic camera frames (numpy) shared by the tests and bench.py.

SURVEY.md section 8d workload: low-pass-filtered uniform noise plus random high-contrast 5x5
blobs so that every detector cell holds a corner; frame t is frame 0 translated by a known
sub-pixel drift (bilinear) plus per-frame noise.
"""
import numpy as np


def _smooth121(a, passes):
    for _ in range(passes):
        p = np.pad(a, ((0, 0), (1, 1)), mode="edge")
        a = (p[:, :-2] + 2.0 * p[:, 1:-1] + p[:, 2:]) * 0.25
        p = np.pad(a, ((1, 1), (0, 0)), mode="edge")
        a = (p[:-2, :] + 2.0 * p[1:-1, :] + p[2:, :]) * 0.25
    return a


def base_frame(width, height, seed=1234, blobs_per_mpix=2900, smooth_passes=4):
    """float32 image in [0,255]: smoothed noise (sigma ~ sqrt(passes/2) px) + 5x5 blobs."""
    rng = np.random.default_rng(seed)
    a = rng.random((height, width), dtype=np.float32) * 255.0
    a = _smooth121(a, smooth_passes)
    a = (a - a.mean()) * (60.0 / max(float(a.std()), 1e-6)) + 128.0
    n = max(1, int(blobs_per_mpix * width * height / 1e6))
    xs = rng.integers(3, max(4, width - 8), n)
    ys = rng.integers(3, max(4, height - 8), n)
    vals = rng.integers(0, 2, n) * 255.0
    for x, y, v in zip(xs, ys, vals):
        a[y:y + 5, x:x + 5] = v
    return np.clip(a, 0, 255).astype(np.float32)


def shift_bilinear(img, dx, dy):
    """out(x, y) = img(x - dx, y - dy), bilinear, replicate border (content moves by +dx,+dy)."""
    h, w = img.shape
    ix, iy = int(np.floor(-dx)), int(np.floor(-dy))
    fx, fy = np.float32(-dx - ix), np.float32(-dy - iy)
    xs = np.clip(np.arange(w) + ix, 0, w - 1)
    xs1 = np.clip(np.arange(w) + ix + 1, 0, w - 1)
    ys = np.clip(np.arange(h) + iy, 0, h - 1)
    ys1 = np.clip(np.arange(h) + iy + 1, 0, h - 1)
    top = img[ys][:, xs] * (1 - fx) + img[ys][:, xs1] * fx
    bot = img[ys1][:, xs] * (1 - fx) + img[ys1][:, xs1] * fx
    return (top * (1 - fy) + bot * fy).astype(np.float32)


def drift(t):
    """known per-frame displacement of frame t relative to frame 0 (pixels)"""
    return (0.8 * t) % 7.0, (0.5 * t) % 5.0


def gray_frame(base, t, seed=1234, noise_sigma=1.0):
    dx, dy = drift(t)
    f = shift_bilinear(base, dx, dy)
    if noise_sigma > 0:
        rng = np.random.default_rng(seed + 1000003 * (t + 1))
        f = f + rng.normal(0.0, noise_sigma, f.shape).astype(np.float32)
    return np.clip(np.rint(f), 0, 255).astype(np.uint8)


def to_rgbx(gray):
    """neutral-grey RGBX frame whose BT.709 luma equals `gray` exactly (54+183+19 = 256)."""
    out = np.empty(gray.shape + (4,), np.uint8)
    out[..., 0] = gray
    out[..., 1] = gray
    out[..., 2] = gray
    out[..., 3] = 255
    return out


def color_rgbx(width, height, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (height, width, 4), dtype=np.uint8)


def _warp_affine_rgb(src, out_w, out_h, m):
    """out(x, y) = src(m @ (x, y, 1)), bilinear, replicate border; src (h, w, 3) u8, m 2x3 float32 -> (out_h, out_w, 3) u8.
    float32 arithmetic in a fixed order, numpy only."""
    h, w = src.shape[:2]
    xs, ys = np.meshgrid(np.arange(out_w, dtype=np.float32), np.arange(out_h, dtype=np.float32))
    m = np.asarray(m, np.float32)
    sx = m[0, 0] * xs + m[0, 1] * ys + m[0, 2]
    sy = m[1, 0] * xs + m[1, 1] * ys + m[1, 2]
    sx = np.clip(sx, 0, w - 1).astype(np.float32)
    sy = np.clip(sy, 0, h - 1).astype(np.float32)
    x0, y0 = np.floor(sx).astype(np.int32), np.floor(sy).astype(np.int32)
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    s = src.astype(np.float32)
    top = s[y0, x0] * (1 - fx) + s[y0, x1] * fx
    bot = s[y1, x0] * (1 - fx) + s[y1, x1] * fx
    return np.clip(np.rint(top * (1 - fy) + bot * fy), 0, 255).astype(np.uint8)


def c1_sequence(sources, n_frames, width=640, height=480):
    """BASELINE configs[0] stand-in: 640x480 RGBX frames synthesised from the still images the
    reference ships (tests/golden/c1_sources.npz: 320x240 RGB crops of lena.jpg / baboon.jpg).  Each clip is the source
    upsampled 2x under a view that rotates 0.15 degrees, zooms 0.3 % and pans (1.3, -0.7) px per frame; the first half
    of the sequence shows lena, the second half baboon (a scene cut: every track is lost and detection starts over)."""
    clips = [sources["lena_rgb"], sources["baboon_rgb"]]
    cut = (n_frames + 1) // 2
    frames = []
    for t in range(n_frames):
        src, k = (clips[0], t) if t < cut else (clips[1], t - cut)
        a = np.deg2rad(0.15 * k)
        z = 1.0 / (1.0 + 0.003 * k)
        cx, cy = 0.5 * (width - 1), 0.5 * (height - 1)
        c, s = np.cos(a) * z, np.sin(a) * z
        # output pixel -> source pixel: rotate/zoom about the frame centre, pan, then the 2x scale of the source
        m = np.array([[c, -s, cx - c * cx + s * cy - 1.3 * k], [s, c, cy - s * cx - c * cy + 0.7 * k]], np.float64) * 0.5
        rgb = _warp_affine_rgb(src, width, height, m)
        frames.append(np.concatenate([rgb, np.full((height, width, 1), 255, np.uint8)], axis=2))
    return frames
