/*
 * vxoracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see vxoracle.h header comment).
 * PARITY UNPINNED: no golden vectors exist in the reference for this path.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -mfma  (see oracle/Makefile).
 * Floating-point contract: every fused multiply-add is an explicit fmaf(); everything
 * else is a separately rounded IEEE binary32 operation.  The HIP kernels follow the
 * same operation order so that float results agree bit for bit, not just to tolerance.
 */
#include "vxoracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------
 * Pyramid level sizes.  Pinned by the caller-side allocation loop in
 * demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:164-176 (and
 * samples/object_tracker_nvxcu/object_tracker_keypoint_nvxcu.cpp:364-375):
 *   cur_scale *= scale; cur_width = (uint32_t)ceilf((float)width * cur_scale);
 * ---------------------------------------------------------------------------------------- */
void orc_pyramid_dims(uint32_t width, uint32_t height, uint32_t levels, float scale,
                      uint32_t *widths, uint32_t *heights)
{
    uint32_t cw = width, ch = height;
    float cur_scale = 1.0f;
    for (uint32_t i = 0; i < levels; ++i) {
        widths[i] = cw;
        heights[i] = ch;
        cur_scale *= scale;
        cw = (uint32_t)ceilf((float)width * cur_scale);
        ch = (uint32_t)ceilf((float)height * cur_scale);
    }
}

/* ------------------------------------------------------------------------------------------
 * RGBX -> U8 luma.  Call sites: demos/feature_tracker/feature_tracker.cpp:322 (vxuColorConvert),
 * :386 (vxColorConvertNode); demos/feature_tracker_nvxcu/feature_tracker_nvxcu.cpp:279-282
 * (NVXCU_COLOR_SPACE_DEFAULT, NVXCU_CHANNEL_RANGE_FULL).
 * [S] OpenVX default colour space is BT.709: Y = 0.2126 R + 0.7152 G + 0.0722 B.
 * [P] 8-bit fixed point with weights 54/183/19 (sum 256) and round-to-nearest.
 * ---------------------------------------------------------------------------------------- */
void orc_color_convert_rgbx_u8(const uint8_t *src, uint32_t src_pitch,
                               uint8_t *dst, uint32_t dst_pitch,
                               uint32_t width, uint32_t height)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t *s = src + (size_t)y * src_pitch;
        uint8_t *d = dst + (size_t)y * dst_pitch;
        for (uint32_t x = 0; x < width; ++x) {
            uint32_t r = s[4 * x + 0], g = s[4 * x + 1], b = s[4 * x + 2];
            d[x] = (uint8_t)((54u * r + 183u * g + 19u * b + 128u) >> 8);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * One HALF-scale Gaussian pyramid step.  Call sites: demos/feature_tracker/feature_tracker.cpp:323
 * (vxuGaussianPyramid), :393 (vxGaussianPyramidNode); feature_tracker_nvxcu.cpp:284-285.
 * [S] 5x5 binomial kernel [1 4 6 4 1]x[1 4 6 4 1]/256 followed by decimation by two.
 * [P] even sample phase dst(x,y) = blur(src)(2x,2y); REPLICATE border (a legal refinement of
 *     the UNDEFINED border the thin-layer demo passes, feature_tracker_nvxcu.cpp:219, and
 *     exactly what object_tracker_keypoint_nvxcu.cpp:1639 asks for); one rounding
 *     (sum + 128) >> 8 on the exact 2-D integer sum.
 * ---------------------------------------------------------------------------------------- */
void orc_gaussian_half(const uint8_t *src, uint32_t src_pitch, uint32_t sw, uint32_t sh,
                       uint8_t *dst, uint32_t dst_pitch, uint32_t dw, uint32_t dh)
{
    static const int wgt[5] = { 1, 4, 6, 4, 1 };
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (uint32_t y = 0; y < dh; ++y) {
        for (uint32_t x = 0; x < dw; ++x) {
            uint32_t sum = 0;
            for (int j = 0; j < 5; ++j) {
                int sy = clampi(2 * (int)y + j - 2, 0, (int)sh - 1);
                const uint8_t *row = src + (size_t)sy * src_pitch;
                uint32_t hs = 0;
                for (int i = 0; i < 5; ++i) {
                    int sx = clampi(2 * (int)x + i - 2, 0, (int)sw - 1);
                    hs += (uint32_t)wgt[i] * row[sx];
                }
                sum += (uint32_t)wgt[j] * hs;
            }
            dst[(size_t)y * dst_pitch + x] = (uint8_t)((sum + 128u) >> 8);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Harris response.  Stage (i) of nvxHarrisTrackNode (demos/feature_tracker/feature_tracker.cpp:
 * 418-422; thin layer feature_tracker_nvxcu.cpp:307-310).
 * [P] Sobel 3x3 gradients (exact int), structure tensor summed over the 3x3 block (exact int32,
 *     |sum| <= 9*1020^2 < 2^24 so the float conversion is exact), gradient scale 1/12 per
 *     gradient (= 1/144 per tensor entry), R = (A*C - B*B) - k*(A+C)*(A+C) in binary32 with no
 *     contraction.  R is defined where the whole 5x5 support lies inside the image
 *     (2 <= x <= w-3, 2 <= y <= h-3); elsewhere -inf.
 * ---------------------------------------------------------------------------------------- */
static inline float harris_r_from_sums(int32_t a, int32_t b, int32_t c, float k)
{
    const float S = 1.0f / 144.0f;
    float A = (float)a * S;
    float B = (float)b * S;
    float C = (float)c * S;
    float det = A * C - B * B;
    float tr = A + C;
    return det - k * tr * tr;
}

void orc_harris_response(const uint8_t *img, uint32_t pitch, uint32_t w, uint32_t h,
                         float k, float *resp)
{
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) resp[i] = -INFINITY;
    if (w < 5 || h < 5) return;
    int16_t *gx = (int16_t *)calloc(n, sizeof(int16_t));
    int16_t *gy = (int16_t *)calloc(n, sizeof(int16_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (uint32_t y = 1; y < h - 1; ++y) {
        const uint8_t *r0 = img + (size_t)(y - 1) * pitch;
        const uint8_t *r1 = img + (size_t)y * pitch;
        const uint8_t *r2 = img + (size_t)(y + 1) * pitch;
        for (uint32_t x = 1; x + 1 < w; ++x) {
            int ix = (r0[x + 1] + 2 * r1[x + 1] + r2[x + 1]) - (r0[x - 1] + 2 * r1[x - 1] + r2[x - 1]);
            int iy = (r2[x - 1] + 2 * r2[x] + r2[x + 1]) - (r0[x - 1] + 2 * r0[x] + r0[x + 1]);
            gx[(size_t)y * w + x] = (int16_t)ix;
            gy[(size_t)y * w + x] = (int16_t)iy;
        }
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (uint32_t y = 2; y < h - 2; ++y) {
        for (uint32_t x = 2; x + 2 < w; ++x) {
            int32_t a = 0, b = 0, c = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    size_t q = (size_t)(y + dy) * w + (x + dx);
                    int32_t ix = gx[q], iy = gy[q];
                    a += ix * ix;
                    b += ix * iy;
                    c += iy * iy;
                }
            resp[(size_t)y * w + x] = harris_r_from_sums(a, b, c, k);
        }
    }
    free(gx);
    free(gy);
}

/* ------------------------------------------------------------------------------------------
 * FAST-N strength.  Stage (i) of nvxFastTrackNode (demos/feature_tracker/feature_tracker.cpp:
 * 426-430; N = fast_type in 9..12, main_feature_tracker.cpp:122-123).
 * [S] 16-pixel Bresenham circle of radius 3; a pixel is a corner at threshold t when N
 *     contiguous circle pixels are all > Ip + t or all < Ip - t.
 * [P] strength = the largest t >= 0 for which the pixel is still a corner (OpenVX FastCorners
 *     definition), -1 when it is not a corner even at t = 0; defined for 3 <= x <= w-4,
 *     3 <= y <= h-4.
 * ---------------------------------------------------------------------------------------- */
static const int fast_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int fast_dy[16] = { -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3 };

void orc_fast_strength(const uint8_t *img, uint32_t pitch, uint32_t w, uint32_t h,
                       uint32_t type_n, int32_t *strength)
{
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) strength[i] = -1;
    if (w < 7 || h < 7) return;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (uint32_t y = 3; y < h - 3; ++y) {
        for (uint32_t x = 3; x + 3 < w; ++x) {
            int ip = img[(size_t)y * pitch + x];
            int d[16];
            for (int i = 0; i < 16; ++i)
                d[i] = (int)img[(size_t)((int)y + fast_dy[i]) * pitch + ((int)x + fast_dx[i])] - ip;
            int best = 0; /* max over arcs of min over arc of +-d; <= 0 means no arc */
            for (int s = 0; s < 16; ++s) {
                int mb = 255, md = 255;
                for (uint32_t i = 0; i < type_n; ++i) {
                    int v = d[(s + (int)i) & 15];
                    if (v < mb) mb = v;
                    if (-v < md) md = -v;
                }
                if (mb > best) best = mb;
                if (md > best) best = md;
            }
            strength[(size_t)y * w + x] = best - 1;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Cell selection + tracked-point merge shared by HarrisTrack and FastTrack.
 * Pinned by the reference: "divide the image into cells of the same size and look for corners
 * independently in each cell" (demos/feature_tracker/feature_tracker_user_guide.md:17-20),
 * cell_size is the NMS cell (:117-120), features only where mask != 0 (:148-153), run after LK
 * "to restore lost points" (:32-33), tracked may be NULL (feature_tracker.cpp:326-330).
 * [P] 3x3 local maximum with raster tie-break (p beats q when s(p) > s(q), or s(p) == s(q) and
 *     p precedes q in raster order); per cell the candidate with the largest score >= threshold
 *     and mask != 0, tie -> lowest raster index; a cell holding a surviving tracked point emits
 *     no new corner; output = surviving tracked points in input order, then new corners in cell
 *     raster order, truncated at capacity.
 * A tracked POINT2F survives when 0 <= x <= w-1 and 0 <= y <= h-1 (lost = (-1,-1),
 * nvxio/src/OpenGLShaders.hpp:635-641); a KEYPOINTF additionally needs tracking_status != 0.
 * ---------------------------------------------------------------------------------------- */
static int point_alive(int item_type, const void *pts, uint32_t i, uint32_t w, uint32_t h,
                       float *px, float *py)
{
    float x, y;
    if (item_type == ORC_ITEM_POINT2F) {
        const orc_point2f_t *p = (const orc_point2f_t *)pts + i;
        x = p->x; y = p->y;
    } else {
        const orc_keypointf_t *p = (const orc_keypointf_t *)pts + i;
        x = p->x; y = p->y;
        if (p->tracking_status == 0) return 0;
    }
    *px = x; *py = y;
    return (x >= 0.0f && y >= 0.0f && x <= (float)(w - 1) && y <= (float)(h - 1)) ? 1 : 0;
}

static uint32_t select_and_merge(const float *score, uint32_t w, uint32_t h,
                                 const uint8_t *mask, uint32_t mask_pitch,
                                 int item_type, const void *tracked, uint32_t n_tracked,
                                 float threshold, uint32_t cs,
                                 void *out, uint32_t capacity, size_t *num_corners,
                                 float *strength_out)
{
    uint32_t ncx = (w + cs - 1) / cs, ncy = (h + cs - 1) / cs;
    size_t ncells = (size_t)ncx * ncy;
    uint8_t *occupied = (uint8_t *)calloc(ncells, 1);
    size_t total = 0;
    uint32_t nout = 0;
    size_t isz = item_type == ORC_ITEM_POINT2F ? sizeof(orc_point2f_t) : sizeof(orc_keypointf_t);

    for (uint32_t i = 0; tracked && i < n_tracked; ++i) {
        float x, y;
        if (!point_alive(item_type, tracked, i, w, h, &x, &y)) continue;
        uint32_t cx = (uint32_t)x / cs, cy = (uint32_t)y / cs;
        if (cx >= ncx) cx = ncx - 1;
        if (cy >= ncy) cy = ncy - 1;
        occupied[(size_t)cy * ncx + cx] = 1;
        if (nout < capacity) {
            memcpy((uint8_t *)out + (size_t)nout * isz, (const uint8_t *)tracked + (size_t)i * isz, isz);
            if (strength_out) strength_out[nout] = 0.0f;
            ++nout;
        }
        ++total;
    }

    uint8_t *cfound = (uint8_t *)calloc(ncells, 1);
    float *cbest = (float *)malloc(ncells * sizeof(float));
    uint32_t *cpos = (uint32_t *)malloc(ncells * 2 * sizeof(uint32_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (uint32_t cy = 0; cy < ncy; ++cy) {
        for (uint32_t cx = 0; cx < ncx; ++cx) {
            size_t ci = (size_t)cy * ncx + cx;
            if (occupied[ci]) continue;
            uint32_t x0 = cx * cs, y0 = cy * cs;
            uint32_t x1 = x0 + cs < w ? x0 + cs : w, y1 = y0 + cs < h ? y0 + cs : h;
            int found = 0;
            float best = 0.0f;
            uint32_t bx = 0, by = 0;
            for (uint32_t y = y0; y < y1; ++y) {
                for (uint32_t x = x0; x < x1; ++x) {
                    float s = score[(size_t)y * w + x];
                    if (!(s >= threshold) || s == -INFINITY) continue;
                    if (mask && mask[(size_t)y * mask_pitch + x] == 0) continue;
                    int is_max = 1;
                    for (int dy = -1; dy <= 1 && is_max; ++dy) {
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (!dx && !dy) continue;
                            int qx = (int)x + dx, qy = (int)y + dy;
                            if (qx < 0 || qy < 0 || qx >= (int)w || qy >= (int)h) continue;
                            float q = score[(size_t)qy * w + qx];
                            int q_first = (dy < 0) || (dy == 0 && dx < 0);
                            if (q > s || (q == s && q_first)) { is_max = 0; break; }
                        }
                    }
                    if (!is_max) continue;
                    if (!found || s > best) { found = 1; best = s; bx = x; by = y; }
                }
            }
            cfound[ci] = (uint8_t)found;
            cbest[ci] = best;
            cpos[2 * ci] = bx;
            cpos[2 * ci + 1] = by;
        }
    }
    for (size_t ci = 0; ci < ncells; ++ci) {
        if (!cfound[ci]) continue;
        float best = cbest[ci];
        uint32_t bx = cpos[2 * ci], by = cpos[2 * ci + 1];
        if (nout < capacity) {
            if (item_type == ORC_ITEM_POINT2F) {
                orc_point2f_t *o = (orc_point2f_t *)out + nout;
                o->x = (float)bx; o->y = (float)by;
            } else {
                orc_keypointf_t *o = (orc_keypointf_t *)out + nout;
                o->x = (float)bx; o->y = (float)by;
                o->strength = best; o->scale = 0.0f; o->orientation = 0.0f;
                o->tracking_status = 1; o->error = 0.0f;
            }
            if (strength_out) strength_out[nout] = best;
            ++nout;
        }
        ++total;
    }
    free(cfound);
    free(cbest);
    free(cpos);
    free(occupied);
    if (num_corners) *num_corners = total;
    return nout;
}

uint32_t orc_harris_track(const uint8_t *img, uint32_t pitch, uint32_t w, uint32_t h,
                          const uint8_t *mask, uint32_t mask_pitch,
                          int item_type, const void *tracked, uint32_t n_tracked,
                          float k, float threshold, uint32_t cell_size,
                          void *out, uint32_t capacity, size_t *num_corners,
                          float *strength_out)
{
    float *resp = (float *)malloc((size_t)w * h * sizeof(float));
    orc_harris_response(img, pitch, w, h, k, resp);
    uint32_t n = select_and_merge(resp, w, h, mask, mask_pitch, item_type, tracked, n_tracked,
                                  threshold, cell_size, out, capacity, num_corners, strength_out);
    free(resp);
    return n;
}

uint32_t orc_fast_track(const uint8_t *img, uint32_t pitch, uint32_t w, uint32_t h,
                        const uint8_t *mask, uint32_t mask_pitch,
                        int item_type, const void *tracked, uint32_t n_tracked,
                        uint32_t type_n, uint32_t threshold, uint32_t cell_size,
                        void *out, uint32_t capacity, size_t *num_corners,
                        float *strength_out)
{
    size_t n = (size_t)w * h;
    int32_t *st = (int32_t *)malloc(n * sizeof(int32_t));
    float *score = (float *)malloc(n * sizeof(float));
    orc_fast_strength(img, pitch, w, h, type_n, st);
    for (size_t i = 0; i < n; ++i) score[i] = st[i] < 0 ? -INFINITY : (float)st[i];
    uint32_t r = select_and_merge(score, w, h, mask, mask_pitch, item_type, tracked, n_tracked,
                                  (float)threshold, cell_size, out, capacity, num_corners,
                                  strength_out);
    free(st);
    free(score);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Pyramidal Lucas-Kanade.  Call sites: demos/feature_tracker/feature_tracker.cpp:402-412
 * (VX_TERM_CRITERIA_BOTH, eps 0.01, use_initial_estimate false), feature_tracker_nvxcu.cpp:
 * 297-301, object_tracker_keypoint_nvxcu.cpp:1791-1828.
 * [S] per level, top to 0: G = sum over the window of [Ix^2 IxIy; IxIy Iy^2] with Scharr 3x3
 *     gradients of the previous image, iterate v += G^-1 * sum((I - J)[Ix Iy]) with bilinear
 *     sampling, stop on |eta| < eps and/or the iteration count, propagate g <- 2 (g + v).
 * [R] lost POINT2F = (-1,-1) (nvxio/src/OpenGLShaders.hpp:635-641, sfm/SfM.cpp:1535);
 *     KEYPOINTF tracked <=> tracking_status == 255 (object_tracker_keypoint_nvxcu.cpp:1887),
 *     strength carried through (:1912).
 * [P] window offsets -win/2 .. win-1-win/2; coordinates clamp to the image (replicate).
 *     Bilinear sampling interpolates along y first, then along x (each an fmaf of the difference).
 *     The template is resampled at the feature's sub-pixel offset on the window plus a one-pixel
 *     ring, and its gradients are the Scharr 3x3 derivative (normalised by 1/32) of that resampled
 *     patch -- Bouguet's formulation; interpolating derivatives of the integer image is the same
 *     linear operator in exact arithmetic.  A level whose window is flat (min-eigenvalue/win^2 < 1e-4 or
 *     det < FLT_EPSILON) is skipped, and the point is lost when that level is 0; a point whose
 *     window centre strays more than one window outside a level is lost; a point whose final
 *     position leaves [0,w-1]x[0,h-1] is lost; inputs already lost stay lost;
 *     |eta| < eps is tested as eta.x^2 + eta.y^2 < eps^2; EPSILON-only termination is capped at
 *     100 iterations.
 *     calcError (thin layer, object_tracker_keypoint_nvxcu.cpp:1799,1824 pass 0): when set, a tracked KEYPOINTF item's
 *     `error` is the mean absolute difference between the level-0 template and the new image sampled at the final
 *     position, sum(|I - J|) / win^2 (same sampling, same summation order); 0 for lost points and when not requested.
 * Summation order (ORC_SUM_WAVE64, canonical): window pixels are dealt to 64 accumulators
 * (row = l / segs, segment = l % segs, seglen consecutive pixels each, ascending x), each
 * accumulating by fmaf in order, then combined by a pairwise butterfly (offsets 1,2,4,..,32).
 * ---------------------------------------------------------------------------------------- */
typedef struct { int segs, seglen, lanes; } lk_map_t;

static lk_map_t lk_map(int win)
{
    lk_map_t m;
    int segs = 64 / win;
    if (segs > win) segs = win;
    if (segs < 1) segs = 1;
    m.seglen = (win + segs - 1) / segs;
    m.segs = (win + m.seglen - 1) / m.seglen;
    m.lanes = m.segs * win;
    return m;
}

static inline float tree64(float *p)
{
    float t[64];
    for (int off = 1; off < 64; off <<= 1) {
        for (int l = 0; l < 64; ++l) t[l] = p[l] + p[l ^ off];
        memcpy(p, t, sizeof(t));
    }
    return p[0];
}

static inline float pix(const orc_image_t *im, int x, int y)
{
    x = clampi(x, 0, (int)im->width - 1);
    y = clampi(y, 0, (int)im->height - 1);
    return (float)im->data[(size_t)y * im->pitch + x];
}

static inline float bilerp(float a, float b, float c, float d, float fx, float fy)
{   /* a b / c d = top-left, top-right / bottom-left, bottom-right */
    float left = fmaf(fy, c - a, a);
    float right = fmaf(fy, d - b, b);
    return fmaf(fx, right - left, left);
}

#define LK_MAX_WIN 32

static int lk_track_one(const orc_image_t *prev, const orc_image_t *curr, uint32_t num_levels,
                        float px, float py, float ex, float ey,
                        int termination, float epsilon, uint32_t num_iterations, int win,
                        int sum_mode, float *ox, float *oy, float *err)
{
    const uint32_t w0 = prev[0].width, h0 = prev[0].height;
    if (!(px >= 0.0f && py >= 0.0f && px <= (float)(w0 - 1) && py <= (float)(h0 - 1))) return 0;

    const int half = win / 2;
    const lk_map_t map = lk_map(win);
    const int use_eps = termination != ORC_TERM_ITERATIONS;
    uint32_t max_it = num_iterations;
    if (termination == ORC_TERM_EPSILON) max_it = 100;
    const float eps2 = epsilon * epsilon;
    const float inv_area = 1.0f / (float)(2 * win * win);

    float Iw[LK_MAX_WIN * LK_MAX_WIN], Ix[LK_MAX_WIN * LK_MAX_WIN], Iy[LK_MAX_WIN * LK_MAX_WIN];
    float T[(LK_MAX_WIN + 2) * (LK_MAX_WIN + 2)];   /* resampled template: window + one-pixel ring */
    const int tw = win + 2;

    const int top = (int)num_levels - 1;
    float s_top = 1.0f / (float)(1u << top);
    float gx = (ex - px) * s_top, gy = (ey - py) * s_top;
    float vx = 0.0f, vy = 0.0f;

    for (int L = top; L >= 0; --L) {
        const orc_image_t *I = &prev[L], *J = &curr[L];
        const float s = 1.0f / (float)(1u << L);
        const float ux = px * s, uy = py * s;
        const float fx0 = floorf(ux), fy0 = floorf(uy);
        const float ax = ux - fx0, ay = uy - fy0;
        const int ix = (int)fx0 - half, iy = (int)fy0 - half;
        vx = 0.0f; vy = 0.0f;

        /* template: resample window + ring, then Scharr/32 of the resampled patch.  Separable form
         * with 3/32 and 10/32 folded in: s = 10/32*mid + 3/32*(up + down) per column, d = down - up,
         * Ix = s[x+1] - s[x-1], Iy = 10/32*d[x] + 3/32*(d[x-1] + d[x+1]). */
        for (int ty = 0; ty < tw; ++ty)
            for (int tx = 0; tx < tw; ++tx) {
                int x = ix + tx - 1, y = iy + ty - 1;
                T[ty * tw + tx] = bilerp(pix(I, x, y), pix(I, x + 1, y), pix(I, x, y + 1), pix(I, x + 1, y + 1), ax, ay);
            }
        for (int wy = 0; wy < win; ++wy)
            for (int wx = 0; wx < win; ++wx) {
                const float *up = &T[wy * tw + wx], *mid = up + tw, *dn = mid + tw;   /* [0..2] = x-1, x, x+1 */
                const float sl = fmaf(0.3125f, mid[0], 0.09375f * (up[0] + dn[0]));
                const float sr = fmaf(0.3125f, mid[2], 0.09375f * (up[2] + dn[2]));
                const float dl = dn[0] - up[0], dc = dn[1] - up[1], dr = dn[2] - up[2];
                int j = wy * win + wx;
                Iw[j] = mid[1];
                Ix[j] = sr - sl;
                Iy[j] = fmaf(0.3125f, dc, 0.09375f * (dl + dr));
            }

        float Gxx, Gxy, Gyy;
        if (sum_mode == ORC_SUM_WAVE64) {
            float pxx[64] = { 0 }, pxy[64] = { 0 }, pyy[64] = { 0 };
            for (int l = 0; l < map.lanes; ++l) {
                int row = l / map.segs, x0 = (l % map.segs) * map.seglen;
                float a = 0.0f, b = 0.0f, c = 0.0f;
                for (int k = 0; k < map.seglen && x0 + k < win; ++k) {
                    int j = row * win + x0 + k;
                    a = fmaf(Ix[j], Ix[j], a);
                    b = fmaf(Ix[j], Iy[j], b);
                    c = fmaf(Iy[j], Iy[j], c);
                }
                pxx[l] = a; pxy[l] = b; pyy[l] = c;
            }
            Gxx = tree64(pxx); Gxy = tree64(pxy); Gyy = tree64(pyy);
        } else {
            double a = 0, b = 0, c = 0;
            for (int j = 0; j < win * win; ++j) {
                a += (double)Ix[j] * Ix[j];
                b += (double)Ix[j] * Iy[j];
                c += (double)Iy[j] * Iy[j];
            }
            Gxx = (float)a; Gxy = (float)b; Gyy = (float)c;
        }

        float det = Gxx * Gyy - Gxy * Gxy;
        float dxy = Gxx - Gyy;
        float disc = sqrtf(dxy * dxy + 4.0f * (Gxy * Gxy));
        float min_eig = ((Gyy + Gxx) - disc) * inv_area;
        if (!(min_eig >= 1e-4f) || !(det >= FLT_EPSILON)) {
            if (L == 0) return 0;
            gx = 2.0f * gx; gy = 2.0f * gy;
            continue;
        }
        const float inv_det = 1.0f / det;
        const float lw = (float)J->width, lh = (float)J->height, fw = (float)win;

        for (uint32_t it = 0; it < max_it; ++it) {
            const float cx = (ux + gx) + vx, cy = (uy + gy) + vy;
            if (!(cx > -fw && cy > -fw && cx < lw + fw && cy < lh + fw)) return 0;
            const float jfx = floorf(cx), jfy = floorf(cy);
            const float bx_ = cx - jfx, by_ = cy - jfy;
            const int jx = (int)jfx - half, jy = (int)jfy - half;
            float b1, b2;
            if (sum_mode == ORC_SUM_WAVE64) {
                float p1[64] = { 0 }, p2[64] = { 0 };
                for (int l = 0; l < map.lanes; ++l) {
                    int row = l / map.segs, x0 = (l % map.segs) * map.seglen;
                    float a = 0.0f, b = 0.0f;
                    for (int k = 0; k < map.seglen && x0 + k < win; ++k) {
                        int wx = x0 + k, j = row * win + wx;
                        int x = jx + wx, y = jy + row;
                        float jv = bilerp(pix(J, x, y), pix(J, x + 1, y), pix(J, x, y + 1),
                                          pix(J, x + 1, y + 1), bx_, by_);
                        float d = Iw[j] - jv;
                        a = fmaf(d, Ix[j], a);
                        b = fmaf(d, Iy[j], b);
                    }
                    p1[l] = a; p2[l] = b;
                }
                b1 = tree64(p1); b2 = tree64(p2);
            } else {
                double a = 0, b = 0;
                for (int wy = 0; wy < win; ++wy)
                    for (int wx = 0; wx < win; ++wx) {
                        int j = wy * win + wx, x = jx + wx, y = jy + wy;
                        float jv = bilerp(pix(J, x, y), pix(J, x + 1, y), pix(J, x, y + 1),
                                          pix(J, x + 1, y + 1), bx_, by_);
                        double d = (double)Iw[j] - jv;
                        a += d * Ix[j];
                        b += d * Iy[j];
                    }
                b1 = (float)a; b2 = (float)b;
            }
            const float ex_ = (Gyy * b1 - Gxy * b2) * inv_det;
            const float ey_ = (Gxx * b2 - Gxy * b1) * inv_det;
            vx += ex_; vy += ey_;
            if (use_eps && fmaf(ex_, ex_, ey_ * ey_) < eps2) break;
        }
        if (L > 0) { gx = 2.0f * (gx + vx); gy = 2.0f * (gy + vy); }
    }
    const float rx = px + (gx + vx), ry = py + (gy + vy);
    if (!(rx >= 0.0f && ry >= 0.0f && rx <= (float)(w0 - 1) && ry <= (float)(h0 - 1))) return 0;
    *ox = rx; *oy = ry;
    if (err) {   /* level 0: Iw still holds its template; (px + gx) + vx is the position an iteration would sample */
        const orc_image_t *J = &curr[0];
        const float cx = (px + gx) + vx, cy = (py + gy) + vy;
        const float jfx = floorf(cx), jfy = floorf(cy);
        const float bx_ = cx - jfx, by_ = cy - jfy;
        const int jx = (int)jfx - half, jy = (int)jfy - half;
        float total;
        if (sum_mode == ORC_SUM_WAVE64) {
            float pe[64] = { 0 };
            for (int l = 0; l < map.lanes; ++l) {
                int row = l / map.segs, x0 = (l % map.segs) * map.seglen;
                float a = 0.0f;
                for (int k = 0; k < map.seglen && x0 + k < win; ++k) {
                    int wx = x0 + k, j = row * win + wx, x = jx + wx, y = jy + row;
                    float jv = bilerp(pix(J, x, y), pix(J, x + 1, y), pix(J, x, y + 1), pix(J, x + 1, y + 1), bx_, by_);
                    a = a + fabsf(Iw[j] - jv);
                }
                pe[l] = a;
            }
            total = tree64(pe);
        } else {
            double a = 0;
            for (int wy = 0; wy < win; ++wy)
                for (int wx = 0; wx < win; ++wx) {
                    int j = wy * win + wx, x = jx + wx, y = jy + wy;
                    float jv = bilerp(pix(J, x, y), pix(J, x + 1, y), pix(J, x, y + 1), pix(J, x + 1, y + 1), bx_, by_);
                    a += fabs((double)Iw[j] - jv);
                }
            total = (float)a;
        }
        *err = total * (1.0f / (float)(win * win));
    }
    return 1;
}

void orc_optical_flow_pyr_lk(const orc_image_t *prev_levels, const orc_image_t *curr_levels,
                             uint32_t num_levels,
                             int item_type, const void *prev_pts, void *pts_inout, uint32_t n,
                             int termination, float epsilon, uint32_t num_iterations,
                             uint32_t window, int sum_mode, int calc_error)
{
    if (window < 3 || window > LK_MAX_WIN || num_levels < 1) return;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (uint32_t i = 0; i < n; ++i) {
        float ox = -1.0f, oy = -1.0f;
        if (item_type == ORC_ITEM_POINT2F) {
            const orc_point2f_t *p = (const orc_point2f_t *)prev_pts + i;
            orc_point2f_t *q = (orc_point2f_t *)pts_inout + i;
            int ok = lk_track_one(prev_levels, curr_levels, num_levels, p->x, p->y, q->x, q->y,
                                  termination, epsilon, num_iterations, (int)window, sum_mode, &ox, &oy, NULL);
            if (ok) { q->x = ox; q->y = oy; } else { q->x = -1.0f; q->y = -1.0f; }
        } else {
            const orc_keypointf_t *p = (const orc_keypointf_t *)prev_pts + i;
            orc_keypointf_t *q = (orc_keypointf_t *)pts_inout + i;
            float ex = q->x, ey = q->y, err = 0.0f;
            int ok = 0;
            if (p->tracking_status != 0)
                ok = lk_track_one(prev_levels, curr_levels, num_levels, p->x, p->y, ex, ey,
                                  termination, epsilon, num_iterations, (int)window, sum_mode, &ox, &oy,
                                  calc_error ? &err : NULL);
            q->strength = p->strength;
            q->scale = p->scale;
            q->orientation = p->orientation;
            q->error = ok ? err : 0.0f;
            if (ok) { q->x = ox; q->y = oy; q->tracking_status = 255; }
            else    { q->x = ex; q->y = ey; q->tracking_status = 0; }
        }
    }
}
