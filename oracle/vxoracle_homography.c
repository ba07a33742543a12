/* vxoracle_homography.c -- CPU restatement of nvxFindHomographyNode (RANSAC) for the parity tests.
 *
 * TEST INFRASTRUCTURE ONLY (see vxoracle.h).  PARITY UNPINNED: the reference implementation lives in
 * closed libvisionworks; /root/reference holds its call sites only:
 *   demos/video_stabilizer/stabilizer.cpp:264-273   (RANSAC, threshold 3.0, 2000 estimate iterations,
 *       10 refine iterations, confidence 0.995, outlier ratio 0.45, U8 mask array out)
 *   VisionWorks-SFM-0.90-Samples/samples/sfm/SfM.cpp:1576-1578  (same, threshold 0.5)
 * and the consumer of its outputs, demos/video_stabilizer/homography_filter_node.cpp:36-159:
 *   [R] the matrix is 3x3 FLOAT32 and is TRANSPOSED before use (x' ~ M^T-as-read applied to x), i.e. element
 *       [3*c + r] of the matrix data is H[r][c];  the mask holds one U8 per input point, non-zero = inlier.
 * Everything else is frozen policy [P]:
 *   * a correspondence is usable when all four coordinates are >= 0 (PyrLK marks lost points (-1,-1));
 *   * the number of hypotheses is the a-priori RANSAC count for (confidence, outlier ratio),
 *     ceil(log(1-confidence) / log(1-(1-outlier_ratio)^4)), capped by the estimate-iteration limit -- all
 *     hypotheses are independent and evaluated as one batch;
 *   * hypothesis k draws 4 distinct usable correspondences with the LCG below (seeded by k), and maps them
 *     exactly (Heckbert's square-to-quadrilateral construction, binary64);
 *   * inliers: reprojection error <= threshold, evaluated in binary32 with the hypothesis rounded to binary32
 *     (h22 = 1) and without a division: with (X, Y, W) = H (x, y, 1), the point is an inlier when W > 1e-6 and
 *     (X - uW)^2 + (Y - vW)^2 <= threshold^2 W^2;  best = most inliers, ties to the lowest k;
 *   * refinement, at most `max_refine_iters` times: least-squares homography (h22 = 1, normal equations in
 *     binary64 on coordinates normalised once over all usable points) over the current inliers; accepted
 *     while the inlier count does not drop; stops when the inlier set stops changing;
 *   * fewer than 4 usable points or no valid hypothesis: identity, empty mask.
 * Summation order of every floating-point reduction: 256 interleaved partial sums (partial t takes points
 * t, t+256, ... in order) combined by a pairwise butterfly (offsets 1,2,..,128).
 */
#include "vxoracle.h"

#include <math.h>
#include <string.h>

#define HG_P 256

uint32_t orc_homography_hypotheses(float confidence, float outlier_ratio, uint32_t max_estimate_iters)
{
    double w = 1.0 - (double)outlier_ratio;
    if (!(w >= 0.01)) w = 0.01;
    if (w > 1.0) w = 1.0;
    double c = (double)confidence;
    if (!(c >= 0.0)) c = 0.0;
    if (c > 0.999999) c = 0.999999;
    const double p = w * w * w * w;
    double k = 1.0;
    if (p < 1.0) k = ceil(log(1.0 - c) / log(1.0 - p));
    if (!(k >= 1.0)) k = 1.0;
    if (max_estimate_iters < 1) max_estimate_iters = 1;
    if (k > (double)max_estimate_iters) k = (double)max_estimate_iters;
    return (uint32_t)k;
}

static inline int hg_usable(const float *s, const float *d, uint32_t i)
{
    return s[2 * i] >= 0.0f && s[2 * i + 1] >= 0.0f && d[2 * i] >= 0.0f && d[2 * i + 1] >= 0.0f;
}

static inline uint32_t hg_draw(uint32_t *s, uint32_t n)
{
    *s = *s * 1664525u + 1013904223u;
    return (uint32_t)(((uint64_t)(*s >> 8) * n) >> 24);
}

static int hg_sample(const float *src, const float *dst, uint32_t n, uint32_t k, uint32_t idx[4])
{
    uint32_t s = (k + 1u) * 0x9E3779B9u;
    for (int j = 0; j < 4; ++j) {
        int ok = 0;
        for (int t = 0; t < 64 && !ok; ++t) {
            const uint32_t i = hg_draw(&s, n);
            if (!hg_usable(src, dst, i)) continue;
            int dup = 0;
            for (int q = 0; q < j; ++q) dup |= idx[q] == i;
            if (dup) continue;
            idx[j] = i;
            ok = 1;
        }
        if (!ok) return 0;
    }
    return 1;
}

/* unit square (0,0),(1,0),(1,1),(0,1) -> quadrilateral (x[i],y[i]) */
static int hg_square_to_quad(const double x[4], const double y[4], double A[9])
{
    const double dx1 = x[1] - x[2], dx2 = x[3] - x[2], dy1 = y[1] - y[2], dy2 = y[3] - y[2];
    const double sx = ((x[0] - x[1]) + x[2]) - x[3], sy = ((y[0] - y[1]) + y[2]) - y[3];
    const double den = dx1 * dy2 - dx2 * dy1;
    if (!(den != 0.0)) return 0;
    const double g = (sx * dy2 - dx2 * sy) / den, h = (dx1 * sy - sx * dy1) / den;
    A[0] = (x[1] - x[0]) + g * x[1]; A[1] = (x[3] - x[0]) + h * x[3]; A[2] = x[0];
    A[3] = (y[1] - y[0]) + g * y[1]; A[4] = (y[3] - y[0]) + h * y[3]; A[5] = y[0];
    A[6] = g; A[7] = h; A[8] = 1.0;
    return 1;
}

/* H (binary64, any scale) -> binary32 with h22 = 1; 0 if it cannot be normalised */
static int hg_to_f32(const double H[9], float Hf[9])
{
    double m = 0.0;
    for (int i = 0; i < 9; ++i) { const double a = fabs(H[i]); if (a > m) m = a; }
    if (!(fabs(H[8]) > 1e-9 * m) || !(m < INFINITY)) return 0;
    for (int i = 0; i < 9; ++i) Hf[i] = (float)(H[i] / H[8]);
    return 1;
}

static int hg_hypothesis(const float *src, const float *dst, const uint32_t idx[4], float Hf[9])
{
    double x[4], y[4], u[4], v[4], A[9], B[9], adj[9], H[9];
    for (int j = 0; j < 4; ++j) {
        x[j] = src[2 * idx[j]]; y[j] = src[2 * idx[j] + 1];
        u[j] = dst[2 * idx[j]]; v[j] = dst[2 * idx[j] + 1];
    }
    if (!hg_square_to_quad(x, y, A) || !hg_square_to_quad(u, v, B)) return 0;
    adj[0] = A[4] * A[8] - A[5] * A[7]; adj[1] = A[2] * A[7] - A[1] * A[8]; adj[2] = A[1] * A[5] - A[2] * A[4];
    adj[3] = A[5] * A[6] - A[3] * A[8]; adj[4] = A[0] * A[8] - A[2] * A[6]; adj[5] = A[2] * A[3] - A[0] * A[5];
    adj[6] = A[3] * A[7] - A[4] * A[6]; adj[7] = A[1] * A[6] - A[0] * A[7]; adj[8] = A[0] * A[4] - A[1] * A[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            H[3 * r + c] = (B[3 * r] * adj[c] + B[3 * r + 1] * adj[3 + c]) + B[3 * r + 2] * adj[6 + c];
    return hg_to_f32(H, Hf);
}

static inline int hg_inlier(const float *H, float x, float y, float u, float v, float thr2)
{
    const float X = fmaf(H[0], x, fmaf(H[1], y, H[2]));
    const float Y = fmaf(H[3], x, fmaf(H[4], y, H[5]));
    const float W = fmaf(H[6], x, fmaf(H[7], y, H[8]));
    const float ex = fmaf(-u, W, X), ey = fmaf(-v, W, Y);
    return (W > 1e-6f) & (fmaf(ex, ex, ey * ey) <= thr2 * (W * W));
}

static double hg_tree(double *p)
{
    double t[HG_P];
    for (int off = 1; off < HG_P; off <<= 1) {
        for (int l = 0; l < HG_P; ++l) t[l] = p[l] + p[l ^ off];
        memcpy(p, t, sizeof(t));
    }
    return p[0];
}

typedef struct { double cx, cy, ss, cu, cv, sd; int ok; } hg_norm_t;

static hg_norm_t hg_normalisation(const float *src, const float *dst, uint32_t n, uint32_t n_usable)
{
    hg_norm_t nm;
    memset(&nm, 0, sizeof(nm));
    double p[4][HG_P];
    {
        memset(p, 0, sizeof(p));
        for (uint32_t i = 0; i < n; ++i)
            if (hg_usable(src, dst, i)) {
                const int t = (int)(i % HG_P);
                p[0][t] += (double)src[2 * i]; p[1][t] += (double)src[2 * i + 1];
                p[2][t] += (double)dst[2 * i]; p[3][t] += (double)dst[2 * i + 1];
            }
        const double cnt = (double)n_usable;
        nm.cx = hg_tree(p[0]) / cnt; nm.cy = hg_tree(p[1]) / cnt;
        nm.cu = hg_tree(p[2]) / cnt; nm.cv = hg_tree(p[3]) / cnt;
        memset(p, 0, sizeof(p));
        for (uint32_t i = 0; i < n; ++i)
            if (hg_usable(src, dst, i)) {
                const int t = (int)(i % HG_P);
                p[0][t] += fabs((double)src[2 * i] - nm.cx) + fabs((double)src[2 * i + 1] - nm.cy);
                p[1][t] += fabs((double)dst[2 * i] - nm.cu) + fabs((double)dst[2 * i + 1] - nm.cv);
            }
        const double ds = hg_tree(p[0]), dd = hg_tree(p[1]);
        nm.ok = ds > 0.0 && dd > 0.0;
        if (nm.ok) { nm.ss = cnt / ds; nm.sd = cnt / dd; }
    }
    return nm;
}

#define HG_NACC 29

/* least-squares homography (h22 = 1) over the points flagged in mask; 0 when the system is not positive definite */
static int hg_fit(const float *src, const float *dst, uint32_t n, const uint8_t *mask, const hg_norm_t *nm, float Hf[9])
{
    double part[HG_NACC][HG_P];
    double S[HG_NACC];
    int result = 0;
    {
        memset(part, 0, sizeof(part));
        for (uint32_t i = 0; i < n; ++i) {
            if (!mask[i]) continue;
            const int t = (int)(i % HG_P);
            const double x = ((double)src[2 * i] - nm->cx) * nm->ss, y = ((double)src[2 * i + 1] - nm->cy) * nm->ss;
            const double u = ((double)dst[2 * i] - nm->cu) * nm->sd, v = ((double)dst[2 * i + 1] - nm->cv) * nm->sd;
            const double q0 = -(x * u), q1 = -(y * u), r0 = -(x * v), r1 = -(y * v);
#define ACC(k, a, b) part[k][t] = fma(a, b, part[k][t])
            ACC(0, x, x); ACC(1, x, y); ACC(2, x, 1.0); ACC(3, y, y); ACC(4, y, 1.0); ACC(5, 1.0, 1.0);
            ACC(6, x, q0); ACC(7, x, q1); ACC(8, y, q0); ACC(9, y, q1); ACC(10, 1.0, q0); ACC(11, 1.0, q1);
            ACC(12, x, r0); ACC(13, x, r1); ACC(14, y, r0); ACC(15, y, r1); ACC(16, 1.0, r0); ACC(17, 1.0, r1);
            ACC(18, q0, q0); ACC(18, r0, r0); ACC(19, q0, q1); ACC(19, r0, r1); ACC(20, q1, q1); ACC(20, r1, r1);
            ACC(21, x, u); ACC(22, y, u); ACC(23, 1.0, u);
            ACC(24, x, v); ACC(25, y, v); ACC(26, 1.0, v);
            ACC(27, q0, u); ACC(27, r0, v); ACC(28, q1, u); ACC(28, r1, v);
#undef ACC
        }
        for (int k = 0; k < HG_NACC; ++k) S[k] = hg_tree(part[k]);

        /* normal equations M h = b, M symmetric 8x8 */
        double M[8][8], b[8], h[8];
        memset(M, 0, sizeof(M));
        const double pp[3][3] = { { S[0], S[1], S[2] }, { S[1], S[3], S[4] }, { S[2], S[4], S[5] } };
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { M[i][j] = pp[i][j]; M[3 + i][3 + j] = pp[i][j]; }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j) {
                M[i][6 + j] = M[6 + j][i] = S[6 + 2 * i + j];
                M[3 + i][6 + j] = M[6 + j][3 + i] = S[12 + 2 * i + j];
            }
        M[6][6] = S[18]; M[6][7] = M[7][6] = S[19]; M[7][7] = S[20];
        b[0] = S[21]; b[1] = S[22]; b[2] = S[23]; b[3] = S[24]; b[4] = S[25]; b[5] = S[26]; b[6] = S[27]; b[7] = S[28];

        /* LDL^T without pivoting (M is positive definite for a non-degenerate inlier set) */
        double Lm[8][8], D[8];
        int pd = 1;
        memset(Lm, 0, sizeof(Lm));
        for (int j = 0; j < 8 && pd; ++j) {
            double d = M[j][j];
            for (int k = 0; k < j; ++k) d -= (Lm[j][k] * Lm[j][k]) * D[k];
            if (!(d > 1e-12)) { pd = 0; break; }
            D[j] = d;
            for (int i = j + 1; i < 8; ++i) {
                double s = M[i][j];
                for (int k = 0; k < j; ++k) s -= (Lm[i][k] * Lm[j][k]) * D[k];
                Lm[i][j] = s / d;
            }
        }
        if (pd) {
            double z[8];
            for (int i = 0; i < 8; ++i) {
                double s = b[i];
                for (int k = 0; k < i; ++k) s -= Lm[i][k] * z[k];
                z[i] = s;
            }
            for (int i = 7; i >= 0; --i) {
                double s = z[i] / D[i];
                for (int k = i + 1; k < 8; ++k) s -= Lm[k][i] * h[k];
                h[i] = s;
            }
            /* denormalise: H = Tdst^-1 * Hn * Tsrc */
            const double Hn[9] = { h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 1.0 };
            double G[9], H[9];
            for (int r = 0; r < 3; ++r) {
                G[3 * r] = Hn[3 * r] * nm->ss;
                G[3 * r + 1] = Hn[3 * r + 1] * nm->ss;
                G[3 * r + 2] = (Hn[3 * r + 2] - G[3 * r] * nm->cx) - G[3 * r + 1] * nm->cy;
            }
            for (int c = 0; c < 3; ++c) {
                H[c] = G[c] / nm->sd + nm->cu * G[6 + c];
                H[3 + c] = G[3 + c] / nm->sd + nm->cv * G[6 + c];
                H[6 + c] = G[6 + c];
            }
            result = hg_to_f32(H, Hf);
        }
    }
    return result;
}

static uint32_t hg_count(const float *src, const float *dst, uint32_t n, const float *H, float thr2,
                         const uint8_t *old_mask, uint8_t *new_mask, uint32_t *changed)
{
    uint32_t cnt = 0, diff = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const int in = hg_usable(src, dst, i) && hg_inlier(H, src[2 * i], src[2 * i + 1], dst[2 * i], dst[2 * i + 1], thr2);
        cnt += (uint32_t)in;
        if (old_mask) diff += (uint32_t)((old_mask[i] != 0) != in);
        if (new_mask) new_mask[i] = (uint8_t)(in ? 255 : 0);
    }
    if (changed) *changed = diff;
    return cnt;
}

uint32_t orc_find_homography(const float *src, const float *dst, uint32_t n, float threshold,
                             uint32_t max_estimate_iters, uint32_t max_refine_iters, float confidence,
                             float outlier_ratio, float *H_out, uint8_t *mask)
{
    static const float eye[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    float H[9];
    memcpy(H, eye, sizeof(H));
    if (mask && n) memset(mask, 0, n);
    uint32_t n_usable = 0;
    for (uint32_t i = 0; i < n; ++i) n_usable += (uint32_t)hg_usable(src, dst, i);
    const float thr2 = threshold * threshold;
    uint32_t best_cnt = 0;
    int have = 0;
    if (n_usable >= 4) {
        const uint32_t K = orc_homography_hypotheses(confidence, outlier_ratio, max_estimate_iters);
        for (uint32_t k = 0; k < K; ++k) {
            uint32_t idx[4];
            float Hk[9];
            if (!hg_sample(src, dst, n, k, idx) || !hg_hypothesis(src, dst, idx, Hk)) continue;
            const uint32_t c = hg_count(src, dst, n, Hk, thr2, NULL, NULL, NULL);
            if (!have || c > best_cnt) { have = 1; best_cnt = c; memcpy(H, Hk, sizeof(H)); }
        }
    }
    if (!have || best_cnt < 4) {
        memcpy(H, eye, sizeof(H));
        best_cnt = 0;
    } else {
        uint8_t *cur = mask;   /* working inlier flags */
        hg_count(src, dst, n, H, thr2, NULL, cur, NULL);
        const hg_norm_t nm = hg_normalisation(src, dst, n, n_usable);
        for (uint32_t r = 0; r < max_refine_iters && nm.ok; ++r) {
            float Hn[9];
            if (!hg_fit(src, dst, n, cur, &nm, Hn)) break;
            uint32_t diff = 0;
            const uint32_t c = hg_count(src, dst, n, Hn, thr2, cur, NULL, &diff);
            if (c < best_cnt) break;
            memcpy(H, Hn, sizeof(H));
            best_cnt = c;
            hg_count(src, dst, n, H, thr2, NULL, cur, NULL);
            if (diff == 0) break;
        }
    }
    /* the reference's consumers transpose the matrix before use (homography_filter_node.cpp:87-91) */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H_out[3 * c + r] = H[3 * r + c];
    return best_cnt;
}

/* ------------------------------------------------------------------------------------------
 * vxWarpPerspectiveNode.  Call site: demos/video_stabilizer/stabilizer.cpp:292-295 (RGBX frame, BILINEAR, default
 * border), matrix produced by truncate_transform_node.cpp:163-165.
 * [S] output(x,y) = input(x0/z0, y0/z0), (x0,y0,z0) = M (x,y,1); `m` here is M row-major, i.e. m[3r+c] is element
 *     [3c+r] of the vx_matrix data (OpenVX keeps the matrix column by column).
 * [P] binary32: X = fma(m0,x,fma(m1,y,m2)) etc., xs = X/Z, ys = Y/Z (IEEE division);
 *     BILINEAR: sampled where 0 <= xs <= w-1 and 0 <= ys <= h-1 (anything else, NaN included, takes the border value);
 *       taps (floor, min(floor+1, size-1)); per channel top = fma(fx, p01-p00, p00), bot likewise,
 *       v = fma(fy, bot-top, top), stored as (int)(v + 0.5);
 *     NEAREST_NEIGHBOR: tap (floor(xs+0.5), floor(ys+0.5)) where that lies inside the image;
 *     border: the constant `border` pixel (VX_BORDER_UNDEFINED is refined to constant 0).
 */
void orc_warp_perspective(const uint8_t *src, uint32_t src_pitch, uint32_t sw, uint32_t sh, uint32_t channels,
                          uint8_t *dst, uint32_t dst_pitch, uint32_t dw, uint32_t dh,
                          const float *m, int bilinear, const uint8_t *border)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (uint32_t y = 0; y < dh; ++y) {
        uint8_t *o = dst + (size_t)y * dst_pitch;
        for (uint32_t x = 0; x < dw; ++x) {
            const float xf = (float)x, yf = (float)y;
            const float X = fmaf(m[0], xf, fmaf(m[1], yf, m[2]));
            const float Y = fmaf(m[3], xf, fmaf(m[4], yf, m[5]));
            const float Z = fmaf(m[6], xf, fmaf(m[7], yf, m[8]));
            const float xs = X / Z, ys = Y / Z;
            for (uint32_t c = 0; c < channels; ++c) o[channels * x + c] = border[c];
            if (bilinear) {
                if (!(xs >= 0.0f && ys >= 0.0f && xs <= (float)(sw - 1) && ys <= (float)(sh - 1))) continue;
                const float x0f = floorf(xs), y0f = floorf(ys);
                const float fx = xs - x0f, fy = ys - y0f;
                const uint32_t x0 = (uint32_t)x0f, y0 = (uint32_t)y0f;
                const uint32_t x1 = x0 + 1 < sw ? x0 + 1 : sw - 1, y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
                const uint8_t *r0 = src + (size_t)y0 * src_pitch, *r1 = src + (size_t)y1 * src_pitch;
                for (uint32_t c = 0; c < channels; ++c) {
                    const float a = r0[channels * x0 + c], b = r0[channels * x1 + c];
                    const float cc = r1[channels * x0 + c], d = r1[channels * x1 + c];
                    const float top = fmaf(fx, b - a, a), bot = fmaf(fx, d - cc, cc);
                    const float v = fmaf(fy, bot - top, top);
                    o[channels * x + c] = (uint8_t)(int)(v + 0.5f);
                }
            } else {
                const float xr = floorf(xs + 0.5f), yr = floorf(ys + 0.5f);
                if (!(xr >= 0.0f && yr >= 0.0f && xr <= (float)(sw - 1) && yr <= (float)(sh - 1))) continue;
                const uint8_t *r = src + (size_t)(uint32_t)yr * src_pitch;
                for (uint32_t c = 0; c < channels; ++c) o[channels * x + c] = r[channels * (uint32_t)xr + c];
            }
        }
    }
}
