/*
 * ws_oracle.c -- CPU oracle for the WindowSearch hot path (see ws_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED by the reference's own tests
 * (there are none for this path) -- see the header for what pins it instead.
 *
 * The loops below keep the reference's iteration order, validity rules,
 * strict-< running minimum and double arithmetic:
 *   exact integer window sum -> sqrt (double) -> [/ area] -> [* smooth] -> '<'
 * so that tie-breaks and the smoothFactor raster dependency come out as the
 * reference's would.  Each function names the reference lines it follows.
 */
#include "ws_oracle.h"

#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

/* Row-parallel evaluation is only legal for smooth == 1.0 (no raster dependency). */
static int g_threads = 1;

void wso_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

static int wso_threads_for(double smooth) { return smooth == 1.0 ? g_threads : 1; }

static int image_ok(const wso_image *im)
{
    return im && im->data && im->width > 0 && im->height > 0 &&
           im->stride >= 3 * im->width;
}

static inline const uint8_t *px(const wso_image *im, int y, int x)
{
    return im->data + (size_t)y * (size_t)im->stride + (size_t)x * 3u;
}

static inline int is_black(const wso_image *im, int y, int x)
{
    const uint8_t *p = px(im, y, x);
    return p[0] == 0 && p[1] == 0 && p[2] == 0;
}

/*
 * cv::absdiff(A_roi, B_roi, diff) followed by cv::norm(diff, NORM_L2 | NORM_L1)
 * before the square root: the exact integer sum over ww x wh pixels x 3
 * channels (BlockSearch.cpp:64-66, :156-158).
 */
static uint64_t window_sum(const wso_image *A, int ax, int ay,
                           const wso_image *B, int bx, int by,
                           int ww, int wh, int cost)
{
    uint64_t acc = 0;
    for (int r = 0; r < wh; ++r) {
        const uint8_t *a = px(A, ay + r, ax);
        const uint8_t *b = px(B, by + r, bx);
        uint32_t row = 0;
        for (int i = 0; i < 3 * ww; ++i) {
            int d = (int)a[i] - (int)b[i];
            if (d < 0) d = -d;
            row += (cost == WSO_COST_SAD) ? (uint32_t)d : (uint32_t)(d * d);
        }
        acc += row;
    }
    return acc;
}

static inline double norm_of(uint64_t sum, int cost)
{
    return (cost == WSO_COST_SAD) ? (double)sum : sqrt((double)sum);
}

/* Build extension: parabola through the aggregated integer cost at d-1,d,d+1. */
static double parabola_offset(uint64_t cm, uint64_t c0, uint64_t cp)
{
    double num = (double)cm - (double)cp;
    double den = (double)cm - 2.0 * (double)c0 + (double)cp;
    if (!(den > 0.0)) return 0.0;
    return num / (2.0 * den);
}

int wso_block_left(const wso_image *L, const wso_image *R, int block_size,
                   int min_disparity, int max_disparity, double smooth,
                   int cost, int subpixel, int y0, int y1,
                   double *out, int out_stride)
{
    (void)min_disparity; /* never read by the reference's left view (BlockSearch.cpp:53) */
    if (!image_ok(L) || !image_ok(R) || !out || block_size < 1 ||
        out_stride < L->width || (cost != WSO_COST_SSD && cost != WSO_COST_SAD))
        return WSO_ERR_ARG;
    const int h1 = L->height, w1 = L->width, h2 = R->height, w2 = R->width;
    const int height = h1 < h2 ? h1 : h2;            /* :30 */
    const int half = (block_size - 1) / 2;           /* :31 */
    if (y0 < 0 || y1 > h1 || y0 > y1) return WSO_ERR_ARG;
    if (smooth != 1.0 && (y0 != 0 || subpixel)) return WSO_ERR_RANGE;

    for (int y = 0; y < h1; ++y)                     /* cv::Mat::zeros, :33 */
        for (int x = 0; x < w1; ++x) out[(size_t)y * out_stride + x] = 0.0;

    /* Rect(x-half, y-half, bs, bs) overruns the image for even bs (:46-49). */
    if ((block_size & 1) == 0 && height - 2 * half > 0 && w1 - 2 * half > 0)
        return WSO_ERR_GEOMETRY;

    int ya = half > y0 ? half : y0;
    int yb = height - half < y1 ? height - half : y1;
    const int nthreads = wso_threads_for(smooth);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1)
    for (int y = ya; y < yb; ++y) {                  /* :36 */
        for (int x = half; x < w1 - half; ++x) {     /* :38 */
            if (is_black(L, y, x)) continue;         /* :41-44 */
            int best_cx = 0;                         /* :50 */
            double best = DBL_MAX;                   /* :51 */
            int found = 0;
            for (int cx = x - max_disparity; cx < x; ++cx) {       /* :53 */
                if (cx < half || cx >= w2 - half) continue;        /* :55-57 */
                uint64_t s = window_sum(L, x - half, y - half, R, cx - half,
                                        y - half, block_size, block_size, cost);
                double dist = norm_of(s, cost);                    /* :64-66 */
                double dcand = (double)(x - cx);
                if (y >= 1 && out[(size_t)(y - 1) * out_stride + x] == dcand)
                    dist *= smooth;                                /* :68-70 */
                if (x >= 1 && out[(size_t)y * out_stride + x - 1] == dcand)
                    dist *= smooth;                                /* :71-73 */
                if (dist < best) {                                 /* :76-79 */
                    best_cx = cx;
                    best = dist;
                    found = 1;
                }
            }
            double value = (double)(x - best_cx);                  /* :82 */
            if (subpixel && found) {
                int cm = best_cx + 1, cp = best_cx - 1; /* d-1 and d+1 */
                int ok_m = cm < x && cm >= half && cm < w2 - half;
                int ok_p = cp >= x - max_disparity && cp >= half && cp < w2 - half;
                if (ok_m && ok_p) {
                    uint64_t c0 = window_sum(L, x - half, y - half, R, best_cx - half, y - half, block_size, block_size, cost);
                    uint64_t c_m = window_sum(L, x - half, y - half, R, cm - half, y - half, block_size, block_size, cost);
                    uint64_t c_p = window_sum(L, x - half, y - half, R, cp - half, y - half, block_size, block_size, cost);
                    value += parabola_offset(c_m, c0, c_p);
                }
            }
            out[(size_t)y * out_stride + x] = value;
        }
    }
    return WSO_OK;
}

/*
 * cv::mean + cv::subtract(window, mean, s) + cv::norm(s, NORM_L2) (BlockSearch.cpp:125-129).
 * OpenCV is un-vendored; restated from its 4.x behaviour: the mean is a per-channel double (exact
 * integer sum / count); a CV_8U Mat minus a non-integer Scalar is evaluated in FLOAT32 (arithm_op
 * narrows the Scalar to float and widens the pixel to float: depth2 = CV_32F, wtype = CV_32F) and
 * converted back with saturate_cast<uchar>(cvRound(float)), round half to even; an all-integer
 * mean takes the u8 saturating path, which gives the same bytes.  Float32 matters: with a mean
 * whose fraction lies within half a float ulp of .5 -- only possible for windows of more than
 * 32768 pixels, i.e. a varBlock window grown past ~182 x 182 on a near-flat region -- the float
 * mean IS k + .5 and the tie rounds to even where a double subtraction would not tie at all
 * (tests/test_oracle_construction.py shows such a window).
 */
double wso_centred_norm(const wso_image *im, int x0, int y0, int ww, int wh)
{
    if (ww <= 0 || wh <= 0) return 0.0;
    double mean[3] = {0, 0, 0};
    for (int r = 0; r < wh; ++r) {
        const uint8_t *p = px(im, y0 + r, x0);
        for (int i = 0; i < ww; ++i)
            for (int c = 0; c < 3; ++c) mean[c] += p[3 * i + c];
    }
    for (int c = 0; c < 3; ++c) mean[c] /= (double)ww * (double)wh;
    uint64_t acc = 0;
    for (int r = 0; r < wh; ++r) {
        const uint8_t *p = px(im, y0 + r, x0);
        for (int i = 0; i < ww; ++i)
            for (int c = 0; c < 3; ++c) {
                /* saturate_cast<uchar>(cvRound((float)p - (float)mean)): round half to even, clamp */
                volatile float fm = (float)mean[c];       /* (volatile: no double-precision shortcut) */
                volatile float fd = (float)p[3 * i + c] - fm;
                long v = lrintf(fd);
                if (v < 0) v = 0;
                if (v > 255) v = 255;
                acc += (uint64_t)(v * v);
            }
    }
    return sqrt((double)acc);
}

int wso_block_right(const wso_image *L, const wso_image *R, int block_size,
                    int min_disparity, int max_disparity, double smooth,
                    int var_block, double thres, int cost, int subpixel,
                    int y0, int y1, double *out, int out_stride,
                    int *max_block_out)
{
    if (!image_ok(L) || !image_ok(R) || !out || block_size < 1 ||
        out_stride < R->width || (cost != WSO_COST_SSD && cost != WSO_COST_SAD))
        return WSO_ERR_ARG;
    const int h1 = L->height, w1 = L->width, h2 = R->height, w2 = R->width;
    const int height = h1 < h2 ? h1 : h2;            /* :94 */
    if (y0 < 0 || y1 > h2 || y0 > y1) return WSO_ERR_ARG;
    if (smooth != 1.0 && (y0 != 0 || subpixel)) return WSO_ERR_RANGE;

    for (int y = 0; y < h2; ++y)                     /* :97 */
        for (int x = 0; x < w2; ++x) out[(size_t)y * out_stride + x] = 0.0;

    int max_block = block_size;                      /* :98 */
    int geometry_error = 0;
    int yb = height < y1 ? height : y1;
    const int nthreads = wso_threads_for(smooth);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1) reduction(max : max_block) reduction(| : geometry_error)
    for (int y = y0; y < yb; ++y) {                  /* :100 */
        for (int x = 0; x < w2 && !geometry_error; ++x) { /* :102 */
            if (is_black(R, y, x)) continue;         /* :105-108 */
            int best_cx = 0;
            double best = DBL_MAX;
            int found = 0;
            int bs = block_size;
            int hb = (bs - 1) / 2;
            int left = x < hb ? x : hb;                              /* :116 */
            int right = (w2 - x - 1) < hb ? (w2 - x - 1) : hb;       /* :117 */
            int up = y < hb ? y : hb;                                /* :118 */
            int down = (h2 - y - 1) < hb ? (h2 - y - 1) : hb;        /* :119 */
            if (var_block) {                                         /* :129-142 */
                while (wso_centred_norm(R, x - left, y - up, left + right, up + down) < thres) {
                    int pl = left, pr = right, pu = up, pd = down;
                    bs += 4;
                    hb = (bs - 1) / 2;
                    left = x < hb ? x : hb;
                    right = (w2 - x - 1) < hb ? (w2 - x - 1) : hb;
                    up = y < hb ? y : hb;
                    down = (h2 - y - 1) < hb ? (h2 - y - 1) : hb;
                    if (pl == left && pr == right && pu == up && pd == down)
                        break; /* cannot grow any more: the reference loops forever here */
                }
            }
            if (bs > max_block) max_block = bs;                      /* :144-145 */
            const int ww = left + right, wh = up + down;
            for (int cx = x + min_disparity; cx < x + max_disparity; ++cx) { /* :147 */
                if (cx + right >= w1) break;                         /* :148-149 */
                if (cx - left < 0 || y + down > h1) {
                    geometry_error = 1;            /* leftImage_(Rect) would throw, :151-154 */
                    break;
                }
                uint64_t s = window_sum(L, cx - left, y - up, R, x - left, y - up, ww, wh, cost);
                double dist = norm_of(s, cost) / (double)(ww * wh);  /* :158 (0/0 -> NaN) */
                double dcand = (double)(x - cx);
                if (y >= 1 && out[(size_t)(y - 1) * out_stride + x] == dcand)
                    dist *= smooth;                                  /* :160-162 */
                if (x >= 1 && out[(size_t)y * out_stride + x - 1] == dcand)
                    dist *= smooth;                                  /* :163-165 */
                if (dist < best) {                                   /* :168-171 */
                    best_cx = cx;
                    best = dist;
                    found = 1;
                }
            }
            double value = (double)(best_cx - x);                    /* :174 */
            if (subpixel && found) {
                int cm = best_cx - 1, cp = best_cx + 1; /* d-1 and d+1 */
                int ok_m = cm >= x + min_disparity && cm - left >= 0;
                int ok_p = cp < x + max_disparity && cp + right < w1;
                if (ok_m && ok_p) {
                    uint64_t c0 = window_sum(L, best_cx - left, y - up, R, x - left, y - up, ww, wh, cost);
                    uint64_t c_m = window_sum(L, cm - left, y - up, R, x - left, y - up, ww, wh, cost);
                    uint64_t c_p = window_sum(L, cp - left, y - up, R, x - left, y - up, ww, wh, cost);
                    value += parabola_offset(c_m, c0, c_p);
                }
            }
            out[(size_t)y * out_stride + x] = value;
        }
    }
    if (geometry_error) return WSO_ERR_GEOMETRY;
    if (max_block_out) *max_block_out = max_block;                   /* :177 */
    return WSO_OK;
}

int wso_linear(const wso_image *L, const wso_image *R, int range, double smooth,
               int y0, int y1, double *out, int out_stride)
{
    if (!image_ok(L) || !image_ok(R) || !out || range < 1 || out_stride < R->width)
        return WSO_ERR_ARG;
    const int h1 = L->height, w1 = L->width, h2 = R->height, w2 = R->width;
    if (y0 < 0 || y1 > h2 || y0 > y1) return WSO_ERR_ARG;
    if (smooth != 1.0 && y0 != 0) return WSO_ERR_RANGE;

    for (int i = 0; i < h2; ++i)                                 /* :19 */
        for (int j = 0; j < w2; ++j) out[(size_t)i * out_stride + j] = 0.0;

    int ib = h1 < y1 ? h1 : y1; /* rows i >= h1 cannot be read: defined as 0 */
    const int nthreads = wso_threads_for(smooth);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1)
    for (int i = y0; i < ib; ++i) {                              /* :21 */
        for (int j = 0; j < w2; ++j) {                           /* :22 */
            if (j < w1 && is_black(L, i, j)) continue;           /* :24-27 */
            int col = 0;                                         /* :29 */
            const uint8_t *pr = px(R, i, j);                     /* :30 */
            double best = DBL_MAX;
            for (int k = j; k < j + range; ++k) {                /* :32 */
                if (k < 0) continue;                             /* :33-34 */
                if (k >= w1) break;            /* defined: the reference reads out of bounds */
                const uint8_t *pl = px(L, i, k);                 /* :35 */
                int d0 = (int)pr[0] - (int)pl[0];
                int d1 = (int)pr[1] - (int)pl[1];
                int d2 = (int)pr[2] - (int)pl[2];
                double dist = sqrt((double)(d0 * d0 + d1 * d1 + d2 * d2)); /* :36-37 */
                double dcand = (double)(j - k);
                if (i >= 1 && out[(size_t)(i - 1) * out_stride + j] == dcand)
                    dist *= smooth;                              /* :39-41 */
                if (j >= 1 && out[(size_t)i * out_stride + j - 1] == dcand)
                    dist *= smooth;                              /* :42-44 */
                if (dist < best) {                               /* :46-49 */
                    col = k;
                    best = dist;
                }
            }
            out[(size_t)i * out_stride + j] = (double)(col - j); /* :53 */
        }
    }
    return WSO_OK;
}

int wso_evaldisp(const float *disp, const float *gt, const uint8_t *mask,
                 int width, int height, float badthresh, float maxdisp,
                 int rounddisp, double res[6])
{
    if (!disp || !gt || !mask || !res || width <= 0 || height <= 0) return WSO_ERR_ARG;
    int n = 0, bad = 0, invalid = 0;                             /* utils.cpp:131-133 */
    float serr = 0;
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            size_t o = (size_t)y * width + x;
            float g = gt[o];
            if (g == INFINITY) continue;                         /* :137 */
            float d = disp[o];
            int valid = (d != 0);                                /* :140 */
            if (valid) d = fmaxf(0.0f, fminf(maxdisp, d));       /* :142 */
            if (valid && rounddisp) d = roundf(d);               /* :144 */
            float err = fabsf(d - g);
            if (mask[o] != 255) continue;                        /* :146 */
            n++;
            if (valid) {
                serr += err;
                if (err > badthresh) bad++;
            } else {
                invalid++;
            }
        }
    }
    res[0] = n;
    res[1] = (float)(100.0 * bad / n);                           /* :158-161 */
    res[2] = (float)(100.0 * invalid / n);
    res[3] = (float)(100.0 * (bad + invalid) / n);
    res[4] = serr / (float)(n - invalid);
    res[5] = 100.0 * n / ((double)width * height);
    return WSO_OK;
}
