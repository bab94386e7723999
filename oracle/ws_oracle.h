/*
 * ws_oracle.h -- CPU oracle for the WindowSearch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under stereo_reconstruction_amd/ (the
 * product) may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed
 * CPU baseline.
 *
 * It restates, in plain C with no third-party dependency, the algorithm of
 *   /root/reference/src/WindowSearch/BlockSearch.cpp:24-86   (left view)
 *   /root/reference/src/WindowSearch/BlockSearch.cpp:88-179  (right view)
 *   /root/reference/src/WindowSearch/LinearSearch.cpp:10-59  (1x1 search)
 * including the OpenCV semantics those lines rely on (cv::absdiff on u8,
 * cv::norm NORM_L2 = sqrt of the exact integer sum of squares, cv::mean,
 * cv::subtract with a Scalar).  OpenCV itself is un-vendored and unpinned in
 * the reference (find_package(OpenCV) without a version, src/CMakeLists.txt:10).
 *
 * PARITY UNPINNED: the reference holds no test, golden vector or stored output
 * for WindowSearch and cannot be built in this image
 * (OpenCV / Eigen absent).  The restatement is pinned instead by construction
 * cases and by an independently written NumPy brute force (oracle/brute.py),
 * see tests/test_oracle_*.py.
 *
 * Images are 8-bit, 3 channels interleaved (BGR as cv::imread gives them),
 * row-major with a byte stride.  Outputs are row-major double maps with the
 * reference's CV_64F meaning (integer-valued unless sub-pixel is requested).
 */
#ifndef WS_ORACLE_H
#define WS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { WSO_COST_SSD = 0, WSO_COST_SAD = 1 };

enum {
    WSO_OK = 0,
    WSO_ERR_ARG = -1,      /* null pointer, non-positive size, stride too small */
    WSO_ERR_GEOMETRY = -2, /* the reference would hit a cv::Exception (ROI outside the image) */
    WSO_ERR_RANGE = -3     /* row band requested together with smoothFactor != 1 */
};

typedef struct {
    const uint8_t *data;
    int width;
    int height;
    int stride; /* bytes per row, >= 3*width */
} wso_image;

/*
 * Number of host threads the three searches may use (OpenMP over rows).  The
 * reference is single-threaded; rows are only independent for smooth == 1.0, so
 * any other smooth value runs on one thread whatever is set here.  Default 1.
 */
void wso_set_threads(int n);

/*
 * BlockSearch::computeDisparityMapLeft (BlockSearch.cpp:24-86).
 * out: h1 x w1 doubles (out_stride in elements).  Rows outside [y0,y1) are
 * left zero; pass y0=0,y1=h1 for the whole map.  A proper sub-band is only
 * legal for smooth == 1.0 (raster dependency otherwise, BlockSearch.cpp:68-73).
 * cost = WSO_COST_SAD is the build's NORM_L1 extension.
 * subpixel != 0 adds the build's parabolic refinement on the aggregated
 * integer cost (extension; smooth must be 1.0).
 */
int wso_block_left(const wso_image *L, const wso_image *R, int block_size,
                   int min_disparity, int max_disparity, double smooth,
                   int cost, int subpixel, int y0, int y1,
                   double *out, int out_stride);

/*
 * BlockSearch::computeDisparityMapRight (BlockSearch.cpp:88-179).
 * out: h2 x w2 doubles.  max_block_out (may be NULL) receives the value the
 * reference prints as "max block size" (BlockSearch.cpp:177).
 * var_block growth is capped when the window can no longer grow (the
 * reference would spin forever there, BlockSearch.cpp:129-142).
 */
int wso_block_right(const wso_image *L, const wso_image *R, int block_size,
                    int min_disparity, int max_disparity, double smooth,
                    int var_block, double thres, int cost, int subpixel,
                    int y0, int y1, double *out, int out_stride,
                    int *max_block_out);

/*
 * LinearSearch::computeDisparityMap (LinearSearch.cpp:10-59).
 * out: h2 x w2 doubles.  range is the reference's hard-coded 200
 * (LinearSearch.cpp:32).  Defined behaviour for the reference's out-of-bounds
 * reads: candidates k >= w1 are skipped; rows i >= h1 stay 0; for j >= w1 the
 * black-pixel test is taken as false.
 */
int wso_linear(const wso_image *L, const wso_image *R, int range, double smooth,
               int y0, int y1, double *out, int out_stride);

/*
 * The varBlock texture test alone: cv::norm(window - cv::mean(window), NORM_L2) of the ww x wh window at
 * (x0, y0) (BlockSearch.cpp:125-129), with OpenCV's float32 subtraction (see ws_oracle.c).
 */
double wso_centred_norm(const wso_image *im, int x0, int y0, int ww, int wh);

/*
 * evaldisp (utils.cpp:123-168): Middlebury bad-pixel statistics.
 * res[0]=n evaluated, res[1]=bad%, res[2]=invalid%, res[3]=total bad%,
 * res[4]=avgErr, res[5]=valid% of all pixels.
 */
int wso_evaldisp(const float *disp, const float *gt, const uint8_t *mask,
                 int width, int height, float badthresh, float maxdisp,
                 int rounddisp, double res[6]);

#ifdef __cplusplus
}
#endif
#endif
