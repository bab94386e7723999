/*
 * ws_stereo.h -- C-ABI of the MI355X-native WindowSearch (dense block matching).
 *
 * This is the drop-in boundary for the reference's src/WindowSearch stage: plain
 * pointers and sizes, no OpenCV / torch / C++ types.  Every entry point names
 * the reference interface it replaces (paths relative to the reference root).
 * The C++ facade with the reference's class names lives in
 * stereo_reconstruction_amd/host/window_search.hpp; the binding a maintainer of
 * the reference would add is shown in INTEGRATION.md.
 *
 * All compute runs in hand-written HIP kernels for gfx950.  There is no CPU
 * fallback: without a usable HIP device ws_create() fails with WS_ERR_HIP.
 *
 * Images: 8-bit, 3 channels interleaved in cv::imread order (BGR), row-major,
 * `stride` bytes per row (CV_8UC3, BlockSearch.cpp:41, :105).  Left and right
 * may differ in size (BlockSearch.cpp:25-30).  Disparity maps: row-major,
 * integer-valued unless sub-pixel refinement is on.
 */
#ifndef WS_STEREO_H
#define WS_STEREO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WS_VERSION 100 /* 0.1.0 */

/* status codes (the reference has none: it throws cv::Exception / loops forever) */
enum {
    WS_OK = 0,
    WS_ERR_ARG = -1,         /* null pointer, bad size / stride / enum */
    WS_ERR_GEOMETRY = -2,    /* the reference would throw cv::Exception: even blockSize in the
                                left view (BlockSearch.cpp:46-49), left ROI outside the image in
                                the right view (BlockSearch.cpp:151-154) */
    WS_ERR_UNSUPPORTED = -3, /* legal for the reference, not implemented on the device yet */
    WS_ERR_HIP = -4,         /* HIP runtime error (see ws_last_error) */
    WS_ERR_IO = -5,          /* file could not be read / written / parsed */
    WS_ERR_NOMEM = -6
};

/* which reference method the call stands for */
enum {
    WS_VIEW_LEFT = 0,   /* BlockSearch::computeDisparityMapLeft   (BlockSearch.cpp:24-86)  */
    WS_VIEW_RIGHT = 1,  /* BlockSearch::computeDisparityMapRight  (BlockSearch.cpp:88-179) */
    WS_VIEW_LINEAR = 2  /* LinearSearch::computeDisparityMap      (LinearSearch.cpp:10-59) */
};

enum {
    WS_COST_SSD = 0, /* the reference's cost: cv::norm(absdiff, NORM_L2) (BlockSearch.cpp:64-66) */
    WS_COST_SAD = 1  /* extension: NORM_L1 in the same loops (BASELINE.json configs 1 and 3) */
};

enum { WS_OUT_F32 = 0, WS_OUT_F64 = 1 }; /* F64 = the reference's CV_64F maps (BlockSearch.cpp:33) */

typedef struct {
    const uint8_t *data; /* host pointer for *_host calls, device pointer for *_device calls */
    int width;
    int height;
    int stride;          /* bytes per row, >= 3 * width */
} ws_image;

/*
 * The constructor + call arguments of the reference, in one struct.
 *   BlockSearch(L, R, blockSize, minDisparity, maxDisparity)        BlockSearch.h:11-15
 *   computeDisparityMapLeft(smoothFactor)                            BlockSearch.h:28
 *   computeDisparityMapRight(smoothFactor, varBlock, thres)          BlockSearch.h:37
 *   LinearSearch(L, R).computeDisparityMap(smoothFactor)             LinearSearch.h:13-19
 * Use ws_params_default() and then set what differs.
 */
typedef struct {
    int view;             /* WS_VIEW_* */
    int cost;             /* WS_COST_* (LINEAR ignores it: always the Euclidean pixel distance) */
    int block_size;       /* blockSize (LINEAR ignores it) */
    int min_disparity;    /* minDisparity: read by the right view only (BlockSearch.cpp:147) */
    int max_disparity;    /* maxDisparity: left tries d = maxD..1, right d = minD..maxD-1 */
    double smooth_factor; /* smoothFactor, any value but NaN (!= 1 in the left view is a true raster-order
                             dependency, SURVEY.md 8f-1: a serial pass of a few ms) */
    int var_block;        /* varBlock (right view): grow the window while its centred norm < thres */
    double thres;         /* thres for varBlock, default 19.0 (BlockSearch.h:37) */
    int subpixel;         /* extension: parabolic refinement on the aggregated integer cost */
    int linear_range;     /* LinearSearch's hard-coded 200 candidates (LinearSearch.cpp:32) */
} ws_params;

typedef struct ws_context ws_context; /* one per device; not to be shared between threads */

/* ---- life cycle ------------------------------------------------------------------- */
int ws_version(void);
void ws_params_default(ws_params *p);
/* Open HIP device `device`.  Owns a stream, scratch planes and staging buffers. */
int ws_create(int device, ws_context **out);
void ws_destroy(ws_context *ctx);
/* Text of the last error on this context (or of the last failed ws_create if ctx == NULL). */
const char *ws_last_error(const ws_context *ctx);
int ws_device_count(void);

/*
 * The argument checks of the search calls without a device: the status a search with these
 * arguments would return before launching anything (WS_OK, WS_ERR_ARG, WS_ERR_GEOMETRY for
 * what makes the reference throw, WS_ERR_UNSUPPORTED).  Message via ws_last_error(NULL).
 */
int ws_validate(const ws_params *p, const ws_image *left, const ws_image *right);

/* How a search would be tiled on a device with num_cus compute units (0 = 256); host logic only. */
typedef struct {
    int marching;                   /* 1: the marching kernel owns the interior; 0: brute force only */
    int x_per_thread, d_per_thread; /* columns x disparities whose window sums one thread keeps */
    int x_runs, d_chunks;           /* per workgroup: tile = x_runs*x_per_thread columns, all d */
    int threads, tiles, strips, strip_rows, lds_bytes;
    int interior_x0, interior_x1, interior_y0, interior_y1; /* outputs the marching kernel writes */
    int passes;                     /* d-group passes (disparity ranges wider than one tile holds) */
    int tile_cols;                  /* columns a tile hands out: x_runs*x_per_thread, one run less for the halo-exchange SAD kernels */
} ws_plan_info;
int ws_plan(const ws_params *p, const ws_image *left, const ws_image *right, int num_cus,
            ws_plan_info *out);

/* ---- the hot path ----------------------------------------------------------------- */
/*
 * Synchronous call on host buffers; replaces
 *   BlockSearch(L,R,bs,minD,maxD).computeDisparityMapLeft/Right(...)   BlockSearch.cpp:24-179
 *   LinearSearch(L,R).computeDisparityMap(s)                           LinearSearch.cpp:10-59
 * as called from ImageRectifier::computeDisparityMapLeft/Right (rectification.cpp:66-88)
 * and rectification_main.cpp:194-195.
 * out: h1 x w1 (LEFT) or h2 x w2 (RIGHT, LINEAR) elements of out_dtype, out_stride in
 * elements.  Copies in, runs, copies out, returns when the map is complete.  Pageable buffers cross
 * through pinned staging memory of the library's -- every entry point taking host buffers does that, and none
 * registers caller memory with the HIP runtime; a buffer the caller has pinned itself is used as it is
 * (INTEGRATION.md section 2).  An integer-valued map crosses PCIe as 16-bit integers and is widened to out_dtype
 * on the host (ws_last_wire_format).
 */
int ws_search_host(ws_context *ctx, const ws_params *p, const ws_image *left,
                   const ws_image *right, void *out, int out_stride, int out_dtype);

/*
 * The same on device-resident buffers (images already in HBM, float32 map written to HBM),
 * enqueued on `stream` (a hipStream_t; NULL = the context's own stream).  Returns after
 * enqueueing; order / wait on the stream as usual.  This is what bench.py times.
 */
int ws_search_device(ws_context *ctx, const ws_params *p, const ws_image *left_dev,
                     const ws_image *right_dev, float *out_dev, int out_stride, void *stream);

/*
 * Batched host form for many independent pairs (BASELINE.json config 4): two pairs are kept in
 * flight -- while one is searched, the next one's images go up and the previous one's map comes
 * down on a second stream.  The images and `out` must stay valid and untouched until ws_wait(),
 * which blocks until every enqueued pair's map has landed in its `out`.
 */
int ws_enqueue_host(ws_context *ctx, const ws_params *p, const ws_image *left,
                    const ws_image *right, void *out, int out_stride, int out_dtype);
int ws_wait(ws_context *ctx);

/*
 * The back-projection the caller of the hot path applies to its result:
 *   cv::warpPerspective(disparityMap_rect, disparityMapLeft, H_.inv(), size, INTER_NEAREST)
 *   (rectification.cpp:70-75, :82-87).  `m` is the 3x3 matrix handed to warpPerspective (row-major,
 *   i.e. H_.inv()); like OpenCV the call inverts it and gathers dst(x,y) = src(round(M^-1 (x,y,1))),
 *   0 outside.  For already rectified pairs m is the identity and this is a copy.
 */
int ws_warp_nearest_host(ws_context *ctx, const double *src, int src_w, int src_h, int src_stride,
                         const double m[9], double *dst, int dst_w, int dst_h, int dst_stride);
int ws_warp_nearest_device(ws_context *ctx, const float *src_dev, int src_w, int src_h, int src_stride,
                           const double m[9], float *dst_dev, int dst_w, int dst_h, int dst_stride,
                           void *stream);

/* ---- consumers of the map: the Reconstruction side of the call surface ------------------- */
/*
 * removeDisparityOutliers(disparityMap, kernelSize, thrFront, thrBack)  (reconstruction.cpp:5-18,
 * main.cpp:53): k x k cv::blur (normalised box, BORDER_REFLECT_101), then every value above
 * thrFront * blurred or below thrBack * blurred is replaced by the blurred one.  In place, float32.
 */
int ws_remove_disparity_outliers(ws_context *ctx, float *map, int width, int height, int stride,
                                 int kernel_size, float thr_front, float thr_back);
/* convertDisparityToDepth(dispImage, focalLength, baseline)  (reconstruction.cpp:30-43): f*b/d, 0 -> -inf */
int ws_convert_disparity_to_depth(ws_context *ctx, const float *disp, int width, int height, int stride,
                                  float focal_length, float baseline, float *depth, int depth_stride);
/*
 * The back-projection loop of reconstruction() (reconstruction.cpp:152-196): positions = w*h x 4
 * floats (x_cam, y_cam, depth, 1; all -inf where depth is -inf), colors = w*h x 4 bytes (R,G,B,255).
 * intrinsics = row-major 3x3 (fx, 0, cx, 0, fy, cy, ...).
 */
int ws_back_project(ws_context *ctx, const float *depth, int width, int height, int stride,
                    const float intrinsics[9], const ws_image *bgr, float *positions, uint8_t *colors);
/* WriteMesh (reconstruction.cpp:72-149): COFF text file; host only. */
int ws_write_mesh_off(const char *path, const float *positions, const uint8_t *colors, int width,
                      int height, float edge_threshold);

/* ---- measurement ------------------------------------------------------------------ */
/* hipEvent pair on `stream` (NULL = context stream): begin, enqueue work, end -> elapsed ms. */
int ws_timer_begin(ws_context *ctx, void *stream);
int ws_timer_end(ws_context *ctx, void *stream, float *elapsed_ms);
/*
 * With profiling on, every ws_search_* call brackets its dominant kernel (the marching
 * kernel) with a hipEvent pair on the launch stream; ws_last_kernel_ms waits for it and
 * returns that one launch's duration.  This is what bench.py's `roofline` is computed from.
 */
int ws_set_profiling(ws_context *ctx, int enable);
int ws_last_kernel_ms(ws_context *ctx, float *elapsed_ms);
/*
 * What the reference prints as "max block size" after computeDisparityMapRight
 * (BlockSearch.cpp:177): the largest block varBlock grew to in the last right-view call of this
 * context (block_size itself if nothing grew or varBlock was off).  Synchronises the device.
 */
int ws_last_max_block(ws_context *ctx, int block_size, int *max_block);
/*
 * After a ws_search_* call: the kernel that dominates it and how the path was tiled
 * (name as it appears in a rocprofv3 kernel trace, threads per workgroup, workgroups,
 * dynamic LDS bytes).  For reports; not part of the reference's surface.
 */
int ws_last_launch_info(const ws_context *ctx, char *kernel_name, int name_cap,
                        int *threads, int *workgroups, int *lds_bytes);
/* Tuning knob for the LDS tile sweep of BASELINE.json config 3: 0 = automatic. */
int ws_set_tuning(ws_context *ctx, int x_runs_per_tile, int strip_rows, int threads);
/*
 * ws_search_host cuts a big call into row bands whose host<->device copies overlap the searches of their
 * neighbours (smoothFactor 1, equal image heights; results are identical: each row only depends on the rows
 * under its window, BlockSearch.cpp:46-66).  bands: -1 = automatic (4 for maps of a megapixel or more),
 * 0 or 1 = never, 2..8 = that many.  For measurements; not part of the reference's surface.
 */
int ws_set_host_bands(ws_context *ctx, int bands);
/*
 * ws_search_device only enqueues.  This waits for `stream` (NULL: the context's) and returns what the kernels
 * flagged since the last check: WS_ERR_HIP if a band of the left view's smoothFactor raster pass
 * (BlockSearch.cpp:68-73 in raster order) gave up waiting for the band above it -- the map is then not valid --
 * else WS_OK.  The host entry points (ws_search_host, ws_wait) make the same check themselves.
 */
int ws_device_status(ws_context *ctx, void *stream);
/*
 * How the bytes of the last host call's three buffers (left, right, out; for a batch: of its last pair) crossed:
 * 0 = not a linear span (gathered rows), 2 = memory the caller (or a framework) had pinned already, used as it is,
 * 3 = through pinned staging memory of the library (pageable memory always does: this library registers no caller
 * memory).  (1 = registered by this library: rounds 2-3 only, never returned any more.)
 * Stands in for nothing in the reference (cv::Mat buffers are pageable and rectification.cpp:66-88 never leaves the
 * host); for tests and reports.
 */
int ws_last_host_paths(const ws_context *ctx, int how[3]);
/*
 * The format the last ws_search_host call's map crossed PCIe in: 1 = 16-bit integers (every value a search stores is
 * an integer in [-width, max(maxDisparity, width)], BlockSearch.cpp:33,82,174 -- taken whenever those bounds fit 16
 * bits and the search kernels write the map themselves: smoothFactor 1, no sub-pixel refine, no varBlock), widened to
 * the caller's CV_32F / CV_64F on the host inside the copy out of the staging memory; 2 = float32.  Doubles never
 * cross.  The map the caller gets is the same either way; for tests and reports.
 */
int ws_last_wire_format(const ws_context *ctx, int *wire);
/*
 * Which kernels the last ws_remove_disparity_outliers call ran: 0 = the double-precision box filter, 1 = the 32-bit
 * integer one (every value of the map an integer in [0, 255] and kernel_size <= 4000: what reconstruction.cpp:5-18 is
 * fed by main.cpp:47-53, an 8-bit PNG blurred over 500 x 500), 2 = the integer one met another value, left the map
 * alone, and the double one ran after it.  The results are identical; for tests and reports.
 */
int ws_last_outliers_path(const ws_context *ctx, int *path);

/* ---- Middlebury plumbing around the path ------------------------------------------ */
/*
 * PFM ("Pf", one channel): replaces the Middlebury SDK ReadImageVerb the reference uses for
 * disp0GT.pfm (data_loader.cpp:110-125).  Rows are returned top-to-bottom; unknown = +inf.
 * ws_pfm_read allocates *data with malloc (release with ws_free).
 */
int ws_pfm_read(const char *path, float **data, int *width, int *height);
int ws_pfm_write(const char *path, const float *data, int width, int height, int stride);
void ws_free(void *p);
/*
 * Binary PPM ("P6") <-> BGR rows: stands in for cv::imread(IMREAD_COLOR) / cv::imwrite of the
 * reference's PNG pairs (data_loader.cpp:71-72); no PNG codec is linked.  ws_ppm_read allocates
 * *bgr with malloc (release with ws_free).
 */
int ws_ppm_read(const char *path, uint8_t **bgr, int *width, int *height);
int ws_ppm_write(const char *path, const uint8_t *bgr, int width, int height, int stride);
/*
 * calib.txt: cam0 / cam1 as the reference parses them (data_loader.cpp:141-164), row-major
 * 3x3 each, plus the keys it leaves unread (ndisp, doffs, baseline, width, height; -1 if absent).
 */
typedef struct {
    float cam0[9];
    float cam1[9];
    float doffs, baseline;
    int width, height, ndisp;
} ws_calib;
int ws_calib_read(const char *path, ws_calib *out);
/*
 * evaldisp (utils.cpp:123-168): the bad-pixel metric ("bad-2.0" = badthresh 2.0).
 * res[0]=n, res[1]=bad %, res[2]=invalid %, res[3]=total bad %, res[4]=avgErr, res[5]=valid %.
 */
int ws_evaldisp(const float *disp, const float *gt, const uint8_t *mask, int width,
                int height, float badthresh, float maxdisp, int rounddisp, double res[6]);

#ifdef __cplusplus
}
#endif
#endif /* WS_STEREO_H */
