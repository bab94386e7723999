// ws_kernels.h -- launch interface between the C-ABI host code (ws_capi.cpp) and the
// gfx950 kernels (ws_march / ws_prepass / ws_border / ws_smooth / ws_consumers .hip).  Internal; the public boundary is include/ws_stereo.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wsamd {

// Poison added to a key for an invalid candidate; valid keys stay inside (-2^28, 2^28).
constexpr int32_t kPoison = 1 << 29;
constexpr int32_t kValidKeyBound = 1 << 28;

// The "canonical" search every view is reduced to:
//   outputs live on plane A, candidates d in [d_lo, d_hi] look at plane B column x - d + boff.
// LEFT  (BlockSearch.cpp:24-86):  A = left,  B = right, window bs x bs centred.
// RIGHT (BlockSearch.cpp:88-179): A = right, B = left, both mirrored in x so that the
//   reference's cx = x + d becomes x' - d + (w1 - w2); window (bs-1) x (bs-1).
struct Canon {
    int wa, ha, wb, hb;     // plane sizes (pixels)
    int ww, wh;             // window width / height
    int wx0, wy0;           // window origin relative to the output pixel
    int boff;               // B column = x - d + boff
    int d_lo, d_hi;         // inclusive candidate range (clamped to what complete windows allow)
    int d_hi_clipped;       // ... and what the clipped windows of the right view's border ring allow
    int b_lo, b_hi;         // inclusive range of valid B centre columns
    int ox0, ox1, oy0, oy1; // outputs computed by the marching kernel: [ox0,ox1) x [oy0,oy1)
    int prefer_large;       // ties: 1 -> largest d wins (left), 0 -> smallest d (right)
    int mirror;             // canonical x = wa - 1 - original x
    int fallback_neg;       // no valid candidate: store +x (left) or -x (right), original x
    int ssd;                // 1 = SSD (NORM_L2), 0 = SAD (NORM_L1)
};

// Packed planes: one uint32 per pixel (B | G<<8 | R<<16), zero outside the image.
struct Plane {
    uint32_t *data;
    int pitch; // dwords per row
    int pad;   // plane column = image column + pad
};

struct MarchLaunch {
    int x_per_thread, nd_per_thread; // X, ND template choice
    int nxr, nch;                    // x-runs per tile, d-chunks per tile (and pass)
    int passes;                      // d-group passes (1 unless the disparity range is very wide)
    int threads;                     // workgroup size
    int tiles, strips, strip_rows;
    int halo;                        // packed SAD with the halo exchange (march_pk_halo): tiles advance by (nxr - 1) * X columns
    int tile_cols;                   // output columns per tile
    size_t lds_bytes;
    int max_threads;                 // launch-bounds variant (1024 or 768)
};

// Which (window, X, ND) instantiations exist.  Returns false if none fits.
bool march_supported(const Canon &c);
// 1 if this SSD window needs centred (byte - 128) planes to keep its sums in 32 bits
int march_centred(const Canon &c);
// The CU count the thread-shape rule (march_shape) plans for; ws_create passes its device's.
void march_set_num_cus(int n);
// Fill the tiling for this problem (tuning values of 0 = automatic).
bool march_plan(const Canon &c, int num_cus, int tune_nxr, int tune_strip_rows, int tune_threads,
                MarchLaunch *out);
// Plane geometry (pad / pitch) the plan needs.
void march_plane_geometry(const Canon &c, const MarchLaunch &m, Plane *a, Plane *b);

struct GenericArgs;
// pack both planes (for the kernels beside the marching kernel that read planes: border ring, refine, smoothFactor passes)
hipError_t launch_pack(const Canon &c, const uint8_t *src_a, int stride_a, Plane dst_a, const uint8_t *src_b, int stride_b,
                       Plane dst_b, hipStream_t s);
// The hot kernel, on the caller's CV_8UC3 rows themselves (no planes, no pre-pass): img_a carries the outputs (left view:
// the left image), img_b the candidates.  border: also write the zeros of the out_w x out_h map outside the marching
// interior (left view).  keys: plane of 8-byte keys (wa x ha, pitch in elements), only touched when m.passes > 1
hipError_t launch_march(const Canon &c, const MarchLaunch &m, const uint8_t *img_a, int stride_a, const uint8_t *img_b, int stride_b,
                        float *out, int16_t *out16, int out_pitch, int border, int out_w, int out_h, void *keys, int keys_pitch,
                        int32_t *cost_out, int cost_pitch,
                        hipStream_t s); // cost_out: optional plane of the winners' costs (SSD: without sum a^2); out16: see GenericArgs
const char *march_kernel_name(const Canon &c, const MarchLaunch &m);
bool march_has_cost(const Canon &c); // is there an instantiation that also writes cost_out?

// Brute-force kernels on the original 8-bit images (original coordinates, literal rules).
struct GenericArgs {
    const uint8_t *L;
    const uint8_t *R;
    int w1, h1, s1, w2, h2, s2;
    int view, ssd, block_size, min_d, max_d, linear_range;
    // pixels inside [skip_x0,skip_x1) x [skip_y0,skip_y1) are left to the marching kernel
    int skip_x0, skip_x1, skip_y0, skip_y1;
    float *out;
    int out_pitch;
    int16_t *out16; // if set, the search kernels store the map as 16-bit integers here (same pitch, in elements) instead of floats to `out`:
                    // the caller (ws_capi.cpp) knows every value is an integer in [-32767, 32767]
    // varBlock (right view): per-pixel block size chosen by ws_varblock_kernel, or null
    const int16_t *bs_plane;
    int bs_pitch;
};
hipError_t launch_generic(const GenericArgs &g, hipStream_t s);
// LinearSearch through LDS (falls back to launch_generic for ranges beyond 4096 candidates)
hipError_t launch_linear(const GenericArgs &g, hipStream_t s);
// Right-view border ring on the packed (mirrored) planes: sliding sums along runs, lanes over d.
hipError_t launch_ring(const Canon &c, Plane a, Plane b, const GenericArgs &skip, float *out, int out_pitch,
                       int32_t *cost_out, int cost_pitch, hipStream_t s);
// Sub-pixel refinement (extension): parabola through the integer cost at d-1, d, d+1.
hipError_t launch_refine(const GenericArgs &g, hipStream_t s);
// the same for the marching interior, on the packed planes (launch_refine then skips g's skip rectangle)
hipError_t launch_refine_planes(const Canon &c, const MarchLaunch &m, Plane a, Plane b, float *out, int out_pitch, hipStream_t s);
// smoothFactor != 1, right view / LinearSearch: g.out must hold the d >= 1 search result
// rows the sel plane must be allocated with (whole LDS chunks are copied)
int smooth_sel_rows(int rows);
// smoothFactor in [0,1], left view: g.out holds the smoothFactor-1 result on entry
// top3: smooth_left_top_bytes(w1, h1, s) of scratch (the row-sum volume at its end only for s outside [0,1])
// gave_up: host-visible word (device pointer) a band of the raster pass sets when it gave up waiting (the map is then invalid)
hipError_t launch_smooth_left(const GenericArgs &g, double s, uint32_t *top3, const Canon *canon, Plane pa, Plane pb,
                              unsigned int *gave_up, hipStream_t st);
size_t smooth_left_top_bytes(int w, int h, double s);
// bytes of the bit-plane scratch launch_smooth wants for a w x h map
size_t smooth_planes_bytes(int w, int h);
// canon / pa / pb: the right view's canonical search and packed planes when the marching kernel ran
// (g's skip rectangle = its interior) together with the cost plane it and the ring kernel wrote,
// else canon == nullptr
hipError_t launch_smooth(const GenericArgs &g, double s, uint8_t *sel, int sel_pitch, unsigned long long *planes,
                         const Canon *canon, Plane pa, Plane pb, const int32_t *cost, int cost_pitch, hipStream_t st);
// nearest-neighbour perspective warp of a float map; minv maps destination -> source pixels
hipError_t launch_warp(const float *src, int sw, int sh, int sp, float *dst, int dw, int dh, int dp,
                       const double minv[9], hipStream_t s);
// removeDisparityOutliers (reconstruction.cpp:5-18); scratch = w*h doubles
hipError_t launch_outliers(float *map, int mp, int w, int h, int k, float thr_front, float thr_back, double *scratch,
                           hipStream_t s);
// the same for 8-bit maps (what the pipeline feeds it: an 8-bit PNG), all sums in 32-bit integers.  A map with any
// value that is not an integer in [0, 255] sets *flag (device word, zero on entry) and *host_word (mapped host word)
// and is left untouched: the caller then runs launch_outliers.  scratch = outliers_u32_scratch_bytes(w, h) bytes.
bool outliers_u32_applies(int w, int h, int k, int num_cus);
size_t outliers_u32_scratch_bytes(int w, int h, int num_cus);
hipError_t launch_outliers_u32(float *map, int mp, int w, int h, int k, float thr_front, float thr_back, uint32_t *scratch,
                               uint32_t *flag, unsigned int *host_word, int num_cus, hipStream_t s);
// convertDisparityToDepth + back-projection (reconstruction.cpp:30-43, :152-196); depth / pos+col may be null
hipError_t launch_depth_vertices(const float *disp, int dp, int w, int h, float focal, float baseline, const float k[9],
                                 const uint8_t *bgr, int bstride, float *depth, int zp, float *pos, uint8_t *col,
                                 int input_is_depth, hipStream_t s);
// varBlock (BlockSearch.cpp:125-145, right view): per-pixel window growth + search, one wave per pixel.
// bs_plane: w2 x h2 int16 (pitch bs_pitch), max_block: one device int (max grown block size)
hipError_t launch_varblock(const GenericArgs &g, double thres, int16_t *bs_plane, int bs_pitch, int *max_block,
                           hipStream_t s);
} // namespace wsamd
