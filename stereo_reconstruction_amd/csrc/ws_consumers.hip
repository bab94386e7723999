// ws_consumers.hip -- warpPerspective back-projection, Reconstruction-side consumers, f32 -> f64
// Part of the gfx950 kernels of the WindowSearch hot path; overview in ws_march.hip.
#include "ws_device.h"

namespace wsamd {

// ------------------------------------------------------------------------------------------
// back-projection of a disparity map: cv::warpPerspective(map, dst, M, dst.size(), INTER_NEAREST)
// as ImageRectifier::computeDisparityMapLeft/Right call it with M = H_.inv()
// (rectification.cpp:70-75, :82-87).  OpenCV (un-vendored, 4.x semantics restated): M is inverted,
// destination pixel (x, y) reads source pixel (cvRound(X/W), cvRound(Y/W)) of (X,Y,W) = M^-1 (x,y,1),
// evaluated per 64-column block as (M0*xb + M1*y + M2 + M0*x1) * (1/W); outside -> 0.
// ------------------------------------------------------------------------------------------
struct WarpArgs {
    const float *src;
    int sw, sh, sp;
    float *dst;
    int dw, dh, dp;
    double m[9]; // already inverted: destination -> source
};

__global__ void __launch_bounds__(256) ws_warp_kernel(const WarpArgs g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= g.dw || y >= g.dh) return;
    const int xb = x & ~63, x1 = x & 63;
    const double X0 = g.m[0] * xb + g.m[1] * y + g.m[2];
    const double Y0 = g.m[3] * xb + g.m[4] * y + g.m[5];
    const double W0 = g.m[6] * xb + g.m[7] * y + g.m[8];
    double W = W0 + g.m[6] * x1;
    W = W != 0.0 ? 1.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + g.m[0] * x1) * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + g.m[3] * x1) * W));
    const long long X = __double2ll_rn(fX), Y = __double2ll_rn(fY); // round half to even, like cvRound
    float v = 0.0f;
    if (X >= 0 && X < g.sw && Y >= 0 && Y < g.sh) v = g.src[(size_t)Y * g.sp + X];
    g.dst[(size_t)y * g.dp + x] = v;
}

hipError_t launch_warp(const float *src, int sw, int sh, int sp, float *dst, int dw, int dh, int dp,
                       const double minv[9], hipStream_t s)
{
    WarpArgs g{};
    g.src = src; g.sw = sw; g.sh = sh; g.sp = sp;
    g.dst = dst; g.dw = dw; g.dh = dh; g.dp = dp;
    for (int i = 0; i < 9; ++i) g.m[i] = minv[i];
    hipLaunchKernelGGL(ws_warp_kernel, dim3(ceil_div(dw, 256), dh), dim3(256), 0, s, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// consumers of the disparity map (src/Reconstruction/reconstruction.cpp) -- O(H*W) streaming
// ------------------------------------------------------------------------------------------
// cv::blur(src, dst, Size(k,k)): normalised box filter, anchor k/2, BORDER_REFLECT_101, sums in
// double, one multiplication by 1/(k*k), cast to float (OpenCV 4.x CV_32F path, un-vendored).
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ---- the box filter as prefix sums ---------------------------------------------------------
// cv::blur sums a k x k window (k = 500 in the pipeline, main.cpp:53) around every pixel.  Here: per row an
// inclusive prefix P of the row in double (LDS), the window sum of the REFLECT_101-extended row as a difference
// of two prefix values -- the extension is periodic with period 2w - 2 and even, so the prefix F(t) of the
// extended row is (t / T) * G(T) + G(t mod T) with G read off P -- then the same down the columns of the row
// sums.  O(1) per pixel instead of O(k); every operation is a double addition / subtraction of sums that are
// exact whenever the map's values are (integer-valued disparities: the pipeline reads an 8-bit PNG), so the
// result is bit-identical to summing the window directly; for other inputs it is deterministic and differs
// from a direct sum by double rounding only (OpenCV itself keeps running sums, in its own order).
__device__ __forceinline__ double ext_prefix(const double *P, int n, long long t)
{
    // sum of the first t elements (t >= 0) of the periodic even extension of a line of n >= 2 elements,
    // P[r] = sum of the line's first r elements
    // (t < 2^31: the callers pass x0 + k <= n + k.  Windows shorter than the period -- the usual case, k = 500
    // on images a few hundred pixels wide or more -- never divide.)
    const uint32_t T = 2u * (uint32_t)n - 2u, ut = (uint32_t)t;
    const uint32_t q = ut < T ? 0u : ut / T;
    const int r = (int)(ut - q * T);
    const double g = r <= n ? P[r] : P[n] + P[n - 1] - P[2 * n - 1 - r];
    if (q == 0) return g;
    return (double)q * (P[n] + P[n - 1] - P[1]) + g;
}

__device__ __forceinline__ double ext_window(const double *P, int n, int x0, int k)
{
    // sum of the k elements [x0, x0 + k) of the extended line (x0 may be negative, x0 + k > 0)
    if (n == 1) return (double)k * P[1];
    if (x0 >= 0) return ext_prefix(P, n, (long long)x0 + k) - ext_prefix(P, n, x0);
    return (ext_prefix(P, n, (long long)(-x0) + 1) - P[1]) + ext_prefix(P, n, (long long)x0 + k);
}

// block-wide exclusive scan of one double per thread (256 threads), through LDS
__device__ __forceinline__ double block_excl_scan_256(double v, double *wave_tot)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    double before = 0.0;
    for (int i = 0; i < wv; ++i) before += wave_tot[i];
    return before + (incl - v);
}

// one workgroup per row: prefix of the row in LDS, then the k-wide window sums (doubles) of the row
__global__ void __launch_bounds__(256) ws_box_rows_kernel(const float *__restrict__ src, int sp, double *__restrict__ dst,
                                                          int dp, int w, int h, int k)
{
    extern __shared__ double box_lds[]; // P[0 .. w], 4 wave totals, then the row itself (floats, staged coalesced)
    double *P = box_lds, *wave_tot = box_lds + (w + 1);
    float *line = reinterpret_cast<float *>(wave_tot + 4);
    const int y = blockIdx.x, tid = threadIdx.x;
    const float *row = src + (size_t)y * sp;
    for (int i0 = 0; i0 < w; i0 += 8 * 256) { // eight independent loads in flight per thread, then the LDS stores
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * 256 + tid;
            v[j] = i < w ? row[i] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * 256 + tid;
            if (i < w) line[i] = v[j];
        }
    }
    __syncthreads();
    const int per = (w + 255) / 256, i0 = tid * per, i1 = min(i0 + per, w);
    double run = 0.0;
    for (int i = i0; i < i1; ++i) run += (double)line[i];
    const double offs = block_excl_scan_256(run, wave_tot);
    run = offs;
    if (tid == 0) P[0] = 0.0;
    for (int i = i0; i < i1; ++i) {
        run += (double)line[i];
        P[i + 1] = run;
    }
    __syncthreads();
    const int x00 = -(k / 2);
    double *out = dst + (size_t)y * dp;
    for (int x = tid; x < w; x += 256) out[x] = ext_window(P, w, x + x00, k);
}

// One workgroup per band of BW columns, all rows: thread (column c, chunk ch) keeps its chunk of the
// column in registers, the chunk totals meet in LDS, the column's prefix C goes to LDS, then the k-tall
// window sums, the scale and the reference's replacement rule (removeDisparityOutliers,
// reconstruction.cpp:5-18: a pixel outside [thr_back, thr_front] x blurred takes the blurred value).
constexpr int kColMaxPer = 32; // rows per thread at most

template <int BW>
__global__ void __launch_bounds__(1024) ws_outlier_cols_kernel(const double *__restrict__ rows, int rp, float *__restrict__ map,
                                                                int mp, int w, int h, int k, float thr_front, float thr_back)
{
    extern __shared__ double box_lds[]; // C[(h + 1)][BW], then tot[NCH][BW]
    constexpr int NCH = 1024 / BW;
    double *C = box_lds, *tot = box_lds + (size_t)(h + 1) * BW;
    const int c = threadIdx.x % BW, ch = threadIdx.x / BW;
    const int x = blockIdx.x * BW + c;
    const int L = (h + NCH - 1) / NCH;
    const int ya = ch * L, yb = min(ya + L, h);
    const bool live = x < w;
    double v[kColMaxPer];
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < kColMaxPer; ++i) {
        v[i] = 0.0;
        if (live && ya + i < yb) v[i] = rows[(size_t)(ya + i) * rp + x];
    }
#pragma unroll
    for (int i = 0; i < kColMaxPer; ++i) sum += v[i];
    // exclusive scan of the chunk totals down the column: the lanes of a wave hold 64 / BW consecutive chunks of
    // each column (lane = chunk-in-wave * BW + c), then the waves' totals meet in LDS
    constexpr int CPW = 64 / BW; // chunks per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double incl = sum;
#pragma unroll
    for (int off = BW; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane >= 64 - BW) tot[wv * BW + c] = incl; // the wave's last chunk holds its column totals
    __syncthreads();
    double run = incl - sum;
    for (int i = 0; i < wv; ++i) run += tot[i * BW + c];
    (void)CPW;
    if (ch == 0) C[c] = 0.0;
#pragma unroll
    for (int i = 0; i < kColMaxPer; ++i) {
        if (ya + i < yb) {
            run += v[i];
            C[(size_t)(ya + i + 1) * BW + c] = run;
        }
    }
    __syncthreads();
    if (!live) return;
    const double scale = 1.0 / ((double)k * (double)k);
    for (int y = ya; y < yb; ++y) {
        // the column's prefix sits at stride BW: gather the handful of values ext_window needs
        const int y0 = y - k / 2;
        double acc;
        if (h == 1) {
            acc = (double)k * C[BW + c];
        } else {
            auto Pc = [&](int r) { return C[(size_t)r * BW + c]; };
            auto pre = [&](long long t) {
                const uint32_t T = 2u * (uint32_t)h - 2u, ut = (uint32_t)t;
                const uint32_t q = ut < T ? 0u : ut / T;
                const int r = (int)(ut - q * T);
                const double g = r <= h ? Pc(r) : Pc(h) + Pc(h - 1) - Pc(2 * h - 1 - r);
                return q == 0u ? g : (double)q * (Pc(h) + Pc(h - 1) - Pc(1)) + g;
            };
            acc = y0 >= 0 ? pre((long long)y0 + k) - pre(y0) : (pre((long long)(-y0) + 1) - Pc(1)) + pre((long long)y0 + k);
        }
        const float blurred = (float)(acc * scale);
        float *p = map + (size_t)y * mp + x;
        const float d = *p;
        if (d > __fmul_rn(thr_front, blurred) || d < __fmul_rn(thr_back, blurred)) *p = blurred;
    }
}

// the literal O(k) form: lines too long for the LDS prefix (rows beyond 8190 pixels, columns beyond ~9500)
__global__ void __launch_bounds__(256) ws_box_rows_direct_kernel(const float *__restrict__ src, int sp, double *__restrict__ dst,
                                                                 int dp, int w, int h, int k)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const float *row = src + (size_t)y * sp;
    const int x0 = x - k / 2;
    double acc = 0.0;
    for (int i = 0; i < k; ++i) acc += (double)row[reflect101(x0 + i, w)];
    dst[(size_t)y * dp + x] = acc;
}

__global__ void __launch_bounds__(256) ws_outlier_direct_kernel(const double *__restrict__ rows, int rp, float *__restrict__ map,
                                                                int mp, int w, int h, int k, float thr_front, float thr_back)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const int y0 = y - k / 2;
    double acc = 0.0;
    for (int i = 0; i < k; ++i) acc += rows[(size_t)reflect101(y0 + i, h) * rp + x];
    const float blurred = (float)(acc * (1.0 / ((double)k * (double)k)));
    float *p = map + (size_t)y * mp + x;
    const float d = *p;
    if (d > __fmul_rn(thr_front, blurred) || d < __fmul_rn(thr_back, blurred)) *p = blurred;
}

// ---- the same on 8-bit maps, in 32-bit integers ------------------------------------------------
// The pipeline's map is an 8-bit PNG (main.cpp:47-53): every value an integer in [0, 255].  When the row pass
// finds nothing else, all sums are integers below 2^32 (k <= 4000): the prefixes and their differences are taken
// modulo 2^32 (exact, whatever the intermediate wraps), the double pipes stay idle, and blurred =
// (float)((double)sum * scale) is the very expression of the double path on the very same sum.  The intermediate
// is ONE 32-bit word per pixel, row sum (< 2^24) | value << 24: the column pass reads 4 bytes per pixel instead of
// 8 + 4 and never loads the map.  A map with any other value raises `flag` (device) and `host_word` (mapped, for
// the host): the column pass then leaves the map alone and the caller runs the double path on it.
// The words are stored BAND-MAJOR, dst[band][y][BW columns] with BW = 1 << lb: what the column pass of a band
// reads is one contiguous block of h * BW words.
constexpr uint32_t kBoxSumMask = 0x00ffffffu;
constexpr int kBoxMaxK = 4000; // 4000 * 255 < 2^24 and 4000 * 4000 * 255 < 2^32

template <class F>
__device__ __forceinline__ uint32_t ext_prefix_u32(F P, int n, uint32_t ut)
{
    const uint32_t T = 2u * (uint32_t)n - 2u;
    const uint32_t q = ut < T ? 0u : ut / T;
    const int r = (int)(ut - q * T);
    const uint32_t g = r <= n ? P(r) : P(n) + P(n - 1) - P(2 * n - 1 - r);
    if (q == 0) return g;
    return q * (P(n) + P(n - 1) - P(1)) + g;
}

template <class F>
__device__ __forceinline__ uint32_t ext_window_u32(F P, int n, int x0, int k)
{
    if (x0 >= 0 && x0 + k <= n) return P(x0 + k) - P(x0); // no reflection: all but the border pixels when k < n
    if (n == 1) return (uint32_t)k * P(1);
    if (x0 >= 0) return ext_prefix_u32(P, n, (uint32_t)x0 + (uint32_t)k) - ext_prefix_u32(P, n, (uint32_t)x0);
    return (ext_prefix_u32(P, n, (uint32_t)(-x0) + 1u) - P(1)) + ext_prefix_u32(P, n, (uint32_t)(x0 + k));
}

// workgroup i of n -> the unit it works on, so that the workgroups of one XCD (i mod 8: the dispatcher deals
// workgroups round-robin over the 8 XCDs) own CONSECUTIVE units and the two halves of a 128-byte line that two
// neighbouring units touch meet in one L2
__device__ __forceinline__ int xcd_unit(int i, int n)
{
    const int q = n >> 3, r = n & 7, xcd = i & 7, j = i >> 3;
    return xcd * q + min(xcd, r) + j;
}

// inclusive scan over the 64 lanes of a wave, all lanes active: Hillis-Steele inside each row of 16 lanes (DPP
// row_shr, out-of-row sources read as 0), then the row totals across the rows (DPP row_bcast:15 / row_bcast:31)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // lane 15 of rows 0, 2 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // lane 31 -> rows 2, 3
    return x;
}

// Window sum of the REFLECT_101-extended line when the window is no longer than the line (k <= n: one reflection
// at most, on one side), without a branch: the part inside the line, what reflects off the left end (elements
// 1 .. -lo), what reflects off the right end (elements 2n-1-hi .. n-2).  P(r) = sum of the first r elements.
template <class F>
__device__ __forceinline__ uint32_t short_window_u32(F P, int n, int lo, int k, uint32_t p1, uint32_t pn1)
{
    const int hi = lo + k;
    return (P(min(hi, n)) - P(max(lo, 0))) + (P(max(1 - lo, 1)) - p1) + (pn1 - P(n - 1 - max(hi - n, 0)));
}

constexpr int kRowPer = 16; // pixels per thread and pass of the row kernel

template <int NT> // threads of the workgroup that owns one row
__global__ void __launch_bounds__(NT) ws_box_rows_u32_kernel(const float *__restrict__ src, int sp, uint32_t *__restrict__ dst,
                                                             int w, int h, int k, int lb, uint32_t *__restrict__ flag,
                                                             unsigned int *__restrict__ host_word)
{
    constexpr int NW = NT / 64, NTOT = kRowPer * NW; // NTOT <= 64: one wave scans the wave totals
    static_assert(NTOT <= 64, "one wave scans the totals");
    extern __shared__ uint32_t box_lds_u[]; // P[0 .. w], then tot[NTOT + 1]
    uint32_t *P = box_lds_u, *tot = box_lds_u + (w + 1);
    const int y = xcd_unit(blockIdx.x, h), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *row = src + (size_t)y * sp;
    bool odd = false; // a value the 8-bit path cannot carry
    uint32_t carry = 0;
    if (tid == 0) P[0] = 0u;
    // the row's inclusive prefix into P: pixel i sits in thread i % NT, register i / NT -- the coalesced order of the
    // loads is the order of the scan (wave scans in registers, the 16 * NW wave totals through LDS)
    for (int i0 = 0; i0 < w; i0 += kRowPer * NT) { // one pass for rows up to 16 * NT pixels
        float f[kRowPer];
#pragma unroll
        for (int j = 0; j < kRowPer; ++j) {
            const int i = i0 + j * NT + tid;
            f[j] = i < w ? row[i] : 0.0f;
        }
        uint32_t sc[kRowPer];
#pragma unroll
        for (int j = 0; j < kRowPer; ++j) {
            const uint32_t u = (uint32_t)f[j]; // saturating: negative and NaN -> 0, which then fails the comparison
            odd |= !((float)u == f[j]) || u > 255u;
            sc[j] = wave_incl_scan_u32(u);
        }
        if (lane == 63) {
#pragma unroll
            for (int j = 0; j < kRowPer; ++j) tot[j * NW + wv] = sc[j];
        }
        __syncthreads();
        if (wv == 0) {
            const uint32_t t = lane < NTOT ? tot[lane] : 0u;
            const uint32_t incl = wave_incl_scan_u32(t);
            if (lane < NTOT) tot[lane] = incl - t;
            if (lane == 63) tot[NTOT] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kRowPer; ++j) {
            const int i = i0 + j * NT + tid;
            if (i < w) P[i + 1] = carry + tot[j * NW + wv] + sc[j];
        }
        carry += tot[NTOT];
        __syncthreads();
    }
    if (__any(odd) && lane == 0) {
        *flag = 1u;
        *host_word = 1u;
    }
    const int a = k / 2, bw1 = (1 << lb) - 1;
    auto Pf = [&](int r) { return P[r]; };
    auto word_at = [&](int x) { return (((uint32_t)(x >> lb) * (uint32_t)h + (uint32_t)y) << lb) + (uint32_t)(x & bw1); };
    if (k <= w) {
        const uint32_t p1 = P[1], pn1 = P[w - 1];
#pragma unroll 4
        for (int x = tid; x < w; x += NT)
            dst[word_at(x)] = short_window_u32(Pf, w, x - a, k, p1, pn1) | (P[x + 1] - P[x]) << 24;
    } else {
        for (int x = tid; x < w; x += NT) dst[word_at(x)] = ext_window_u32(Pf, w, x - a, k) | (P[x + 1] - P[x]) << 24;
    }
}

constexpr int kColMaxPerU = 36; // rows per thread at most

template <int BW, int NT>
__global__ void __launch_bounds__(NT) ws_outlier_cols_u32_kernel(const uint32_t *__restrict__ rows, float *__restrict__ map,
                                                                  int mp, int w, int h, int k, float thr_front, float thr_back,
                                                                  const uint32_t *__restrict__ flag)
{
    if (*flag) return; // the row pass met a value the 8-bit path cannot carry: the map stays as it is
    extern __shared__ uint32_t box_lds_u[]; // C[(h + 1)][BW], then tot[NT / 64][BW]
    constexpr int NCH = NT / BW, NWV = NT / 64;
    uint32_t *C = box_lds_u, *tot = box_lds_u + (size_t)(h + 1) * BW;
    const int band = xcd_unit(blockIdx.x, gridDim.x);
    const int c = threadIdx.x % BW, ch = threadIdx.x / BW;
    const int x = band * BW + c;
    const int L = (h + NCH - 1) / NCH;
    const int ya = min(ch * L, h), yb = min(ya + L, h);
    const bool live = x < w;
    const int n = yb - ya; // this thread's rows: ya .. ya + n - 1
    const uint32_t *col = rows + (size_t)band * h * BW + c;
    const uint32_t *mine = col + (size_t)ya * BW;
    uint32_t v[kColMaxPerU];
    // everything this thread will read from memory, in flight at once.  No guards: rows past the thread's own are
    // the next thread's or the next band's, past the last band lies the padding outliers_u32_scratch_bytes adds
#pragma unroll
    for (int i = 0; i < kColMaxPerU; ++i) v[i] = mine[i * BW];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kColMaxPerU; ++i) {
        v[i] = i < n ? v[i] : 0u;
        sum += v[i] & kBoxSumMask;
    }
    // exclusive scan of the chunk totals down the column (lane = chunk-in-wave * BW + c), then across the waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = sum;
#pragma unroll
    for (int off = BW; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane >= 64 - BW) tot[wv * BW + c] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
#pragma unroll
    for (int i = 0; i < NWV; ++i) { // all the reads at once, the waves above this one selected afterwards
        const uint32_t t = tot[i * BW + c];
        run += i < wv ? t : 0u;
    }
    if (ch == 0) C[c] = 0u;
    uint32_t *cmine = C + (ya + 1) * BW + c;
#pragma unroll
    for (int i = 0; i < kColMaxPerU; ++i) {
        run += v[i] & kBoxSumMask;
        if (i < n) cmine[i * BW] = run;
    }
    __syncthreads();
    if (!live) return;
    const double scale = 1.0 / ((double)k * (double)k);
    const int a = k / 2;
    auto Pc = [&](int r) { return C[r * BW + c]; };
    float *mrow = map + (size_t)ya * mp + x; // walks down the column with the rows
    auto settle = [&](uint32_t acc, uint32_t word) {
        const float blurred = (float)((double)acc * scale);
        const float d = (float)(word >> 24);
        if (d > __fmul_rn(thr_front, blurred) || d < __fmul_rn(thr_back, blurred)) *mrow = blurred;
        mrow += mp;
    };
    if (k <= h) {
        const uint32_t p1 = Pc(1), pn1 = Pc(h - 1);
#pragma unroll
        for (int i = 0; i < kColMaxPerU; ++i) {
            if (i < n) settle(short_window_u32(Pc, h, ya + i - a, k, p1, pn1), v[i]);
            if (i % 9 == 8) __builtin_amdgcn_sched_barrier(0); // nine rows' LDS reads in flight at a time: registers
        }
    } else { // windows taller than the map (small maps): the periodic form, the words read again
        for (int y = ya; y < yb; ++y) settle(ext_window_u32(Pc, h, y - a, k), col[(size_t)y * BW]);
    }
}

// band width of the column pass: a workgroup of 1024 threads (one per CU: ~100 VGPRs) owns BW columns of all rows;
// narrow bands make more workgroups than CUs (rounds), wide ones leave CUs idle.  Modelled cost per round: a fixed
// latency plus the band's pixels (fitted on 3840x2160 / 1500x1000 / 900x750, profiles/r03/consumers.txt).
static int outliers_u32_band(int w, int h, int num_cus)
{
    constexpr size_t kLdsBudget = 150 * 1024;
    static const char *forced = getenv("WS_BOX_COLS"); // band width: developer knob
    int best = 0;
    double best_cost = 0.0;
    for (int bw : {4, 8, 16}) {
        if ((size_t)(h + 1 + 16) * bw * sizeof(uint32_t) > kLdsBudget || ceil_div(h, 1024 / bw) > kColMaxPerU) continue;
        if (forced && atoi(forced) == bw) return bw;
        const int rounds = ceil_div(ceil_div(w, bw), std::max(1, num_cus));
        const double cost = rounds * (4.0 + 0.43e-3 * (double)h * bw);
        if (!best || cost < best_cost) {
            best = bw;
            best_cost = cost;
        }
    }
    return best;
}

size_t outliers_u32_scratch_bytes(int w, int h, int num_cus)
{
    const int bw = outliers_u32_band(w, h, num_cus);
    // (+ the rows a thread of the column pass may read past the last band's end)
    return bw ? ((size_t)ceil_div(w, bw) * h + kColMaxPerU) * bw * sizeof(uint32_t) : 0;
}

bool outliers_u32_applies(int w, int h, int k, int num_cus)
{
    return outliers_u32_band(w, h, num_cus) != 0 && (size_t)(w + 1 + 80) * sizeof(uint32_t) <= 64 * 1024 && k <= kBoxMaxK &&
           (size_t)(w + 16) * h < (1u << 30);
}

hipError_t launch_outliers_u32(float *map, int mp, int w, int h, int k, float thr_front, float thr_back, uint32_t *scratch,
                               uint32_t *flag, unsigned int *host_word, int num_cus, hipStream_t s)
{
    const int bw = outliers_u32_band(w, h, num_cus);
    if (!bw || !outliers_u32_applies(w, h, k, num_cus)) return hipErrorInvalidValue;
    const int lb = bw == 16 ? 4 : bw == 8 ? 3 : 2;
    // rows: one workgroup of 256 threads per row (several rows per workgroup with the next row fetched during the
    // stores measured slower at every size: profiles/r03/consumers.txt)
    const size_t row_lds = (size_t)(w + 1 + 80) * sizeof(uint32_t);
    hipLaunchKernelGGL(ws_box_rows_u32_kernel<256>, dim3(h), dim3(256), row_lds, s, map, mp, scratch, w, h, k, lb, flag, host_word);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto launch = [&](auto kernel) -> hipError_t {
        const size_t lds = (size_t)(h + 1 + 16) * bw * sizeof(uint32_t);
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kernel, dim3(ceil_div(w, bw)), dim3(1024), lds, s, scratch, map, mp, w, h, k, thr_front, thr_back, flag);
        return hipGetLastError();
    };
    if (bw == 16) return launch(ws_outlier_cols_u32_kernel<16, 1024>);
    if (bw == 8) return launch(ws_outlier_cols_u32_kernel<8, 1024>);
    return launch(ws_outlier_cols_u32_kernel<4, 1024>);
}

hipError_t launch_outliers(float *map, int mp, int w, int h, int k, float thr_front, float thr_back, double *scratch,
                           hipStream_t s)
{
    constexpr size_t kLdsBudget = 150 * 1024;
    // rows: one workgroup per row with the row's prefix in LDS
    const size_t row_lds = (size_t)(w + 1 + 4) * sizeof(double) + (size_t)w * sizeof(float);
    if (row_lds <= 64 * 1024) {
        hipLaunchKernelGGL(ws_box_rows_kernel, dim3(h), dim3(256), row_lds, s, map, mp, scratch, w, w, h, k);
    } else {
        hipLaunchKernelGGL(ws_box_rows_direct_kernel, dim3(ceil_div(w, 256), h), dim3(256), 0, s, map, mp, scratch, w, w, h, k);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // columns: the widest band whose column prefixes fit in LDS (and whose threads hold at most 32 rows each)
    auto fits = [&](int bw) {
        const int nch = 1024 / bw;
        return (size_t)(h + 1 + nch) * bw * sizeof(double) <= kLdsBudget && ceil_div(h, nch) <= kColMaxPer;
    };
    auto launch = [&](auto kernel, int bw) -> hipError_t {
        const size_t lds = (size_t)(h + 1 + 1024 / bw) * bw * sizeof(double);
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kernel, dim3(ceil_div(w, bw)), dim3(1024), lds, s, scratch, w, map, mp, w, h, k, thr_front, thr_back);
        return hipGetLastError();
    };
    // (narrower bands when the wide ones would leave most CUs without a workgroup)
    if (fits(16) && ceil_div(w, 16) >= 200) return launch(ws_outlier_cols_kernel<16>, 16);
    if (fits(8)) return launch(ws_outlier_cols_kernel<8>, 8);
    if (fits(16)) return launch(ws_outlier_cols_kernel<16>, 16);
    if (fits(4)) return launch(ws_outlier_cols_kernel<4>, 4);
    if (fits(2)) return launch(ws_outlier_cols_kernel<2>, 2);
    hipLaunchKernelGGL(ws_outlier_direct_kernel, dim3(ceil_div(w, 256), h), dim3(256), 0, s, scratch, w, map, mp, w, h, k, thr_front, thr_back);
    return hipGetLastError();
}

// convertDisparityToDepth (reconstruction.cpp:30-43) fused with the back-projection of
// reconstruction() (reconstruction.cpp:152-196): depth = f*b/d (0 -> -inf), vertex =
// ((x*depth - cx*depth)/fx, (y*depth - cy*depth)/fy, depth, 1), colour = (R, G, B, 255).
__global__ void __launch_bounds__(256) ws_depth_vertices_kernel(const float *__restrict__ disp, int dp, int w, int h,
                                                                float focal, float baseline, float fx, float fy, float cx,
                                                                float cy, const uint8_t *__restrict__ bgr, int bs,
                                                                float *__restrict__ depth, int zp,
                                                                float4 *__restrict__ pos, uchar4 *__restrict__ col,
                                                                int input_is_depth)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const float d = disp[(size_t)y * dp + x];
    const float minf = -INFINITY;
    const float z = input_is_depth ? d : (d == 0.0f ? minf : __fdiv_rn(__fmul_rn(focal, baseline), d));
    if (depth) depth[(size_t)y * zp + x] = z;
    if (pos) {
        const size_t idx = (size_t)y * w + x;
        if (z == minf) {
            pos[idx] = make_float4(minf, minf, minf, minf);
            col[idx] = make_uchar4(0, 0, 0, 0);
        } else {
            // separately rounded products, as the reference's x86-64 build evaluates them (no FMA)
            const float xc = __fdiv_rn(__fsub_rn(__fmul_rn((float)x, z), __fmul_rn(cx, z)), fx);
            const float yc = __fdiv_rn(__fsub_rn(__fmul_rn((float)y, z), __fmul_rn(cy, z)), fy);
            pos[idx] = make_float4(xc, yc, z, 1.0f);
            const uint8_t *p = bgr + (size_t)y * bs + 3 * x;
            col[idx] = make_uchar4(p[2], p[1], p[0], 255);
        }
    }
}

// convertDisparityToDepth alone, four pixels per thread (16-byte loads and stores) for dense, aligned maps
__global__ void __launch_bounds__(256) ws_depth4_kernel(const float4 *__restrict__ disp, float4 *__restrict__ depth, size_t n4,
                                                        float focal, float baseline)
{
    const float minf = -INFINITY;
    const float fb = __fmul_rn(focal, baseline);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 d = disp[i];
        float4 z;
        z.x = d.x == 0.0f ? minf : __fdiv_rn(fb, d.x);
        z.y = d.y == 0.0f ? minf : __fdiv_rn(fb, d.y);
        z.z = d.z == 0.0f ? minf : __fdiv_rn(fb, d.z);
        z.w = d.w == 0.0f ? minf : __fdiv_rn(fb, d.w);
        depth[i] = z;
    }
}

hipError_t launch_depth_vertices(const float *disp, int dp, int w, int h, float focal, float baseline, const float k[9],
                                 const uint8_t *bgr, int bstride, float *depth, int zp, float *pos, uint8_t *col,
                                 int input_is_depth, hipStream_t s)
{
    const size_t n = (size_t)w * h;
    if (!pos && depth && !input_is_depth && dp == w && zp == w && (n & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(disp) | reinterpret_cast<uintptr_t>(depth)) & 15) == 0) {
        const size_t n4 = n / 4;
        const unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(ws_depth4_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4 *>(disp),
                           reinterpret_cast<float4 *>(depth), n4, focal, baseline);
        return hipGetLastError();
    }
    dim3 grid(ceil_div(w, 256), h);
    hipLaunchKernelGGL(ws_depth_vertices_kernel, grid, dim3(256), 0, s, disp, dp, w, h, focal, baseline, k ? k[0] : 1.0f,
                       k ? k[4] : 1.0f, k ? k[2] : 0.0f, k ? k[5] : 0.0f, bgr, bstride, depth, zp,
                       reinterpret_cast<float4 *>(pos), reinterpret_cast<uchar4 *>(col), input_is_depth);
    return hipGetLastError();
}

} // namespace wsamd
