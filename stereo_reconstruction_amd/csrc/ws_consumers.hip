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

__global__ void __launch_bounds__(256) ws_box_rows_kernel(const float *__restrict__ src, int sp, double *__restrict__ dst,
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

// column sums of the row sums, scale, and the reference's replacement rule
// (removeDisparityOutliers, reconstruction.cpp:5-18)
__global__ void __launch_bounds__(256) ws_outlier_kernel(const double *__restrict__ rows, int rp, float *__restrict__ map,
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

hipError_t launch_outliers(float *map, int mp, int w, int h, int k, float thr_front, float thr_back, double *scratch,
                           hipStream_t s)
{
    dim3 grid(ceil_div(w, 256), h);
    hipLaunchKernelGGL(ws_box_rows_kernel, grid, dim3(256), 0, s, map, mp, scratch, w, w, h, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ws_outlier_kernel, grid, dim3(256), 0, s, scratch, w, map, mp, w, h, k, thr_front, thr_back);
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

hipError_t launch_depth_vertices(const float *disp, int dp, int w, int h, float focal, float baseline, const float k[9],
                                 const uint8_t *bgr, int bstride, float *depth, int zp, float *pos, uint8_t *col,
                                 int input_is_depth, hipStream_t s)
{
    dim3 grid(ceil_div(w, 256), h);
    hipLaunchKernelGGL(ws_depth_vertices_kernel, grid, dim3(256), 0, s, disp, dp, w, h, focal, baseline, k ? k[0] : 1.0f,
                       k ? k[4] : 1.0f, k ? k[2] : 0.0f, k ? k[5] : 0.0f, bgr, bstride, depth, zp,
                       reinterpret_cast<float4 *>(pos), reinterpret_cast<uchar4 *>(col), input_is_depth);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) ws_widen_kernel(const float *__restrict__ src, int sp,
                                                       double *__restrict__ dst, int dp, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x < w && y < h) dst[(size_t)y * dp + x] = (double)src[(size_t)y * sp + x];
}

hipError_t launch_widen(const float *src, int src_pitch, double *dst, int dst_pitch, int w, int h,
                        hipStream_t s)
{
    dim3 grid(ceil_div(w, 256), h);
    hipLaunchKernelGGL(ws_widen_kernel, grid, dim3(256), 0, s, src, src_pitch, dst, dst_pitch, w, h);
    return hipGetLastError();
}

} // namespace wsamd
