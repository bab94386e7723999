// ws_prepass.hip -- pack (BGR bytes -> dword planes) and bias (box-summed squares / poison) kernels
// Part of the gfx950 kernels of the WindowSearch hot path; overview in ws_march.hip.
#include "ws_device.h"

namespace wsamd {

// ------------------------------------------------------------------------------------------
// pack: CV_8UC3 rows -> padded dword plane
// ------------------------------------------------------------------------------------------
struct PackArgs { // blockIdx.z selects the image: both planes are packed by one launch
    const uint8_t *src[2];
    uint32_t *dst[2];
    int w[2], h[2], stride[2], pitch[2], pad[2];
    int mirror;
    uint32_t xor_mask; // kCentre for SSD planes, 0 for SAD
};

__global__ void __launch_bounds__(256) ws_pack_kernel(const PackArgs g)
{
    // one thread = 4 consecutive plane columns (one 16-byte store); pitch is a multiple of 4
    const int z = blockIdx.z;
    const int w = g.w[z], h = g.h[z], pitch = g.pitch[z];
    const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (col >= pitch || y >= h) return;
    const int x = col - g.pad[z];
    const uint8_t *row = g.src[z] + (size_t)y * g.stride[z];
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (!g.mirror && x >= 0 && x + 3 < w && ((reinterpret_cast<uintptr_t>(row) + 3 * (size_t)x) & 3) == 0) {
        // 12 bytes = 3 aligned dwords = 4 BGR pixels
        const uint32_t *p = reinterpret_cast<const uint32_t *>(row + 3 * (size_t)x);
        const uint32_t a = p[0], b = p[1], c = p[2];
        v[0] = (a & 0xffffffu) ^ g.xor_mask;
        v[1] = ((a >> 24) | ((b & 0xffffu) << 8)) ^ g.xor_mask;
        v[2] = ((b >> 16) | ((c & 0xffu) << 16)) ^ g.xor_mask;
        v[3] = (c >> 8) ^ g.xor_mask;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int xs = x + k;
            if (xs >= 0 && xs < w) {
                if (g.mirror) xs = w - 1 - xs;
                const uint8_t *p = row + (size_t)xs * 3;
                v[k] = ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)) ^ g.xor_mask;
            }
        }
    }
    *reinterpret_cast<uint4 *>(g.dst[z] + (size_t)y * pitch + col) = make_uint4(v[0], v[1], v[2], v[3]);
}

hipError_t launch_pack(const uint8_t *src_a, int wa, int ha, int stride_a, Plane dst_a,
                       const uint8_t *src_b, int wb, int hb, int stride_b, Plane dst_b, int mirror,
                       int centred, hipStream_t s)
{
    PackArgs g{};
    g.xor_mask = centred ? kCentre : 0u;
    g.src[0] = src_a; g.dst[0] = dst_a.data; g.w[0] = wa; g.h[0] = ha; g.stride[0] = stride_a;
    g.pitch[0] = dst_a.pitch; g.pad[0] = dst_a.pad;
    g.src[1] = src_b; g.dst[1] = dst_b.data; g.w[1] = wb; g.h[1] = hb; g.stride[1] = stride_b;
    g.pitch[1] = dst_b.pitch; g.pad[1] = dst_b.pad;
    g.mirror = mirror;
    dim3 grid(ceil_div(std::max(dst_a.pitch, dst_b.pitch) / 4, 256), std::max(ha, hb), 2);
    hipLaunchKernelGGL(ws_pack_kernel, grid, dim3(256), 0, s, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// bias rows: poison for invalid B centres; for SSD the box-summed squares of B
// ------------------------------------------------------------------------------------------
struct BiasArgs {
    const uint32_t *B;
    int pitch_b, pad_b; // the packed target plane
    int pitch, pad;     // the bias plane (own padding: its row copies must start 16-byte aligned)
    int ww, wh, wx0, wy0;
    int b_lo, b_hi, oy0, oy1;
    int ssd, shift, centred;
    int32_t *bias;
};

constexpr int kBiasRows = 32;  // output rows per workgroup
constexpr int kBiasMaxWh = 17; // tallest window with a marching instantiation

__device__ __forceinline__ uint32_t row_square_sum(const uint32_t *row, int ww, int centred)
{
    uint32_t acc = 0;
    if (centred)
        for (int wx = 0; wx < ww; ++wx) acc = pix_dot<true>(row[wx], row[wx], acc);
    else
        for (int wx = 0; wx < ww; ++wx) acc = pix_dot<false>(row[wx], row[wx], acc);
    return acc;
}

// Separable box filter of the squared target pixels: a workgroup (64 x 4 threads) owns 64 columns
// x kBiasRows rows.  Thread (tx, ty) fills every 4th horizontal sum of column tx in LDS, then
// slides the vertical sum down its quarter of the strip.
__global__ void __launch_bounds__(256) ws_bias_kernel(const BiasArgs g)
{
    __shared__ uint32_t hs[kBiasRows + kBiasMaxWh - 1][64];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int col = blockIdx.x * 64 + tx;
    const int y0 = g.oy0 + blockIdx.y * kBiasRows;
    const int y1 = min(y0 + kBiasRows, g.oy1);
    const int xb = col - g.pad;
    const bool in_plane = col < g.pitch;
    const bool centre_ok = in_plane && xb >= g.b_lo && xb <= g.b_hi;
    if (centre_ok && g.ssd) {
        const int nrows = (y1 - y0) + g.wh - 1;
        const uint32_t *src = g.B + (size_t)(y0 + g.wy0) * g.pitch_b + (xb + g.pad_b + g.wx0);
        for (int k = ty; k < nrows; k += 4) hs[k][tx] = row_square_sum(src + (size_t)k * g.pitch_b, g.ww, g.centred);
    }
    __syncthreads();
    if (!in_plane) return;
    const int seg = kBiasRows / 4;
    const int ya = y0 + ty * seg, yb = min(ya + seg, y1);
    if (ya >= yb) return;
    int32_t *dst = g.bias + (size_t)ya * g.pitch + col;
    if (!centre_ok || !g.ssd) {
        const int32_t v = centre_ok ? 0 : kPoison;
        for (int y = ya; y < yb; ++y, dst += g.pitch) *dst = v;
        return;
    }
    uint32_t acc = 0;
    for (int k = 0; k < g.wh; ++k) acc += hs[ya - y0 + k][tx];
    for (int y = ya; y < yb; ++y, dst += g.pitch) {
        *dst = (int32_t)(acc << g.shift);
        const int k = y - y0;
        if (y + 1 < yb) acc += hs[k + g.wh][tx] - hs[k][tx];
    }
}

hipError_t launch_bias(const Canon &c, const MarchLaunch &m, Plane b, Plane bias, hipStream_t s)
{
    BiasArgs g{};
    g.B = b.data;
    g.pitch_b = b.pitch;
    g.pad_b = b.pad;
    g.pitch = bias.pitch;
    g.pad = bias.pad;
    g.ww = c.ww;
    g.wh = c.wh;
    g.wx0 = c.wx0;
    g.wy0 = c.wy0;
    g.b_lo = c.b_lo;
    g.b_hi = c.b_hi;
    g.oy0 = c.oy0;
    g.oy1 = c.oy1;
    g.ssd = c.ssd;
    g.shift = ilog2c(m.nd_per_thread);
    g.centred = march_centred(c);
    g.bias = reinterpret_cast<int32_t *>(bias.data);
    dim3 grid(ceil_div(bias.pitch, 64), ceil_div(c.oy1 - c.oy0, kBiasRows));
    hipLaunchKernelGGL(ws_bias_kernel, grid, dim3(64, 4), 0, s, g);
    return hipGetLastError();
}

} // namespace wsamd
