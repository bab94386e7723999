// ws_prepass.hip -- pack: BGR bytes -> padded dword planes, for the kernels BESIDE the hot one that still read planes
// (right view's border ring, sub-pixel refine, the smoothFactor passes).  The marching kernel itself stages the caller's
// bytes (ws_march_kernel.h); a left-view search with smoothFactor 1 launches nothing but that kernel.  Rounds 1-3 also
// summed the SSD "bias" plane here (8.7 MB in, 20.8 MB out per config-2 pair): it is computed inside the marching
// kernel now.  Part of the gfx950 kernels of the WindowSearch hot path; overview in ws_march.hip.
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

constexpr int kPackRows = 4; // rows per thread: independent loads in flight, a quarter of the workgroups

__device__ __forceinline__ void pack_block(const PackArgs &g, int z, int bx, int by)
{
    // one thread = 4 consecutive plane columns (one 16-byte store) of kPackRows rows; pitch is a multiple of 4
    const int w = g.w[z], h = g.h[z], pitch = g.pitch[z];
    const int col = (bx * 256 + (int)threadIdx.x) * 4;
    if (col >= pitch) return;
    const int x = col - g.pad[z];
    const int y0 = by * kPackRows;
    uint32_t v[kPackRows][4];
    const bool quad = !g.mirror && x >= 0 && x + 3 < w;
#pragma unroll
    for (int r = 0; r < kPackRows; ++r) {
        v[r][0] = v[r][1] = v[r][2] = v[r][3] = 0u;
        const int y = y0 + r;
        if (y >= h) continue;
        const uint8_t *row = g.src[z] + (size_t)y * g.stride[z];
        if (quad && ((reinterpret_cast<uintptr_t>(row) + 3 * (size_t)x) & 3) == 0) {
            // 12 bytes = 3 aligned dwords = 4 BGR pixels
            const uint32_t *p = reinterpret_cast<const uint32_t *>(row + 3 * (size_t)x);
            const uint32_t a = p[0], b = p[1], c = p[2];
            v[r][0] = (a & 0xffffffu) ^ g.xor_mask;
            v[r][1] = ((a >> 24) | ((b & 0xffffu) << 8)) ^ g.xor_mask;
            v[r][2] = ((b >> 16) | ((c & 0xffu) << 16)) ^ g.xor_mask;
            v[r][3] = (c >> 8) ^ g.xor_mask;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int xs = x + k;
                if (xs >= 0 && xs < w) {
                    if (g.mirror) xs = w - 1 - xs;
                    const uint8_t *p = row + (size_t)xs * 3;
                    v[r][k] = ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)) ^ g.xor_mask;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kPackRows; ++r)
        if (y0 + r < h)
            *reinterpret_cast<uint4 *>(g.dst[z] + (size_t)(y0 + r) * pitch + col) = make_uint4(v[r][0], v[r][1], v[r][2], v[r][3]);
}

// both planes in one launch of independent workgroups
struct PackLaunchArgs {
    PackArgs pack;
    int pack_gx[2], n_pack[2];
};

__global__ void __launch_bounds__(256) ws_pack_kernel(const PackLaunchArgs g)
{
    int b = blockIdx.x;
    if (b < g.n_pack[0]) { pack_block(g.pack, 0, b % g.pack_gx[0], b / g.pack_gx[0]); return; }
    b -= g.n_pack[0];
    pack_block(g.pack, 1, b % g.pack_gx[1], b / g.pack_gx[1]);
}

hipError_t launch_pack(const Canon &c, const uint8_t *src_a, int stride_a, Plane dst_a, const uint8_t *src_b, int stride_b,
                       Plane dst_b, hipStream_t s)
{
    PackLaunchArgs g{};
    const int centred = march_centred(c);
    g.pack.xor_mask = centred ? kCentre : 0u;
    g.pack.src[0] = src_a; g.pack.dst[0] = dst_a.data; g.pack.w[0] = c.wa; g.pack.h[0] = c.ha; g.pack.stride[0] = stride_a;
    g.pack.pitch[0] = dst_a.pitch; g.pack.pad[0] = dst_a.pad;
    g.pack.src[1] = src_b; g.pack.dst[1] = dst_b.data; g.pack.w[1] = c.wb; g.pack.h[1] = c.hb; g.pack.stride[1] = stride_b;
    g.pack.pitch[1] = dst_b.pitch; g.pack.pad[1] = dst_b.pad;
    g.pack.mirror = c.mirror;
    g.pack_gx[0] = ceil_div(dst_a.pitch / 4, 256); g.n_pack[0] = g.pack_gx[0] * ceil_div(c.ha, kPackRows);
    g.pack_gx[1] = ceil_div(dst_b.pitch / 4, 256); g.n_pack[1] = g.pack_gx[1] * ceil_div(c.hb, kPackRows);
    hipLaunchKernelGGL(ws_pack_kernel, dim3((unsigned)(g.n_pack[0] + g.n_pack[1])), dim3(256), 0, s, g);
    return hipGetLastError();
}

} // namespace wsamd
