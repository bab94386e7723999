// ws_prepass.hip -- one launch before the marching kernel: pack (BGR bytes -> dword planes), bias
// (box-summed squares / poison) and, for the left view, the pixels outside the marching interior
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

// ------------------------------------------------------------------------------------------
// bias rows (SSD): poison for invalid B centres, the box-summed squares of B, and -- for the fused
// marching kernel -- the correction term of its complemented leaving rows
// ------------------------------------------------------------------------------------------
struct BiasArgs {
    const uint8_t *src; // the target image's bytes (the pack of the same launch may not have run yet)
    int wb, hb, stride, mirror;
    uint32_t xor_mask;
    int pitch, pad; // the bias plane (own padding: its row copies must start 16-byte aligned)
    int ww, wh, wx0, wy0;
    int b_lo, b_hi, oy0, oy1;
    int shift, centred;
    int strip_rows;  // the marching kernel's strips: output rows [oy0 + s * strip_rows, ...)
    int kmul;        // fused kernel: 2 K (510 for plain bytes, -2 for centred ones, 0: not fused), see march_fused_ssd
    int32_t *bias;
};

#ifndef WS_BIAS_ROWS
#define WS_BIAS_ROWS 16
#endif
constexpr int kBiasRows = WS_BIAS_ROWS; // (round 3, a workgroup walking a strip chunk by chunk: 16 / 24 / 32 / 48 -> 2.91 / 2.86 / 2.77 / 2.77 * 10^6 Mdisp/s in flight at config 2) output rows per chunk (8 / 16 / 32 measured at config 2 in round 2: 16.3 / 14.6 / 15.5 us for the launch)
constexpr int kBiasMaxWh = 17; // tallest (and widest) window with a marching instantiation
constexpr int kBiasStage = kBiasRows + kBiasMaxWh - 1;

struct BiasLds {
    uint32_t raw[kBiasStage][64];              // the rows' bytes as aligned dwords; then the horizontal sums of the squares
    uint32_t sq[kBiasStage][64 + kBiasMaxWh + 3];  // squares of the pixels (3 channels summed); rows 16-byte aligned
    uint32_t cs[kBiasStage][64 + kBiasMaxWh + 3];  // channel sums of the pixels (fused kernel)
    uint32_t hb[kBiasStage][64];               // their horizontal sums over the window
};

// A workgroup (64 x 4 threads) owns 64 columns of one STRIP of the marching kernel and walks down it in chunks of
// kBiasRows output rows.  Per chunk: the squares (and channel sums) of its pixels + window halo go to LDS straight
// from the image bytes, then the horizontal window sums, then thread (tx, ty) slides the vertical sum down its
// quarter of the chunk.  The fused marching kernel (march_fused_ssd) leaves K * (sum over the window columns of the
// target bytes of every row that LEFT the window since the strip's first step) in its running sums; the bias entry of
// output row y carries 2 K E(y) with E(y) = that sum over the window rows of the strip above y's window, i.e. rows
// [ys + wy0, y + wy0): a prefix down the strip, carried from chunk to chunk in a register (e0).  All modulo 2^32, like
// the kernel's own sums; bias + V is exact.  Invalid centres carry the term too (poison + 2 K E): the kernel's V holds
// it whether or not the candidate is valid.
__device__ __forceinline__ void bias_strip(const BiasArgs &g, int bx, int strip, BiasLds &l)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = bx * 64 + tx;
    const int xb = col - g.pad;
    const bool in_plane = col < g.pitch;
    const bool centre_ok = in_plane && xb >= g.b_lo && xb <= g.b_hi;
    const int ys = g.oy0 + strip * g.strip_rows, ye = min(ys + g.strip_rows, g.oy1);
    const int ncols = 64 + g.ww - 1;
    const int xc0 = bx * 64 - g.pad + g.wx0; // canonical column of sq[.][0]
    // the image columns behind sq[.][0 .. ncols): one contiguous byte range per row
    const int ca = max(xc0, 0), cb = min(xc0 + ncols, g.wb); // canonical [ca, cb)
    const int xs_lo = g.mirror ? g.wb - cb : ca;             // first image column of the range
    uint32_t (*hs)[64] = l.raw;
    uint32_t e0 = 0; // 2 K E of the chunk's first output row, this thread's column
    for (int y0 = ys; y0 < ye; y0 += kBiasRows) {
        const int y1 = min(y0 + kBiasRows, ye);
        const int nrows = (y1 - y0) + g.wh - 1;
        constexpr int kMaxTrips = (kBiasStage + 3) / 4;
        // (1) the rows' bytes as aligned dwords, every load of a thread in flight at once
        uint32_t ld[kMaxTrips];
#pragma unroll
        for (int t = 0; t < kMaxTrips; ++t) {
            const int k = ty + 4 * t, yy = y0 + g.wy0 + k;
            ld[t] = 0;
            if (k < nrows && yy >= 0 && yy < g.hb && cb > ca) {
                const uintptr_t first = reinterpret_cast<uintptr_t>(g.src + (size_t)yy * g.stride + 3 * (size_t)xs_lo);
                const uintptr_t a0 = first & ~(uintptr_t)3;
                const int ndw = (int)((first + 3 * (size_t)(cb - ca) + 3 - a0) >> 2); // an aligned dword holding an image
                if (tx < ndw) ld[t] = reinterpret_cast<const uint32_t *>(a0)[tx];    // byte stays inside its page
            }
        }
        if (y0 > ys) __syncthreads(); // the previous chunk's sums are read
#pragma unroll
        for (int t = 0; t < kMaxTrips; ++t) {
            const int k = ty + 4 * t;
            if (k < nrows) l.raw[k][tx] = ld[t];
        }
        __syncthreads();
        // (2) squares and channel sums of the pixels: the nrows x ncols items dealt flat over the 256 threads (by rows
        // of 64 lanes the 17 halo columns cost a whole second pass with a quarter of its lanes at work)
        {
            const uint32_t items = (uint32_t)(nrows * ncols), magic = 0xffffffffu / (uint32_t)ncols + 1u;
            const uint32_t base_lo = (uint32_t)reinterpret_cast<uintptr_t>(g.src) + 3u * (uint32_t)xs_lo;
            for (uint32_t it = threadIdx.x; it < items; it += 256) {
                const int k = (int)__umulhi(it, magic), cx = (int)it - k * ncols; // (exact: it < 2^16, ncols <= 80)
                const int yy = y0 + g.wy0 + k, xc = xc0 + cx;
                uint32_t v = 0, c = 0;
                if (yy >= 0 && yy < g.hb && xc >= ca && xc < cb) {
                    const uint32_t sh0 = (base_lo + (uint32_t)yy * (uint32_t)g.stride) & 3u;
                    const uint32_t off = sh0 + 3u * (uint32_t)(g.mirror ? cb - 1 - xc : xc - ca); // byte offset in raw[k]
                    const uint32_t lo = l.raw[k][off >> 2], hi = l.raw[k][(off >> 2) + 1];
                    const uint32_t px = (__builtin_amdgcn_alignbyte(hi, lo, off & 3u) & 0xffffffu) ^ g.xor_mask;
                    v = g.centred ? pix_dot<true>(px, px, 0u) : pix_dot<false>(px, px, 0u);
                    if (g.kmul) c = g.centred ? pix_dot<true>(px, 0x00010101u, 0u) : pix_dot<false>(px, 0x00010101u, 0u);
                }
                l.sq[k][cx] = v;
                l.cs[k][cx] = c;
            }
        }
        __syncthreads();
        // horizontal window sums.  The BASELINE windows (7, 9): a thread takes 4 consecutive columns of a row, reads their
        // 4 + ww - 1 values as three 16-byte words and forms the 4 sums from a running prefix (3.5 adds per sum instead of
        // ww, a ninth of the LDS reads); other widths: a thread per (row, column), ww reads each.
        auto hsum8 = [&](auto wwc) __attribute__((always_inline)) {
            constexpr int WWC = decltype(wwc)::value;
            static_assert(4 + WWC - 1 <= 12, "three 16-byte reads");
            // (one plane after the other, four columns a task: the kernel must stay within 48 VGPRs to share a SIMD with
            // the marching kernel's two waves, which is where a queue of pairs hides this pre-pass)
            auto one = [&](auto plane, int k, int x0) __attribute__((always_inline)) {
                constexpr bool SQ = decltype(plane)::value; // (the arrays named directly: LDS addresses stay 32-bit)
                uint32_t q[12];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const uint4 a = SQ ? *reinterpret_cast<const uint4 *>(&l.sq[k][x0 + 4 * m]) : *reinterpret_cast<const uint4 *>(&l.cs[k][x0 + 4 * m]);
                    q[4 * m] = a.x; q[4 * m + 1] = a.y; q[4 * m + 2] = a.z; q[4 * m + 3] = a.w;
                }
#pragma unroll
                for (int m = 1; m < 4 + WWC - 1; ++m) q[m] += q[m - 1]; // running prefix in place
                const uint4 o = make_uint4(q[WWC - 1], q[WWC] - q[0], q[WWC + 1] - q[1], q[WWC + 2] - q[2]);
                if (SQ) *reinterpret_cast<uint4 *>(&l.raw[k][x0]) = o; // (= hs)
                else *reinterpret_cast<uint4 *>(&l.hb[k][x0]) = o;
            };
            for (int task = threadIdx.x; task < nrows * 16; task += 256) {
                const int k = task >> 4, x0 = (task & 15) * 4;
                one(std::true_type(), k, x0);
                __builtin_amdgcn_sched_barrier(0); // (not both planes' values in registers at once)
                if (g.kmul) one(std::false_type(), k, x0);
            }
        };
        if (g.ww == 7) {
            hsum8(std::integral_constant<int, 7>());
        } else if (g.ww == 9) {
            hsum8(std::integral_constant<int, 9>());
        } else {
#pragma unroll 1
            for (int k = ty; k < nrows; k += 4) {
                uint32_t acc = 0, bcc = 0;
#pragma unroll 2
                for (int i = 0; i < g.ww; ++i) { acc += l.sq[k][tx + i]; bcc += l.cs[k][tx + i]; }
                hs[k][tx] = acc;
                l.hb[k][tx] = bcc;
            }
        }
        __syncthreads();
        // (3) down the chunk
        const int seg = kBiasRows / 4;
        const int ya = y0 + ty * seg, yb = min(ya + seg, y1);
        if (in_plane && ya < yb) {
            int32_t *dst = g.bias + (size_t)ya * g.pitch + col;
            uint32_t e = e0;
            if (g.kmul) {
#pragma unroll 1
                for (int k = 0; k < ya - y0; ++k) e += (uint32_t)g.kmul * l.hb[k][tx];
            }
            uint32_t acc = 0;
            if (centre_ok) {
#pragma unroll 1
                for (int k = 0; k < g.wh; ++k) acc += hs[ya - y0 + k][tx];
            }
#pragma unroll 1
            for (int y = ya; y < yb; ++y, dst += g.pitch) {
                const int k = y - y0;
                *dst = centre_ok ? (int32_t)((acc + e) << g.shift) : (int32_t)((uint32_t)kPoison + (e << g.shift));
                if (centre_ok && y + 1 < yb) acc += hs[k + g.wh][tx] - hs[k][tx];
                if (g.kmul) e += (uint32_t)g.kmul * l.hb[k][tx];
            }
        }
        if (g.kmul) {
#pragma unroll 1
            for (int k = 0; k < y1 - y0; ++k) e0 += (uint32_t)g.kmul * l.hb[k][tx];
        }
    }
}

// ------------------------------------------------------------------------------------------
// One launch for everything the marching kernel needs and everything beside it that only reads
// the images: the first nBias workgroups sum the bias rows, the next nA pack the reference plane, then
// nB the target plane, the rest (left view) write the pixels outside the marching interior.
// The four jobs are independent, so the small ones fill the CUs together instead of one after
// the other.
// ------------------------------------------------------------------------------------------
struct PrepareArgs {
    PackArgs pack;
    BiasArgs bias;
    GenericArgs generic;
    int pack_gx[2], n_pack[2];
    int bias_gx, n_bias;
    int n_generic;
};

// (Kept within 48 VGPRs -- loops of the bias role not unrolled, its sums four columns at a time: beside the marching
// kernel's two waves of up to 232 registers a SIMD has 48 left, and that is where the pre-pass of the next pair of a
// queue runs, under the current pair's search.  The round-3 strip walk first took 93 and the overlap was gone.)
__global__ void __launch_bounds__(256) ws_prepare_kernel(const PrepareArgs g)
{
    // static LDS bounds the occupancy of every role (37 KB: four workgroups a CU): the raw bytes ((64 + 16) * 3 +
    // alignment slack < 256 per row) are dead once the squares exist, the horizontal sums reuse their space
    __shared__ BiasLds lds;
    int b = blockIdx.x; // the longest-running workgroups first
    if (b < g.n_bias) { bias_strip(g.bias, b % g.bias_gx, b / g.bias_gx, lds); return; }
    b -= g.n_bias;
    if (b < g.n_pack[0]) { pack_block(g.pack, 0, b % g.pack_gx[0], b / g.pack_gx[0]); return; }
    b -= g.n_pack[0];
    if (b < g.n_pack[1]) { pack_block(g.pack, 1, b % g.pack_gx[1], b / g.pack_gx[1]); return; }
    b -= g.n_pack[1];
    generic_pixel(g.generic, (long long)b * 256 + threadIdx.x);
}

hipError_t launch_prepare(const Canon &c, const MarchLaunch &m, const uint8_t *src_a, int stride_a, Plane dst_a,
                          const uint8_t *src_b, int stride_b, Plane dst_b, Plane bias, const GenericArgs *generic,
                          hipStream_t s)
{
    PrepareArgs g{};
    const int centred = march_centred(c);
    g.pack.xor_mask = centred ? kCentre : 0u;
    g.pack.src[0] = src_a; g.pack.dst[0] = dst_a.data; g.pack.w[0] = c.wa; g.pack.h[0] = c.ha; g.pack.stride[0] = stride_a;
    g.pack.pitch[0] = dst_a.pitch; g.pack.pad[0] = dst_a.pad;
    g.pack.src[1] = src_b; g.pack.dst[1] = dst_b.data; g.pack.w[1] = c.wb; g.pack.h[1] = c.hb; g.pack.stride[1] = stride_b;
    g.pack.pitch[1] = dst_b.pitch; g.pack.pad[1] = dst_b.pad;
    g.pack.mirror = c.mirror;
    g.pack_gx[0] = ceil_div(dst_a.pitch / 4, 256); g.n_pack[0] = g.pack_gx[0] * ceil_div(c.ha, kPackRows);
    g.pack_gx[1] = ceil_div(dst_b.pitch / 4, 256); g.n_pack[1] = g.pack_gx[1] * ceil_div(c.hb, kPackRows);
    if (c.ssd) {
        BiasArgs &bi = g.bias;
        bi.src = src_b; bi.wb = c.wb; bi.hb = c.hb; bi.stride = stride_b; bi.mirror = c.mirror;
        bi.xor_mask = g.pack.xor_mask;
        bi.pitch = bias.pitch; bi.pad = bias.pad;
        bi.ww = c.ww; bi.wh = c.wh; bi.wx0 = c.wx0; bi.wy0 = c.wy0;
        bi.b_lo = c.b_lo; bi.b_hi = c.b_hi; bi.oy0 = c.oy0; bi.oy1 = c.oy1;
        bi.shift = ilog2c(m.nd_per_thread); bi.centred = centred;
        bi.strip_rows = m.strip_rows;
        bi.kmul = march_fused(c) ? (centred ? -2 : 510) : 0;
        bi.bias = reinterpret_cast<int32_t *>(bias.data);
        if (c.ww > kBiasMaxWh || c.wh > kBiasMaxWh) return hipErrorInvalidValue;
        g.bias_gx = ceil_div(bias.pitch, 64);
        g.n_bias = g.bias_gx * m.strips; // (a workgroup walks down one strip of the marching kernel)
    }
    if (generic) {
        g.generic = *generic;
        const int ow = generic->view == 0 ? generic->w1 : generic->w2, oh = generic->view == 0 ? generic->h1 : generic->h2;
        const long long inside = (long long)(generic->skip_x1 - generic->skip_x0) * (generic->skip_y1 - generic->skip_y0);
        const long long n = (long long)ow * oh - (inside > 0 ? inside : 0);
        g.n_generic = n > 0 ? (int)((n + 255) / 256) : 0;
    }
    const long long total = (long long)g.n_pack[0] + g.n_pack[1] + g.n_bias + g.n_generic;
    hipLaunchKernelGGL(ws_prepare_kernel, dim3((unsigned)total), dim3(256), 0, s, g);
    return hipGetLastError();
}

} // namespace wsamd
