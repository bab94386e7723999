// ws_device.h -- small device / host helpers shared by the kernel translation units.
#pragma once

#include "ws_kernels.h"

#include <limits.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace wsamd {

__host__ __device__ constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int round_up_dev(int v, int m) { return (v + m - 1) / m * m; }

// LDS regions of the target-image rows: a thread's run there starts at quad (X/4)*r + (ND/4)*k, so the
// region of its m-th quad is only fixed at compile time if the region count divides both
__host__ __device__ constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }
__host__ __device__ constexpr int march_nreg_b(int x, int nd) { return gcd_c(x / 4, nd / 4); }

// dwords per LDS region for a row of n dwords split into nreg regions (+1 quad: runs may over-read)
__host__ __device__ constexpr int march_region_dwords(int n, int nreg)
{
    return 4 * ((((n + 3) >> 2) + nreg - 1) / nreg + 1);
}

// ---- the marching kernel's LDS (one definition for the kernel and for the planner that prices it) ----------------
// [stage area, compile-time offsets: 2 raw rows of the reference image | 2 raw rows of the target image | SSD: 2 rows of
//  column sums]  then  the stages' descriptors | ring A | ring A complemented (SSD) | ring B | 2 bias rows (SSD) | 2 rows of merge slots.
// The stage area is sized for the widest tile row any plan makes, so that its addresses are immediates in the code of the
// stages (ws_march_kernel.h, produce) instead of scalar registers held across the chains.
// bytes a raw row buffer needs (raw_dma): up to 15 of alignment in front, 3 per pixel, a dword of over-read behind
__host__ __device__ constexpr int march_raw_bytes(int n4) { return ((16 + 3 * n4 + 16 + 15) >> 4) << 4; }
constexpr int kStageThreads = 512; // the plans' workgroup limit (ws_march_kernel.h: kMaxT)
// widest rows: nxr * nch <= 512 threads, nxr >= 4 x-runs of 8 columns, 8 <= nch <= 64 d-chunks, windows up to 17 wide
constexpr int kStageMaxNA = 8 * (kStageThreads / 8) + 16 + 4;
__host__ __device__ constexpr int march_max_nb(int nd)
{
    return (((8 * (kStageThreads / 8) + 8 * nd) > (64 + 64 * nd) ? (8 * (kStageThreads / 8) + 8 * nd) : (64 + 64 * nd)) + 17 + 3 + 3) & ~3;
}
__host__ __device__ constexpr int march_stage_bytes(int nd, bool ssd)
{
    return 2 * march_raw_bytes(kStageMaxNA) + 2 * march_raw_bytes(march_max_nb(nd)) + (ssd ? 2 * 4 * (march_max_nb(nd) + 16) : 0);
}
// The stages' work is dealt to whole waves as ROLES: image B's quads first -- SSD: 16 - NQN owner lanes per row of 16
// lanes (the row's last NQN lanes only feed the DPP fetches of their neighbours, NQN = the quads to the right whose
// column sums a quad's bias values need), SAD: 64 per wave -- then image A's, 64 per wave.
__host__ __device__ constexpr int march_nqn(int ww, bool ssd) { return ssd ? (ww + 2) / 4 : 0; }
struct MarchLds {
    int n_a, n_b, n_bi;   // pixels of a row the tile's threads read: reference image, target image, bias values
    int n_a4, n_b4;       // ... rounded up to quads (what is unpacked)
    int a_w, b_w, bi_w;   // dwords per ring row (region layout)
    int nr;               // ring rows
    int roles_b, roles_a; // whole waves of stage work per step
    int nslots;           // roles a wave takes at most (a 16-byte descriptor per thread and slot)
    int desc_bytes;
    int bytes;            // 0: a row wider than the stage area
};
__host__ __device__ inline MarchLds march_lds_layout(int x, int nd, int ww, int wh, bool ssd, bool short_runs, int nxr, int nch)
{
    MarchLds l{};
    const int tx = nxr * x, dt = nch * nd;
    const int nreg = x / 4, nregb = march_nreg_b(x, nd);
    // (short_runs: the packed SAD halo-exchange kernels, whose threads read their own columns only)
    l.n_a = short_runs ? tx : tx + ww - 1;
    l.n_b = short_runs ? tx + dt - 1 : tx + ww + dt - 2;
    l.n_bi = tx + dt - 1;
    l.n_a4 = (l.n_a + 3) & ~3;
    l.n_b4 = (l.n_b + 3) & ~3;
    l.a_w = nreg * march_region_dwords(l.n_a, nreg);
    l.b_w = nregb * march_region_dwords(l.n_b, nregb);
    l.bi_w = ssd ? nregb * march_region_dwords(l.n_bi, nregb) : 0;
    l.nr = wh + 2;
    const int threads = (nxr * nch + 63) / 64 * 64, nwaves = threads / 64;
    const int qw = ssd ? 4 * (16 - march_nqn(ww, true)) : 64;
    l.roles_b = (l.n_b4 / 4 + qw - 1) / qw;
    l.roles_a = (l.n_a4 / 4 + 63) / 64;
    l.nslots = (l.roles_b + l.roles_a + nwaves - 1) / nwaves;
    l.desc_bytes = 16 * threads * l.nslots;
    // (SSD: ring A twice -- the reference rows as they are and complemented, what the fused chain multiplies a leaving row by)
    l.bytes = march_stage_bytes(nd, ssd) + l.desc_bytes + 4 * (l.nr * ((ssd ? 2 : 1) * l.a_w + l.b_w) + 2 * l.bi_w) + 2 * tx * (ssd ? 8 : 4);
    if (l.n_a4 > kStageMaxNA || l.n_b4 > march_max_nb(nd)) l.bytes = 0;
    return l;
}

__device__ __forceinline__ uint32_t pix_sad(uint32_t a, uint32_t b, uint32_t acc)
{
    return __builtin_amdgcn_sad_u8(a, b, acc); // v_sad_u8: acc + sum |a.b[i] - b.b[i]|
}
// SSD cross products.  Windows up to 9x9 use the bytes as they are (v_dot4_u32_u8).  Larger
// windows would overflow the 32-bit keys, so their planes hold centred pixels (byte - 128, i.e.
// byte ^ 0x80, 4th byte 0) multiplied by v_dot4_i32_i8: the differences and hence the SSD are
// unchanged, the products are 4x smaller.  (The signed form measured 18 % slower on MI355X, so it
// is only used where it is needed.)
template <bool CENTRED>
__device__ __forceinline__ uint32_t pix_dot(uint32_t a, uint32_t b, uint32_t acc)
{
    if constexpr (CENTRED)
        return (uint32_t)__builtin_amdgcn_sdot4((int)a, (int)b, (int)acc, false); // v_dot4_i32_i8
    else
        return __builtin_amdgcn_udot4(a, b, acc, false); // v_dot4_u32_u8
}
constexpr uint32_t kCentre = 0x00808080u;
// does an SSD window of ww x wh need centred pixels to keep |key| < 2^28 with nd tags per thread?
__host__ __device__ constexpr bool ssd_needs_centring(int ww, int wh, int nd)
{
    return 2LL * ww * wh * 3 * 255 * 255 * nd >= (1LL << 28);
}


__device__ __forceinline__ uint32_t window_cost(const uint8_t *a, int sa, const uint8_t *b, int sb,
                                                int ww, int wh, int ssd)
{
    uint32_t acc = 0;
    for (int r = 0; r < wh; ++r) {
        const uint8_t *pa = a + (size_t)r * sa;
        const uint8_t *pb = b + (size_t)r * sb;
        for (int i = 0; i < 3 * ww; ++i) {
            const int d = (int)pa[i] - (int)pb[i];
            acc += ssd ? (uint32_t)(d * d) : (uint32_t)(d < 0 ? -d : d);
        }
    }
    return acc;
}

__device__ __forceinline__ unsigned long long window_cost64(const uint8_t *a, int sa, const uint8_t *b, int sb,
                                                            int ww, int wh, int ssd)
{
    unsigned long long acc = 0; // grown (varBlock) windows can exceed 32 bits
    for (int r = 0; r < wh; ++r) {
        const uint8_t *pa = a + (size_t)r * sa;
        const uint8_t *pb = b + (size_t)r * sb;
        uint32_t row = 0;
        for (int i = 0; i < 3 * ww; ++i) {
            const int d = (int)pa[i] - (int)pb[i];
            row += ssd ? (uint32_t)(d * d) : (uint32_t)(d < 0 ? -d : d);
        }
        acc += row;
    }
    return acc;
}

__device__ __forceinline__ bool black3(const uint8_t *p) { return (p[0] | p[1] | p[2]) == 0; }

// The pixels outside the skip rectangle, enumerated densely: rows above it, rows below it, then
// for the rows beside it the columns left and right of it.
__device__ __forceinline__ bool ring_pixel(const GenericArgs &g, int ow, int oh, long long idx, int *px, int *py)
{
    const long long n_top = (long long)g.skip_y0 * ow;
    const long long n_bot = (long long)(oh - g.skip_y1) * ow;
    const int side = g.skip_x0 + (ow - g.skip_x1);
    const long long n_side = (long long)(g.skip_y1 - g.skip_y0) * side;
    if (idx < n_top) {
        *py = (int)(idx / ow);
        *px = (int)(idx % ow);
    } else if (idx < n_top + n_bot) {
        idx -= n_top;
        *py = g.skip_y1 + (int)(idx / ow);
        *px = (int)(idx % ow);
    } else if (idx < n_top + n_bot + n_side) {
        idx -= n_top + n_bot;
        *py = g.skip_y0 + (int)(idx / side);
        const int k = (int)(idx % side);
        *px = k < g.skip_x0 ? k : g.skip_x1 + (k - g.skip_x0);
    } else {
        return false;
    }
    return true;
}

// One output pixel of the literal brute force on the original images; idx enumerates the pixels
// outside the skip rectangle (ring_pixel).
__device__ __forceinline__ void generic_pixel(const GenericArgs &g, long long idx)
{
    const int ow = g.view == 0 ? g.w1 : g.w2, oh = g.view == 0 ? g.h1 : g.h2;
    int x, y;
    if (!ring_pixel(g, ow, oh, idx, &x, &y)) return;
    const int height = min(g.h1, g.h2);
    float val = 0.0f;
    if (g.view == 0) { // BlockSearch.cpp:24-86
        const int half = (g.block_size - 1) / 2;
        if (y >= half && y < height - half && x >= half && x < g.w1 - half &&
            !black3(g.L + (size_t)y * g.s1 + 3 * x)) {
            const uint8_t *lw = g.L + (size_t)(y - half) * g.s1 + 3 * (x - half);
            uint32_t best = 0xffffffffu;
            int best_cx = 0;
            for (int cx = x - g.max_d; cx < x; ++cx) {
                if (cx < half || cx >= g.w2 - half) continue;
                const uint8_t *rw = g.R + (size_t)(y - half) * g.s2 + 3 * (cx - half);
                const uint32_t cst = window_cost(lw, g.s1, rw, g.s2, g.block_size, g.block_size, g.ssd);
                if (cst < best) {
                    best = cst;
                    best_cx = cx;
                }
            }
            val = (float)(x - best_cx);
        }
    } else if (g.view == 1) { // BlockSearch.cpp:88-179 (varBlock off)
        if (y < height && !black3(g.R + (size_t)y * g.s2 + 3 * x)) {
            const int half = (g.block_size - 1) / 2;
            const int left = min(x, half), right = min(g.w2 - x - 1, half);
            const int up = min(y, half), down = min(g.h2 - y - 1, half);
            const int ww = left + right, wh = up + down;
            uint32_t best = 0xffffffffu;
            int best_cx = 0;
            if (ww > 0 && wh > 0) { // empty window: 0/0 = NaN never wins (BlockSearch.cpp:158)
                const uint8_t *rw = g.R + (size_t)(y - up) * g.s2 + 3 * (x - left);
                for (int cx = x + g.min_d; cx < x + g.max_d; ++cx) {
                    if (cx + right >= g.w1) break;
                    const uint8_t *lw = g.L + (size_t)(y - up) * g.s1 + 3 * (cx - left);
                    const uint32_t cst = window_cost(lw, g.s1, rw, g.s2, ww, wh, g.ssd);
                    if (cst < best) {
                        best = cst;
                        best_cx = cx;
                    }
                }
            }
            val = (float)(best_cx - x);
        }
    } else { // LinearSearch.cpp:10-59
        if (y < g.h1 && !(x < g.w1 && black3(g.L + (size_t)y * g.s1 + 3 * x))) {
            const uint8_t *pr = g.R + (size_t)y * g.s2 + 3 * x;
            uint32_t best = 0xffffffffu;
            int col = 0;
            for (int k = x + g.min_d; k < x + g.linear_range; ++k) {
                if (k >= g.w1) break;
                const uint32_t cst = window_cost(pr, 0, g.L + (size_t)y * g.s1 + 3 * k, 0, 1, 1, 1);
                if (cst < best) {
                    best = cst;
                    col = k;
                }
            }
            val = (float)(col - x);
        }
    }
    if (g.out16) g.out16[(size_t)y * g.out_pitch + x] = (int16_t)(int)val;
    else g.out[(size_t)y * g.out_pitch + x] = val;
}

// wave-wide sum by a butterfly of shuffles
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

} // namespace wsamd
