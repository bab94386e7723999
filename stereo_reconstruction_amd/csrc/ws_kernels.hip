// ws_kernels.hip -- gfx950 (CDNA4) kernels of the WindowSearch hot path.
//
// What the reference computes (BlockSearch.cpp:24-179): for every pixel, for every candidate
// disparity, the L2 norm of the absolute difference of two bs x bs x 3 windows, and the
// candidate with the strictly smallest value.  It re-sums the window for every (pixel, d).
//
// What runs here instead (same integers, same winner):
//   ws_pack_kernel   BGR bytes -> one dword per pixel (B | G<<8 | R<<16), zero padded plane,
//                    mirrored in x for the right view.
//   ws_bias_kernel   per (row, B column): validity poison, and for SSD the box sum of the
//                    squared target pixels (the part of sum (a-b)^2 that does not need a).
//   ws_march_kernel  the hot kernel.  A workgroup owns a tile of X*nxr columns and a strip of
//                    rows; thread (r, c) owns X consecutive columns and ND consecutive
//                    disparities and keeps their X*ND window sums in registers while the
//                    workgroup marches down the strip one row at a time:
//                      - rows are staged once per step into an LDS ring and re-used by every
//                        disparity chunk of the tile,
//                      - per row and d a prefix chain of v_sad_u8 / v_dot4_u32_u8 (one
//                        instruction per pixel pair, 3 channels at once) gives all horizontal
//                        window sums by differences; the row leaving the window is removed the
//                        same way (sliding box filter, exact in integers),
//                      - the running minimum is a single signed v_min on (cost << k | tie tag),
//                      - the d-chunks of a pixel meet through one ds_min_u64 per thread and row.
//   ws_generic_kernel  literal per-pixel brute force: right-view border ring (clipped windows),
//                    LinearSearch, and window sizes without a marching instantiation.
//   ws_refine_kernel sub-pixel parabola (extension).
//
// No MFMA: the hot loop is a stencil + reduction on bytes, bounded by VALU issue and LDS, see
// DESIGN.md.  Wave64 throughout; nothing here assumes 32-wide warps.
#include "ws_kernels.h"

#include <limits.h>

namespace wsamd {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ uint32_t pix_sad(uint32_t a, uint32_t b, uint32_t acc)
{
    return __builtin_amdgcn_sad_u8(a, b, acc); // v_sad_u8: acc + sum |a.b[i] - b.b[i]|
}
__device__ __forceinline__ uint32_t pix_dot(uint32_t a, uint32_t b, uint32_t acc)
{
    return __builtin_amdgcn_udot4(a, b, acc, false); // v_dot4_u32_u8: acc + sum a.b[i] * b.b[i]
}

// ------------------------------------------------------------------------------------------
// pack: CV_8UC3 rows -> padded dword plane
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ws_pack_kernel(const uint8_t *__restrict__ src, int w, int h,
                                                      int stride, int mirror,
                                                      uint32_t *__restrict__ dst, int pitch, int pad)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (col >= pitch || y >= h) return;
    int x = col - pad;
    uint32_t v = 0;
    if (x >= 0 && x < w) {
        if (mirror) x = w - 1 - x;
        const uint8_t *p = src + (size_t)y * stride + (size_t)x * 3;
        v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
    dst[(size_t)y * pitch + col] = v;
}

hipError_t launch_pack(const uint8_t *src, int w, int h, int stride, int mirror, Plane dst,
                       hipStream_t s)
{
    dim3 grid(ceil_div(dst.pitch, 256), h);
    hipLaunchKernelGGL(ws_pack_kernel, grid, dim3(256), 0, s, src, w, h, stride, mirror, dst.data,
                       dst.pitch, dst.pad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// bias rows: poison for invalid B centres; for SSD the box-summed squares of B
// ------------------------------------------------------------------------------------------
struct BiasArgs {
    const uint32_t *B;
    int pitch, pad;
    int ww, wh, wx0, wy0;
    int b_lo, b_hi, oy0, oy1;
    int ssd, shift;
    int32_t *bias;
};

__global__ void __launch_bounds__(256) ws_bias_kernel(const BiasArgs g)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = g.oy0 + blockIdx.y;
    if (col >= g.pitch || y >= g.oy1) return;
    const int xb = col - g.pad;
    int32_t v = kPoison;
    if (xb >= g.b_lo && xb <= g.b_hi) {
        uint32_t acc = 0;
        if (g.ssd) {
            for (int wy = 0; wy < g.wh; ++wy) {
                const uint32_t *row = g.B + (size_t)(y + g.wy0 + wy) * g.pitch + (col + g.wx0);
                for (int wx = 0; wx < g.ww; ++wx) acc = pix_dot(row[wx], row[wx], acc);
            }
        }
        v = (int32_t)(acc << g.shift);
    }
    g.bias[(size_t)y * g.pitch + col] = v;
}

// ------------------------------------------------------------------------------------------
// the marching kernel
// ------------------------------------------------------------------------------------------
struct MarchArgs {
    const uint32_t *A;
    const uint32_t *B;
    const int32_t *bias;
    float *out;
    int pitch_a, pad_a, pitch_b, pad_b, out_pitch;
    int wa;
    int nxr, nch;
    int wx0, wy0, boff;
    int d_lo, d_hi;
    int ox0, ox1, oy0, oy1;
    int strip_rows;
    int prefer_large, mirror, fallback_neg;
};

// N consecutive dwords from a 16-byte aligned LDS address: ds_read_b128 for the quads.
template <int N>
__device__ __forceinline__ void lds_run(uint32_t (&dst)[N], const uint32_t *src)
{
    constexpr int Q = N / 4;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const uint4 v = reinterpret_cast<const uint4 *>(src)[q];
        dst[4 * q + 0] = v.x;
        dst[4 * q + 1] = v.y;
        dst[4 * q + 2] = v.z;
        dst[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int k = 4 * Q; k < N; ++k) dst[k] = src[k];
}

// One row entering (SIGN=+1) or leaving (SIGN=-1) the window of every (column, disparity) this
// thread owns.  V holds   SAD: (window sum << SH) + tag      SSD: tag - (2 * cross sum << LT).
// With KEY the candidate keys  bias[xb] + V  are folded into best[] (signed min; ties go to the
// smaller tag, i.e. to the disparity the reference's strict '<' would have kept).
template <int X, int ND, int WW, bool SSD, int SIGN, bool KEY>
__device__ __forceinline__ void march_row(int32_t (&V)[X][ND], int32_t (&best)[X],
                                          const uint32_t *rowA, const uint32_t *rowB,
                                          const int32_t *rowBias)
{
    constexpr int NA = X + WW - 1;
    constexpr int NB = NA + ND - 1;
    constexpr int NBI = X + ND - 1;
    constexpr int LT = ilog2c(ND);
    constexpr int SH = SSD ? LT + 1 : LT;
    // SAD accumulates +cost, SSD accumulates -2*cross: flip the sign of the update for SSD
    constexpr bool ADD = ((SIGN > 0) != SSD);

    uint32_t pa[NA], pb[NB];
    lds_run<NA>(pa, rowA);
    lds_run<NB>(pb, rowB);
    uint32_t bi[NBI];
    if (KEY) lds_run<NBI>(bi, reinterpret_cast<const uint32_t *>(rowBias));

#pragma unroll
    for (int j = 0; j < ND; ++j) {
        uint32_t S[NA];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const uint32_t b = pb[i - j + ND - 1];
            s = SSD ? pix_dot(pa[i], b, s) : pix_sad(pa[i], b, s);
            S[i] = s;
        }
#pragma unroll
        for (int x = 0; x < X; ++x) {
            const uint32_t hi = S[x + WW - 1];
            const uint32_t lo = x ? S[x - 1] : 0u;
            const uint32_t w = ADD ? hi - lo : lo - hi;
            V[x][j] = (int32_t)((w << SH) + (uint32_t)V[x][j]);
            if (KEY) {
                const int32_t key = (int32_t)bi[x - j + ND - 1] + V[x][j];
                best[x] = min(best[x], key);
            }
        }
    }
}

template <int X, int ND, int WW, int WH, bool SSD, int MAXT>
__global__ void __launch_bounds__(MAXT) ws_march_kernel(const MarchArgs g)
{
    constexpr int LT = ilog2c(ND);
    constexpr int NR = WH + 2; // ring rows: WH+1 in use by a step, 1 being filled for the next
    constexpr int PA = 2, PB = 3, PBI = 2; // prefetch registers per thread

    extern __shared__ uint4 ws_smem4[];
    uint32_t *smem = reinterpret_cast<uint32_t *>(ws_smem4);

    const int NT = blockDim.x, tid = threadIdx.x;
    const int tx = g.nxr * X, dt = g.nch * ND;
    const int n_a = tx + WW - 1, n_b = tx + WW + dt - 2, n_bi = tx + dt - 1;
    const int a_w = (n_a + 3) & ~3, b_w = (n_b + 3) & ~3, bi_w = (n_bi + 3) & ~3;
    uint32_t *ringA = smem;
    uint32_t *ringB = ringA + NR * a_w;
    int32_t *biasr = reinterpret_cast<int32_t *>(ringB + NR * b_w);
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(biasr + 2 * bi_w);

    const int tile_x0 = g.ox0 + blockIdx.x * tx;
    const int ys = g.oy0 + blockIdx.y * g.strip_rows;
    const int ye = min(ys + g.strip_rows, g.oy1);
    if (ys >= ye) return; // uniform per workgroup

    const int dhi_t = g.d_lo + dt - 1;
    const uint32_t *gA = g.A + (tile_x0 + g.wx0 + g.pad_a);
    const uint32_t *gB = g.B + (tile_x0 + g.wx0 + g.boff - dhi_t + g.pad_b);
    const int32_t *gBi = g.bias + (tile_x0 + g.boff - dhi_t + g.pad_b);

    for (int k = tid; k < 2 * tx; k += NT) slots[k] = ~0ull;

    const int r = tid % g.nxr, c = tid / g.nxr;
    const bool worker = c < g.nch;
    const int lb0 = r * X + (g.nch - 1 - (worker ? c : 0)) * ND;
    const int d0 = g.d_lo + c * ND; // first disparity of this thread's chunk

    int32_t V[X][ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int tag = g.prefer_large ? (ND - 1 - j) : j;
        const int32_t init = (d0 + j <= g.d_hi) ? tag : (kPoison + tag);
#pragma unroll
        for (int x = 0; x < X; ++x) V[x][j] = init;
    }

    const int ra0 = ys + g.wy0; // first window row of the first output row
    const int nsteps = (ye - ys) + WH - 1;

    // prologue: row ra0 (and the bias row of step 0 when the window is one row high)
    for (int k = tid; k < n_a; k += NT) ringA[k] = gA[(size_t)ra0 * g.pitch_a + k];
    for (int k = tid; k < n_b; k += NT) ringB[k] = gB[(size_t)ra0 * g.pitch_b + k];
    if (WH == 1)
        for (int k = tid; k < n_bi; k += NT) biasr[k] = gBi[(size_t)ys * g.pitch_b + k];
    __syncthreads();

    int add_slot = 0;       // ring slot of the row entering at this step   (a     mod NR)
    int sub_slot = 2 % NR;  // ring slot of the row leaving at this step    (a-WH  mod NR)
    for (int a = 0; a < nsteps; ++a) {
        const int oi = a - (WH - 1); // output row index inside the strip produced by this step

        // 1. hand the row finished in the previous step to HBM
        if (oi >= 1) {
            const int y = ys + oi - 1;
            unsigned long long *sl = slots + ((oi - 1) & 1) * tx;
            for (int k = tid; k < tx; k += NT) {
                const unsigned long long key = sl[k];
                sl[k] = ~0ull;
                const int x = tile_x0 + k;
                if (x < g.ox1) {
                    const int xo = g.mirror ? g.wa - 1 - x : x;
                    float val;
                    if (key == ~0ull) {
                        val = g.fallback_neg ? -(float)xo : (float)xo;
                    } else {
                        const int gtag = (int)(uint32_t)key;
                        val = (float)(g.prefer_large ? g.d_hi - gtag : g.d_lo + gtag);
                    }
                    if (g.A[(size_t)y * g.pitch_a + x + g.pad_a] == 0u) val = 0.0f; // black pixel
                    g.out[(size_t)y * g.out_pitch + xo] = val;
                }
            }
        }

        // 2. issue the loads of the next step now; they land in LDS after the arithmetic
        const bool more = a + 1 < nsteps;
        const bool more_bias = more && (oi + 1 >= 0);
        uint32_t fa[PA], fb[PB];
        int32_t fbi[PBI];
        if (more) {
            const uint32_t *srcA = gA + (size_t)(ra0 + a + 1) * g.pitch_a;
            const uint32_t *srcB = gB + (size_t)(ra0 + a + 1) * g.pitch_b;
#pragma unroll
            for (int q = 0; q < PA; ++q) {
                const int k = tid + q * NT;
                fa[q] = k < n_a ? srcA[k] : 0u;
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int k = tid + q * NT;
                fb[q] = k < n_b ? srcB[k] : 0u;
            }
        }
        if (more_bias) {
            const int32_t *srcBi = gBi + (size_t)(ys + oi + 1) * g.pitch_b;
#pragma unroll
            for (int q = 0; q < PBI; ++q) {
                const int k = tid + q * NT;
                fbi[q] = k < n_bi ? srcBi[k] : 0;
            }
        }

        // 3. arithmetic
        if (worker) {
            int32_t best[X];
#pragma unroll
            for (int x = 0; x < X; ++x) best[x] = INT_MAX;
            if (a >= WH)
                march_row<X, ND, WW, SSD, -1, false>(V, best, ringA + sub_slot * a_w + r * X,
                                                     ringB + sub_slot * b_w + lb0, nullptr);
            if (oi >= 0) {
                march_row<X, ND, WW, SSD, +1, true>(V, best, ringA + add_slot * a_w + r * X,
                                                    ringB + add_slot * b_w + lb0,
                                                    biasr + (oi & 1) * bi_w + lb0);
                unsigned long long *sl = slots + (oi & 1) * tx + r * X;
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    const int32_t bk = best[x];
                    const int32_t t = bk >> LT;
                    if (t < (kValidKeyBound >> LT)) {
                        const int jt = bk & (ND - 1);
                        const int d = d0 + (g.prefer_large ? ND - 1 - jt : jt);
                        const uint32_t gtag = (uint32_t)(g.prefer_large ? g.d_hi - d : d - g.d_lo);
                        const unsigned long long key =
                            ((unsigned long long)((uint32_t)t ^ 0x80000000u) << 32) | gtag;
                        atomicMin(sl + x, key); // ds_min_u64
                    }
                }
            } else {
                march_row<X, ND, WW, SSD, +1, false>(V, best, ringA + add_slot * a_w + r * X,
                                                     ringB + add_slot * b_w + lb0, nullptr);
            }
        }

        // 4. land the prefetched row in the ring slot nobody reads during this step
        int nxt_slot = add_slot + 1;
        if (nxt_slot == NR) nxt_slot = 0;
        if (more) {
            uint32_t *dstA = ringA + nxt_slot * a_w;
            uint32_t *dstB = ringB + nxt_slot * b_w;
#pragma unroll
            for (int q = 0; q < PA; ++q) {
                const int k = tid + q * NT;
                if (k < n_a) dstA[k] = fa[q];
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int k = tid + q * NT;
                if (k < n_b) dstB[k] = fb[q];
            }
            // tiles too wide for the prefetch registers: finish synchronously
            const uint32_t *srcA = gA + (size_t)(ra0 + a + 1) * g.pitch_a;
            const uint32_t *srcB = gB + (size_t)(ra0 + a + 1) * g.pitch_b;
            for (int k = tid + PA * NT; k < n_a; k += NT) dstA[k] = srcA[k];
            for (int k = tid + PB * NT; k < n_b; k += NT) dstB[k] = srcB[k];
        }
        if (more_bias) {
            int32_t *dstBi = biasr + ((oi + 1) & 1) * bi_w;
#pragma unroll
            for (int q = 0; q < PBI; ++q) {
                const int k = tid + q * NT;
                if (k < n_bi) dstBi[k] = fbi[q];
            }
            const int32_t *srcBi = gBi + (size_t)(ys + oi + 1) * g.pitch_b;
            for (int k = tid + PBI * NT; k < n_bi; k += NT) dstBi[k] = srcBi[k];
        }
        __syncthreads();
        add_slot = nxt_slot;
        if (++sub_slot == NR) sub_slot = 0;
    }

    // last row of the strip
    {
        const int oi = nsteps - 1 - (WH - 1);
        const int y = ys + oi;
        unsigned long long *sl = slots + (oi & 1) * tx;
        for (int k = tid; k < tx; k += NT) {
            const unsigned long long key = sl[k];
            const int x = tile_x0 + k;
            if (x < g.ox1) {
                const int xo = g.mirror ? g.wa - 1 - x : x;
                float val;
                if (key == ~0ull) {
                    val = g.fallback_neg ? -(float)xo : (float)xo;
                } else {
                    const int gtag = (int)(uint32_t)key;
                    val = (float)(g.prefer_large ? g.d_hi - gtag : g.d_lo + gtag);
                }
                if (g.A[(size_t)y * g.pitch_a + x + g.pad_a] == 0u) val = 0.0f;
                g.out[(size_t)y * g.out_pitch + xo] = val;
            }
        }
    }
}

// ---- instantiation table -----------------------------------------------------------------
constexpr int kX = 8, kND = 8, kMaxT = 768;
constexpr int kMinXRuns = 4; // narrowest tile: 4 x-runs = 32 columns (D up to 1536)

typedef void (*MarchFn)(const MarchArgs);
struct MarchEntry {
    int ww, wh, ssd;
    MarchFn fn;
    const char *name;
};
#define WS_MARCH_ENTRY(W, H)                                                                     \
    {W, H, 0, ws_march_kernel<kX, kND, W, H, false, kMaxT>, "ws_march_kernel<sad," #W "x" #H ">"}, \
    {W, H, 1, ws_march_kernel<kX, kND, W, H, true, kMaxT>, "ws_march_kernel<ssd," #W "x" #H ">"}
static const MarchEntry kMarchTable[] = {
    WS_MARCH_ENTRY(3, 3), WS_MARCH_ENTRY(5, 5), WS_MARCH_ENTRY(7, 7), WS_MARCH_ENTRY(9, 9),
    WS_MARCH_ENTRY(2, 2), WS_MARCH_ENTRY(4, 4), WS_MARCH_ENTRY(6, 6), WS_MARCH_ENTRY(8, 8),
};

static const MarchEntry *find_march(const Canon &c)
{
    for (const MarchEntry &e : kMarchTable)
        if (e.ww == c.ww && e.wh == c.wh && e.ssd == c.ssd) return &e;
    return nullptr;
}

bool march_supported(const Canon &c)
{
    if (!find_march(c)) return false;
    if (c.ox1 <= c.ox0 || c.oy1 <= c.oy0) return false;
    const int dcount = c.d_hi - c.d_lo + 1;
    if (dcount < 1) return false;
    if (ceil_div(dcount, kND) > kMaxT / kMinXRuns) return false;
    // keys must stay inside (-2^28, 2^28): 2 * window * 3 * 255^2 << log2(ND)
    const long long worst = 2LL * c.ww * c.wh * 3 * 255 * 255 * kND;
    return worst < (long long)kValidKeyBound;
}

bool march_plan(const Canon &c, int num_cus, int tune_nxr, int tune_strip_rows, int tune_threads,
                MarchLaunch *out)
{
    if (!march_supported(c)) return false;
    MarchLaunch m{};
    m.x_per_thread = kX;
    m.nd_per_thread = kND;
    m.max_threads = kMaxT;
    const int dcount = c.d_hi - c.d_lo + 1;
    const int out_w = c.ox1 - c.ox0, out_h = c.oy1 - c.oy0;
    m.nch = ceil_div(dcount, kND);
    if (m.nch < 8) m.nch = 8;
    int maxt = kMaxT;
    if (tune_threads >= 64 && tune_threads < kMaxT) maxt = tune_threads / 64 * 64;
    int nxr = maxt / m.nch;
    if (tune_nxr > 0 && tune_nxr < nxr) nxr = tune_nxr;
    const int need = ceil_div(out_w, kX); // no point in tiles wider than the image
    if (nxr > need) nxr = need;
    if (nxr < kMinXRuns) nxr = kMinXRuns;
    if (nxr * m.nch > kMaxT) return false;
    m.nxr = nxr;
    m.threads = round_up(nxr * m.nch, 64);
    const int tx = nxr * kX;
    m.tiles = ceil_div(out_w, tx);
    int strips;
    if (tune_strip_rows > 0) {
        strips = ceil_div(out_h, tune_strip_rows);
    } else {
        // one workgroup per CU; fill the chip once if the strips stay reasonably tall,
        // otherwise aim at ~64-row strips in whole multiples of the CU count
        strips = num_cus / m.tiles;
        if (strips < 1) strips = 1;
        if (ceil_div(out_h, strips) > 96) {
            const int rounds = ceil_div(ceil_div(out_h, 64) * m.tiles, num_cus);
            strips = rounds * num_cus / m.tiles;
            if (strips < 1) strips = 1;
        }
        if (strips > out_h) strips = out_h;
    }
    m.strip_rows = ceil_div(out_h, strips);
    m.strips = ceil_div(out_h, m.strip_rows);
    const int dt = m.nch * kND;
    const int a_w = round_up(tx + c.ww - 1, 4), b_w = round_up(tx + c.ww + dt - 2, 4),
              bi_w = round_up(tx + dt - 1, 4);
    const int nr = c.wh + 2;
    m.lds_bytes = (size_t)(nr * a_w + nr * b_w + 2 * bi_w) * 4 + (size_t)2 * tx * 8;
    if (m.lds_bytes > 160 * 1024) return false;
    *out = m;
    return true;
}

void march_plane_geometry(const Canon &c, const MarchLaunch &m, int *pad_a, int *pitch_a,
                          int *pad_b, int *pitch_b)
{
    const int tx = m.nxr * m.x_per_thread, dt = m.nch * m.nd_per_thread;
    const int dhi_t = c.d_lo + dt - 1;
    // A columns touched: [ox0 + wx0, ox0 + tiles*tx + ww - 1 + wx0)
    int lo_a = c.ox0 + c.wx0, hi_a = c.ox0 + m.tiles * tx + c.ww - 1 + c.wx0;
    if (lo_a > 0) lo_a = 0;
    if (hi_a < c.wa) hi_a = c.wa;
    *pad_a = round_up(-lo_a, 4);
    *pitch_a = round_up(hi_a + *pad_a, 64);
    // B columns touched: from ox0 + wx0 + boff - dhi_t (also the bias rows, without wx0)
    int lo_b = c.ox0 + c.wx0 + c.boff - dhi_t;
    int hi_b = c.ox0 + (m.tiles - 1) * tx + c.boff - dhi_t + (tx + c.ww + dt - 2) + (c.wx0 > 0 ? c.wx0 : 0);
    if (lo_b > 0) lo_b = 0;
    if (hi_b < c.wb) hi_b = c.wb;
    *pad_b = round_up(-lo_b, 4);
    *pitch_b = round_up(hi_b + *pad_b + 4, 64);
}

hipError_t launch_bias(const Canon &c, const MarchLaunch &m, Plane b, int32_t *bias, hipStream_t s)
{
    BiasArgs g{};
    g.B = b.data;
    g.pitch = b.pitch;
    g.pad = b.pad;
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
    g.bias = bias;
    dim3 grid(ceil_div(b.pitch, 256), c.oy1 - c.oy0);
    hipLaunchKernelGGL(ws_bias_kernel, grid, dim3(256), 0, s, g);
    return hipGetLastError();
}

const char *march_kernel_name(const Canon &c, const MarchLaunch &)
{
    const MarchEntry *e = find_march(c);
    return e ? e->name : "";
}

hipError_t launch_march(const Canon &c, const MarchLaunch &m, Plane a, Plane b,
                        const int32_t *bias, float *out, int out_pitch, hipStream_t s)
{
    const MarchEntry *e = find_march(c);
    if (!e) return hipErrorInvalidValue;
    MarchArgs g{};
    g.A = a.data;
    g.B = b.data;
    g.bias = bias;
    g.out = out;
    g.pitch_a = a.pitch;
    g.pad_a = a.pad;
    g.pitch_b = b.pitch;
    g.pad_b = b.pad;
    g.out_pitch = out_pitch;
    g.wa = c.wa;
    g.nxr = m.nxr;
    g.nch = m.nch;
    g.wx0 = c.wx0;
    g.wy0 = c.wy0;
    g.boff = c.boff;
    g.d_lo = c.d_lo;
    g.d_hi = c.d_hi;
    g.ox0 = c.ox0;
    g.ox1 = c.ox1;
    g.oy0 = c.oy0;
    g.oy1 = c.oy1;
    g.strip_rows = m.strip_rows;
    g.prefer_large = c.prefer_large;
    g.mirror = c.mirror;
    g.fallback_neg = c.fallback_neg;
    if (m.lds_bytes > 48 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(e->fn),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)m.lds_bytes);
        if (err != hipSuccess) return err;
    }
    dim3 grid(m.tiles, m.strips);
    hipLaunchKernelGGL(e->fn, grid, dim3(m.threads), m.lds_bytes, s, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// literal brute force on the original images
// ------------------------------------------------------------------------------------------
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

__device__ __forceinline__ bool black3(const uint8_t *p) { return (p[0] | p[1] | p[2]) == 0; }

__global__ void __launch_bounds__(256) ws_generic_kernel(const GenericArgs g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int ow = g.view == 0 ? g.w1 : g.w2, oh = g.view == 0 ? g.h1 : g.h2;
    if (x >= ow || y >= oh) return;
    if (x >= g.skip_x0 && x < g.skip_x1 && y >= g.skip_y0 && y < g.skip_y1) return;
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
            for (int k = x; k < x + g.linear_range; ++k) {
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
    g.out[(size_t)y * g.out_pitch + x] = val;
}

hipError_t launch_generic(const GenericArgs &g, hipStream_t s)
{
    const int ow = g.view == 0 ? g.w1 : g.w2, oh = g.view == 0 ? g.h1 : g.h2;
    dim3 grid(ceil_div(ow, 256), oh);
    hipLaunchKernelGGL(ws_generic_kernel, grid, dim3(256), 0, s, g);
    return hipGetLastError();
}

// Sub-pixel refinement (build extension): the integer map is already final; a
// pixel is refined when d-1, d and d+1 are all candidates the search itself would have tried.
__global__ void __launch_bounds__(256) ws_refine_kernel(const GenericArgs g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int height = min(g.h1, g.h2);
    const int half = (g.block_size - 1) / 2;
    uint32_t cm, c0, cp;
    float *o;
    if (g.view == 0) {
        if (x < half || x >= g.w1 - half || y < half || y >= height - half) return;
        if (black3(g.L + (size_t)y * g.s1 + 3 * x)) return;
        o = g.out + (size_t)y * g.out_pitch + x;
        const int d = (int)*o;
        const int cx = x - d;
        // was there any valid candidate, and are both neighbours valid ones?
        if (d < 1 || d > g.max_d || cx < half || cx >= g.w2 - half) return;
        const int cxm = cx + 1, cxp = cx - 1;
        if (!(cxm < x && cxm < g.w2 - half)) return;
        if (!(cxp >= x - g.max_d && cxp >= half)) return;
        const uint8_t *lw = g.L + (size_t)(y - half) * g.s1 + 3 * (x - half);
        const uint8_t *rw = g.R + (size_t)(y - half) * g.s2 + 3 * (cx - half);
        c0 = window_cost(lw, g.s1, rw, g.s2, g.block_size, g.block_size, g.ssd);
        cm = window_cost(lw, g.s1, rw + 3, g.s2, g.block_size, g.block_size, g.ssd);
        cp = window_cost(lw, g.s1, rw - 3, g.s2, g.block_size, g.block_size, g.ssd);
    } else {
        if (x >= g.w2 || y >= height) return;
        if (black3(g.R + (size_t)y * g.s2 + 3 * x)) return;
        const int left = min(x, half), right = min(g.w2 - x - 1, half);
        const int up = min(y, half), down = min(g.h2 - y - 1, half);
        const int ww = left + right, wh = up + down;
        if (ww <= 0 || wh <= 0) return;
        o = g.out + (size_t)y * g.out_pitch + x;
        const int d = (int)*o;
        const int cx = x + d;
        if (d < g.min_d || d >= g.max_d || cx + right >= g.w1) return; // fallback value, not a match
        if (!(cx - 1 >= x + g.min_d && cx - 1 - left >= 0)) return;
        if (!(cx + 1 < x + g.max_d && cx + 1 + right < g.w1)) return;
        const uint8_t *rw = g.R + (size_t)(y - up) * g.s2 + 3 * (x - left);
        const uint8_t *lw = g.L + (size_t)(y - up) * g.s1 + 3 * (cx - left);
        c0 = window_cost(lw, g.s1, rw, g.s2, ww, wh, g.ssd);
        cm = window_cost(lw - 3, g.s1, rw, g.s2, ww, wh, g.ssd);
        cp = window_cost(lw + 3, g.s1, rw, g.s2, ww, wh, g.ssd);
    }
    // exact integer numerator / denominator, one float division
    const long long num = (long long)cm - (long long)cp;
    const long long den = (long long)cm - 2LL * (long long)c0 + (long long)cp;
    if (den > 0) *o = *o + (float)((double)num / (2.0 * (double)den));
}

hipError_t launch_refine(const GenericArgs &g, hipStream_t s)
{
    const int ow = g.view == 0 ? g.w1 : g.w2, oh = g.view == 0 ? g.h1 : g.h2;
    dim3 grid(ceil_div(ow, 256), oh);
    hipLaunchKernelGGL(ws_refine_kernel, grid, dim3(256), 0, s, g);
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
