// ws_border.hip -- literal brute force (generic), right-view border ring, sub-pixel refine, varBlock
// Part of the gfx950 kernels of the WindowSearch hot path; overview in ws_march.hip.
#include "ws_device.h"

namespace wsamd {

// ------------------------------------------------------------------------------------------
// literal brute force on the original images
// ------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) ws_generic_kernel(const GenericArgs g)
{
    generic_pixel(g, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// ------------------------------------------------------------------------------------------
// right-view border ring: clipped windows (BlockSearch.cpp:116-123), sliding sums along runs
//
// The marching kernel owns the pixels whose (bs-1)^2 window is complete.  On the ring the window
// is clipped by the image border, so its size changes from pixel to pixel; there are only
// ~2*half*(W+H) such pixels.  They are taken in short runs along the border (rows of the top /
// bottom band, columns of the side bands), lanes over d, see ws_ring_kernel below.
// ------------------------------------------------------------------------------------------
struct RingArgs {
    const uint32_t *A;
    const uint32_t *B;
    int pitch_a, pad_a, pitch_b, pad_b;
    int wa, ha, wb;          // canonical plane sizes: A = mirrored right image, B = mirrored left
    int height;              // min(h1, h2)
    int half, boff, d_lo, d_hi;
    int ssd, centred;
    int skip_x0, skip_x1, skip_y0, skip_y1; // marching interior, ORIGINAL coordinates
    float *out;
    int16_t *out16; // if set: 16-bit integers here instead of floats to `out`
    int out_pitch;
    int32_t *cost_out; // optional: the winner's cost (SSD: without the sum of a^2), for the smoothFactor passes
    int cost_pitch;
};

constexpr int kRingTile = 8;                          // a workgroup owns up to 8 x 8 ring pixels
constexpr int kRingBatch = 8;                         // rows of plane B in flight per thread while a tile's strip is staged
constexpr int kRingMaxHalf = 8;                       // right-view marching windows: (2 half)^2 <= 16 x 16
constexpr int kRingCols = kRingTile + 2 * kRingMaxHalf; // window columns under a tile (<= tile + 2 half - 1)
constexpr int kRingSlack = 32;                         // words behind the staged strips that a row's unused columns may read

// One pixel pair's term with nothing to add it to (SSD: a dot product; the whole cost of a pair is b.b - 2 a.b, a^2
// being the same for every d).  Spelled out: for a zero accumulator the compiler takes the in-place v_dot4c form and
// clears its destination with a v_mov first.
template <int MODE> // 0 SAD, 1 SSD, 2 SSD on centred planes
__device__ __forceinline__ uint32_t ring_dot0(uint32_t a, uint32_t b)
{
    uint32_t r;
    if constexpr (MODE == 0) asm("v_sad_u8 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    else if constexpr (MODE == 1) asm("v_dot4_u32_u8 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// A window row's pixels of both planes under the tile: every read in flight before the first one is used (a tile is
// latency-, not throughput-bound; left to itself the compiler waits for each pair of columns in turn)
template <int NCOLS>
__device__ __forceinline__ void ring_row(uint32_t (&av)[NCOLS], uint32_t (&bv)[NCOLS], const uint32_t *ar, const uint32_t *br)
{
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
        av[c] = ar[c];
        bv[c] = br[c];
    }
    // (the reads stay in front of this point, the arithmetic on what they return behind it)
    static_assert(NCOLS % 8 == 0, "pinned eight at a time");
    asm volatile("" : : : "memory");
#pragma unroll
    for (int c = 0; c < NCOLS; c += 8) {
        asm volatile("" : "+v"(av[c]), "+v"(av[c + 1]), "+v"(av[c + 2]), "+v"(av[c + 3]), "+v"(av[c + 4]), "+v"(av[c + 5]),
                          "+v"(av[c + 6]), "+v"(av[c + 7]));
        asm volatile("" : "+v"(bv[c]), "+v"(bv[c + 1]), "+v"(bv[c + 2]), "+v"(bv[c + 3]), "+v"(bv[c + 4]), "+v"(bv[c + 5]),
                          "+v"(bv[c + 6]), "+v"(bv[c + 7]));
    }
}

// The clipped window of ring pixel (x, y), original coordinates -> canonical columns [ca, ce), rows [ra, re)
struct RingWin { int ca, ce, ra, re, d_end; };
__device__ __forceinline__ RingWin ring_window(const RingArgs &g, int x, int y)
{
    const int left = min(x, g.half), right = min(g.wa - x - 1, g.half);
    const int up = min(y, g.half), down = min(g.ha - y - 1, g.half);
    const int xm = g.wa - 1 - x;
    RingWin w;
    w.ca = xm - right + 1; w.ce = w.ca + left + right;
    w.ra = y - up; w.re = y + down;
    w.d_end = min(g.d_hi, g.wb - right - x - 1); // x + d + right < w1 (BlockSearch.cpp:147-149)
    return w;
}

// A small marching kernel for the ring.  A workgroup owns a tile of up to 8 x 8 ring pixels; its
// 256 threads are 256 consecutive disparities (per round).  Every thread keeps the COLUMN sums of
// its disparity's costs over the current row's window rows, for the <= 23 window columns under the
// tile (the plane strips under the tile are staged in LDS).  Going down a row, a column sum gains the row that enters the (clipped) window and
// loses the one that leaves -- at the top of the image rows only enter, at the bottom they only
// leave, beside it both -- and a pixel's window cost is the sum of its (clipped) column range, a
// difference of two prefix sums.  So a hypothesis costs about three pixel
// operations instead of a whole window.  Per row the costs go to LDS ([d][pixel], padded), lane
// (pixel, part) scans 8 disparities in ascending order -- the reference's tie rule -- and
// everything meets in one 64-bit LDS min per pixel.
template <int MODE, int NCOLS> // NCOLS >= tile + 2 half - 1 window columns under a tile: 16 (half <= 4) or 24
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) ws_ring_kernel(const RingArgs g, int xtiles, int seg_top, int seg_bot, int xt_left,
                                                      int xt_right, int seg_mid)
{
    extern __shared__ uint32_t ring_lds[]; // the tile's strips of plane A and plane B
    __shared__ int32_t prefix[NCOLS + 1][256];
    __shared__ int32_t costs[4][64][kRingTile + 1];
    __shared__ long long best[kRingTile * kRingTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // which tile: the bands above and below the marching interior over the whole width, then the
    // columns left and right of it
    int b = blockIdx.x, x0, y0, nx, ny;
    const int n_band = (seg_top + seg_bot) * xtiles;
    if (b < n_band) {
        const int seg = b / xtiles;
        x0 = (b - seg * xtiles) * kRingTile;
        nx = min(kRingTile, g.wa - x0);
        if (seg < seg_top) { y0 = seg * kRingTile; ny = min(kRingTile, g.skip_y0 - y0); }
        else { y0 = g.skip_y1 + (seg - seg_top) * kRingTile; ny = min(kRingTile, g.ha - y0); }
    } else {
        b -= n_band;
        const int xt = xt_left + xt_right, seg = b / xt, t = b - seg * xt;
        if (t < xt_left) { x0 = t * kRingTile; nx = min(kRingTile, g.skip_x0 - x0); }
        else { x0 = g.skip_x1 + (t - xt_left) * kRingTile; nx = min(kRingTile, g.wa - x0); }
        y0 = g.skip_y0 + seg * kRingTile;
        ny = min(kRingTile, g.skip_y1 - y0);
    }
    // Rows y >= min(h1, h2) are never searched (BlockSearch.cpp:94,100): they stay 0, and their windows
    // must not be staged either -- with h2 > h1 plane B (the left image) simply has no such rows.
    const int nyc = min(ny, g.height - y0); // rows of this tile that are searched (uniform per workgroup)
    // canonical window columns under the tile (they fall as x rises) and window rows (they rise with y)
    const RingWin wfirst = ring_window(g, x0, y0), wlast = ring_window(g, x0 + nx - 1, y0);
    const int CA = min(wfirst.ca, wlast.ca), CE = max(wfirst.ce, wlast.ce);
    const int nc = min(max(CE - CA, 0), NCOLS);
    const int RA = wfirst.ra, RE = nyc > 0 ? max(ring_window(g, x0, y0 + nyc - 1).re, RA) : RA;
    const int nr = min(RE - RA, NCOLS);
    // the strips of both planes under the tile's windows go through LDS: A once, B per round of 256
    // disparities (columns c - d + boff for c in [CA, CE), d in [dblk, dtop])
    uint32_t *sA = ring_lds, *sB = ring_lds + nc * nr;
    // (A strip's loads are all in flight before its first store, addresses clamped instead of branched around: a
    // workgroup is alone on its SIMDs, and a load-store-load chain costs a trip to memory per element.  No divisions:
    // A's rows are at most 32 columns -- a thread is (row tid / 32 + 8 k, column tid % 32) -- and B's at most
    // 256 + 31: column tid of eight rows at a time, then the columns past 256 like A's.)
    constexpr int kPasses = (NCOLS + 7) / 8;
    const int c32 = tid & 31, r8 = tid >> 5;
    if (nc * nr > 0) {
        uint32_t v[kPasses];
#pragma unroll
        for (int k = 0; k < kPasses; ++k) {
            const bool ok = c32 < nc && r8 + 8 * k < nr;
            v[k] = g.A[(size_t)(RA + (ok ? r8 + 8 * k : 0)) * g.pitch_a + (CA + (ok ? c32 : 0) + g.pad_a)]; // (the windows lie inside the image)
        }
#pragma unroll
        for (int k = 0; k < kPasses; ++k)
            if (c32 < nc && r8 + 8 * k < nr) sA[(r8 + 8 * k) * nc + c32] = v[k];
    }
    if (tid < kRingTile * kRingTile) best[tid] = LLONG_MAX;
    for (int dblk = g.d_lo; dblk <= g.d_hi && nyc > 0; dblk += 256) {
        const int dtop = min(dblk + 255, g.d_hi);
        const int CB = CA - dtop + g.boff, WB = nc + (dtop - dblk);
        __syncthreads(); // (the previous round's readers are done with sB; first round: sA / best are written)
        for (int rb = 0; rb < nr; rb += kRingBatch) {
            uint32_t v[kRingBatch];
            const int c = CB + tid + g.pad_b;
            const bool in = tid < WB && c >= 0 && c < g.pitch_b;
#pragma unroll
            for (int k = 0; k < kRingBatch; ++k) {
                const uint32_t px = g.B[(size_t)(RA + min(rb + k, nr - 1)) * g.pitch_b + (in ? c : 0)];
                v[k] = in ? px : 0u;
            }
#pragma unroll
            for (int k = 0; k < kRingBatch; ++k)
                if (rb + k < nr && tid < WB) sB[(rb + k) * WB + tid] = v[k];
        }
        if (WB > 256 && nr > 0) {
            uint32_t v[kPasses];
            const int cc = 256 + c32, c = CB + cc + g.pad_b;
            const bool in = cc < WB && c >= 0 && c < g.pitch_b;
#pragma unroll
            for (int k = 0; k < kPasses; ++k) {
                const uint32_t px = g.B[(size_t)(RA + min(r8 + 8 * k, nr - 1)) * g.pitch_b + (in ? c : 0)];
                v[k] = in ? px : 0u;
            }
#pragma unroll
            for (int k = 0; k < kPasses; ++k)
                if (cc < WB && r8 + 8 * k < nr) sB[(r8 + 8 * k) * WB + cc] = v[k];
        }
        __syncthreads();
        const int d = dblk + tid;
        // A0[r * nc + c] / B0[r * WB + c]: the planes at window row RA + r, window column CA + c (B shifted by d);
        // a thread beyond d_hi reads in-bounds garbage (clamped shift) that no candidate test accepts
        const uint32_t *A0 = sA, *B0 = sB + max(dtop - d, 0);
        // the column sums stay in registers; everything below is unrolled over the NCOLS columns at compile-time
        // offsets from the row's start (one address per row and plane, the columns in the reads' offset fields).
        // Columns past nc read whatever follows in LDS -- the next row, the other plane, the kRingSlack words
        // behind the strips -- into sums that no pixel's column range reaches (ce <= nc).
        uint32_t sum[NCOLS];
        int ra = 0, re = 0;
        for (int j = 0; j < nyc; ++j) {
            const int y = y0 + j;
            const RingWin wr = ring_window(g, x0, y); // (the window rows do not depend on x)
            const int r0 = wr.ra - RA, r1 = max(wr.re - RA, r0);
            // 1. column sums of the window rows [r0, r1)
            if (j == 0) {
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) sum[c] = 0;
                    for (int r = r0; r < r1; ++r) {
                        uint32_t av[NCOLS], bv[NCOLS];
                        ring_row<NCOLS>(av, bv, A0 + r * nc, B0 + r * WB);
#pragma unroll
                        for (int c = 0; c < NCOLS; ++c) sum[c] = pix_sad(av[c], bv[c], sum[c]);
                    }
                } else { // b.b and a.b on accumulators of their own: two instructions per pixel pair
                    uint32_t ab[NCOLS];
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) sum[c] = ab[c] = 0;
                    for (int r = r0; r < r1; ++r) {
                        uint32_t av[NCOLS], bv[NCOLS];
                        ring_row<NCOLS>(av, bv, A0 + r * nc, B0 + r * WB);
#pragma unroll
                        for (int c = 0; c < NCOLS; ++c) {
                            sum[c] = pix_dot<MODE == 2>(bv[c], bv[c], sum[c]);
                            ab[c] = pix_dot<MODE == 2>(av[c], bv[c], ab[c]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) sum[c] -= 2 * ab[c];
                }
            } else {
                for (int r = ra; r < r0; ++r) { // rows that left the window (at most one)
                    uint32_t av[NCOLS], bv[NCOLS];
                    ring_row<NCOLS>(av, bv, A0 + r * nc, B0 + r * WB);
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) {
                        if constexpr (MODE == 0) sum[c] -= ring_dot0<MODE>(av[c], bv[c]);
                        else sum[c] = sum[c] - ring_dot0<MODE>(bv[c], bv[c]) + 2 * ring_dot0<MODE>(av[c], bv[c]);
                    }
                }
                for (int r = re; r < r1; ++r) { // rows that entered it (at most one)
                    uint32_t av[NCOLS], bv[NCOLS];
                    ring_row<NCOLS>(av, bv, A0 + r * nc, B0 + r * WB);
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) {
                        if constexpr (MODE == 0) sum[c] = pix_sad(av[c], bv[c], sum[c]);
                        else sum[c] = pix_dot<MODE == 2>(bv[c], bv[c], sum[c]) - 2 * ring_dot0<MODE>(av[c], bv[c]);
                    }
                }
            }
            ra = r0; re = r1;
            // prefix sums over the columns, to LDS for the pixels' (run-time) column ranges
            {
                uint32_t acc = 0;
                prefix[0][tid] = 0;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    acc += sum[c];
                    prefix[c + 1][tid] = acc;
                }
            }
            // 2. the row's pixels: window cost = prefix[ce] - prefix[ca]
#pragma unroll
            for (int i = 0; i < kRingTile; ++i) {
                const RingWin w = ring_window(g, x0 + min(i, nx - 1), y);
                const bool empty = w.ce <= w.ca || w.re <= w.ra; // 0/0 = NaN never wins (BlockSearch.cpp:158)
                const int ca = min(max(w.ca - CA, 0), nc), ce = min(max(w.ce - CA, 0), nc);
                const int32_t wsum = prefix[ce][tid] - prefix[ca][tid];
                const bool valid = !empty && d <= w.d_end;
                costs[wave][lane][i] = valid ? wsum : INT_MAX; // (a real cost stays far below INT_MAX)
            }
            // 3. this wave's 64 disparities of pixel p: lanes (p, part) scan 8 each, ascending d, strict '<'
            const int p = lane & 7, part = lane >> 3;
            if (p < nx) {
                int32_t bc = INT_MAX;
                int bd = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int32_t c = costs[wave][part * 8 + k][p];
                    if (c < bc) { bc = c; bd = dblk + 64 * wave + part * 8 + k; }
                }
                if (bc != INT_MAX) atomicMin(&best[j * kRingTile + p], ((long long)bc << 32) | (uint32_t)bd); // ties: smaller d
            }
        }
    }
    __syncthreads();
    if (tid < kRingTile * kRingTile) {
        const int i = tid & 7, j = tid >> 3;
        if (i < nx && j < ny) {
            const int x = x0 + i, y = y0 + j;
            const uint32_t black = MODE == 2 ? kCentre : 0u;
            const long long k = best[j * kRingTile + i];
            float val = 0.0f;
            if (y < g.height && g.A[(size_t)y * g.pitch_a + (g.wa - 1 - x) + g.pad_a] != black)
                val = k == LLONG_MAX ? -(float)x : (float)(uint32_t)(k & 0xffffffffll);
            if (g.out16) g.out16[(size_t)y * g.out_pitch + x] = (int16_t)(int)val;
            else g.out[(size_t)y * g.out_pitch + x] = val;
            if (g.cost_out && k != LLONG_MAX) g.cost_out[(size_t)y * g.cost_pitch + x] = (int32_t)(k >> 32);
        }
    }
}

hipError_t launch_ring(const Canon &c, Plane a, Plane b, const GenericArgs &skip, float *out, int out_pitch,
                       int32_t *cost_out, int cost_pitch, hipStream_t s)
{
    RingArgs g{};
    g.A = a.data; g.B = b.data;
    g.pitch_a = a.pitch; g.pad_a = a.pad; g.pitch_b = b.pitch; g.pad_b = b.pad;
    g.wa = c.wa; g.ha = c.ha; g.wb = c.wb;
    g.height = std::min(c.ha, c.hb);
    g.half = c.wh / 2; // right view: window (bs-1)^2 = (2*half)^2
    g.boff = c.boff; g.d_lo = c.d_lo; g.d_hi = c.d_hi_clipped;
    g.ssd = c.ssd; g.centred = march_centred(c);
    g.skip_x0 = skip.skip_x0; g.skip_x1 = skip.skip_x1; g.skip_y0 = skip.skip_y0; g.skip_y1 = skip.skip_y1;
    if (g.skip_x1 <= g.skip_x0 || g.skip_y1 <= g.skip_y0) { // no interior: every row is a "top" row
        g.skip_x0 = g.skip_x1 = 0;
        g.skip_y0 = g.skip_y1 = c.ha;
    }
    g.out = out; g.out16 = skip.out16; g.out_pitch = out_pitch;
    g.cost_out = cost_out; g.cost_pitch = cost_pitch;
    if (g.half > kRingMaxHalf) return hipErrorInvalidValue;
    const int xtiles = ceil_div(c.wa, kRingTile);
    const int seg_top = ceil_div(g.skip_y0, kRingTile), seg_bot = ceil_div(c.ha - g.skip_y1, kRingTile);
    const int xt_left = ceil_div(g.skip_x0, kRingTile), xt_right = ceil_div(c.wa - g.skip_x1, kRingTile);
    const int seg_mid = ceil_div(g.skip_y1 - g.skip_y0, kRingTile);
    const long long blocks = (long long)(seg_top + seg_bot) * xtiles + (long long)(xt_left + xt_right) * seg_mid;
    if (blocks <= 0) return hipSuccess;
    dim3 grid((unsigned)blocks);
    const int side = kRingTile + 2 * g.half; // >= window columns and window rows under a tile
    const size_t lds = (size_t)(side * side + side * (side + 255) + kRingSlack) * sizeof(uint32_t);
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, g, xtiles, seg_top, seg_bot, xt_left, xt_right, seg_mid); };
    const bool narrow = kRingTile + 2 * g.half - 1 <= 16;
    if (!g.ssd) narrow ? launch(ws_ring_kernel<0, 16>) : launch(ws_ring_kernel<0, kRingCols>);
    else if (!g.centred) narrow ? launch(ws_ring_kernel<1, 16>) : launch(ws_ring_kernel<1, kRingCols>);
    else narrow ? launch(ws_ring_kernel<2, 16>) : launch(ws_ring_kernel<2, kRingCols>);
    return hipGetLastError();
}

hipError_t launch_generic(const GenericArgs &g, hipStream_t s)
{
    const int ow = g.view == 0 ? g.w1 : g.w2, oh = g.view == 0 ? g.h1 : g.h2;
    const long long inside = (long long)(g.skip_x1 - g.skip_x0) * (g.skip_y1 - g.skip_y0);
    const long long n = (long long)ow * oh - (inside > 0 ? inside : 0);
    if (n <= 0) return hipSuccess;
    dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(ws_generic_kernel, grid, dim3(256), 0, s, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// LinearSearch (LinearSearch.cpp:10-59): single pixels, no window.  A workgroup takes 256
// consecutive columns of one row, packs the left row's segment [x0, x0 + 256 + range) into LDS
// once (pixel dword + its sum of squares), and every thread sweeps its candidates from there:
// (a-b)^2 summed over the channels = a.a + b.b - 2 a.b, one v_dot4 per candidate.
// ------------------------------------------------------------------------------------------
constexpr int kLinearMaxRange = 4096;

__global__ void __launch_bounds__(256) ws_linear_kernel(const GenericArgs g)
{
    extern __shared__ uint32_t lin_lds[]; // [n] pixels, [n] sums of squares
    const int y = blockIdx.y, x0 = blockIdx.x * 256, tid = threadIdx.x;
    const int n = 256 + g.linear_range;
    uint32_t *lp = lin_lds, *lq = lin_lds + n;
    if (y < g.h1) {
        const uint8_t *lrow = g.L + (size_t)y * g.s1;
        for (int i = tid; i < n; i += 256) {
            const int k = x0 + i;
            uint32_t v = 0;
            if (k < g.w1) v = (uint32_t)lrow[3 * k] | ((uint32_t)lrow[3 * k + 1] << 8) | ((uint32_t)lrow[3 * k + 2] << 16);
            lp[i] = v;
            lq[i] = pix_dot<false>(v, v, 0u);
        }
    }
    __syncthreads();
    const int x = x0 + tid;
    if (x >= g.w2) return;
    float val = 0.0f;
    // the black test reads the LEFT pixel (LinearSearch.cpp:24), where there is one
    if (y < g.h1 && !(x < g.w1 && lp[tid] == 0u)) {
        const uint8_t *pr = g.R + (size_t)y * g.s2 + 3 * x;
        const uint32_t a = (uint32_t)pr[0] | ((uint32_t)pr[1] << 8) | ((uint32_t)pr[2] << 16);
        const uint32_t aa = pix_dot<false>(a, a, 0u);
        uint32_t best = 0xffffffffu;
        int bd = -x; // no candidate: col stays 0 (LinearSearch.cpp:33,53)
        const int d_end = min(g.linear_range, g.w1 - x); // candidates beyond the row's end are skipped, not read
#pragma unroll 4
        for (int d = g.min_d; d < d_end; ++d) {
            const uint32_t c = aa + lq[tid + d] - 2u * pix_dot<false>(a, lp[tid + d], 0u);
            if (c < best) { best = c; bd = d; } // strict: ties go to the smaller d
        }
        val = (float)bd;
    }
    if (g.out16) g.out16[(size_t)y * g.out_pitch + x] = (int16_t)(int)val;
    else g.out[(size_t)y * g.out_pitch + x] = val;
}

hipError_t launch_linear(const GenericArgs &g, hipStream_t s)
{
    if (g.linear_range > kLinearMaxRange) return launch_generic(g, s);
    dim3 grid(ceil_div(g.w2, 256), g.h2);
    hipLaunchKernelGGL(ws_linear_kernel, grid, dim3(256), (size_t)(256 + g.linear_range) * 8, s, g);
    return hipGetLastError();
}

// Sub-pixel refinement (build extension): the integer map is already final; a
// pixel is refined when d-1, d and d+1 are all candidates the search itself would have tried.
__global__ void __launch_bounds__(256) ws_refine_kernel(const GenericArgs g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    // the marching interior, when there is one, is refined on the packed planes (below)
    if (x >= g.skip_x0 && x < g.skip_x1 && y >= g.skip_y0 && y < g.skip_y1) return;
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

// The same refinement for the marching interior (complete windows) on the packed planes: one
// v_sad_u8 / two v_dot4 per pixel pair instead of byte arithmetic.  sum a^2 cancels in both the
// numerator and the denominator, so only sum b^2 - 2 sum ab is needed for the three candidates.
struct RefineArgs {
    const uint32_t *A;
    const uint32_t *B;
    int pitch_a, pad_a, pitch_b, pad_b;
    int wa, ww, wh, wx0, wy0, boff;
    int d_lo, d_hi, b_lo, b_hi;
    int ox0, ox1, oy0, oy1;
    int ssd, centred, mirror;
    float *out;
    int out_pitch;
};

template <bool SSD, bool CENTRED, int WW = 0> // WW: the window width when it is known to the compiler (a row's loads
                                              // are then all in flight together), 0 = g.ww
__global__ void __launch_bounds__(256) ws_refine_planes_kernel(const RefineArgs g)
{
    const int x = g.ox0 + blockIdx.x * blockDim.x + threadIdx.x; // canonical column
    const int y = g.oy0 + blockIdx.y;
    if (x >= g.ox1 || y >= g.oy1) return;
    if (g.A[(size_t)y * g.pitch_a + x + g.pad_a] == (CENTRED ? kCentre : 0u)) return; // black pixel
    float *o = g.out + (size_t)y * g.out_pitch + (g.mirror ? g.wa - 1 - x : x);
    const int d = (int)*o;
    const int xb = x - d + g.boff; // target centre of the winner
    if (d < g.d_lo || d > g.d_hi || xb < g.b_lo || xb > g.b_hi) return; // fallback value, not a match
    if (d - 1 < g.d_lo || d + 1 > g.d_hi || xb - 1 < g.b_lo || xb + 1 > g.b_hi) return;
    // costs at d-1 (target column +1), d, d+1 (target column -1); the three target windows overlap,
    // so a row costs ww + 2 target loads; for SSD the three sums of b^2 come from the same pixels (the marching
    // kernel's bias rows live in its LDS only and carry a per-strip correction term besides them: not read here)
    long long cm = 0, c0 = 0, cp = 0;
    for (int r = 0; r < g.wh; ++r) {
        const uint32_t *pa = g.A + (size_t)(y + g.wy0 + r) * g.pitch_a + (x + g.wx0 + g.pad_a);
        const uint32_t *pb = g.B + (size_t)(y + g.wy0 + r) * g.pitch_b + (xb + g.wx0 + g.pad_b);
        uint32_t sm = 0, s0 = 0, sp = 0;
        uint32_t vp = pb[-1], v0 = pb[0];
        uint32_t qt = 0; // SSD: sum of b^2 over pb[1 .. ww], the window of d - 1
        const uint32_t q_m1 = SSD ? pix_dot<CENTRED>(vp, vp, 0u) : 0u, q_0 = SSD ? pix_dot<CENTRED>(v0, v0, 0u) : 0u;
        auto pixel = [&](int i) {
            const uint32_t a = pa[i], vm = pb[i + 1];
            if constexpr (SSD) {
                sm = pix_dot<CENTRED>(a, vm, sm); s0 = pix_dot<CENTRED>(a, v0, s0); sp = pix_dot<CENTRED>(a, vp, sp);
                qt = pix_dot<CENTRED>(vm, vm, qt);
            } else {
                sm = pix_sad(a, vm, sm); s0 = pix_sad(a, v0, s0); sp = pix_sad(a, vp, sp);
            }
            vp = v0; v0 = vm;
        };
        if constexpr (WW > 0) {
#pragma unroll
            for (int i = 0; i < WW; ++i) pixel(i);
        } else {
#pragma unroll 4
            for (int i = 0; i < g.ww; ++i) pixel(i);
        }
        if constexpr (SSD) {
            // (after the loop v0 = pb[ww], vp = pb[ww - 1]) windows: d - 1 = pb[1 .. ww], d = pb[0 .. ww - 1], d + 1 = pb[-1 .. ww - 2]
            const uint32_t b0 = qt - pix_dot<CENTRED>(v0, v0, 0u) + q_0;
            const uint32_t bp = b0 - pix_dot<CENTRED>(vp, vp, 0u) + q_m1;
            cm += (long long)qt - 2LL * (int32_t)sm; c0 += (long long)b0 - 2LL * (int32_t)s0; cp += (long long)bp - 2LL * (int32_t)sp;
        } else {
            cm += sm; c0 += s0; cp += sp;
        }
    }
    const long long num = cm - cp, den = cm - 2 * c0 + cp;
    if (den > 0) *o = *o + (float)((double)num / (2.0 * (double)den));
}

hipError_t launch_refine_planes(const Canon &c, const MarchLaunch &, Plane a, Plane b, float *out, int out_pitch, hipStream_t s)
{
    RefineArgs g{};
    g.A = a.data; g.B = b.data;
    g.pitch_a = a.pitch; g.pad_a = a.pad; g.pitch_b = b.pitch; g.pad_b = b.pad;
    g.wa = c.wa; g.ww = c.ww; g.wh = c.wh; g.wx0 = c.wx0; g.wy0 = c.wy0; g.boff = c.boff;
    g.d_lo = c.d_lo; g.d_hi = c.d_hi; g.b_lo = c.b_lo; g.b_hi = c.b_hi;
    g.ox0 = c.ox0; g.ox1 = c.ox1; g.oy0 = c.oy0; g.oy1 = c.oy1;
    g.ssd = c.ssd; g.centred = march_centred(c); g.mirror = c.mirror;
    g.out = out; g.out_pitch = out_pitch;
    dim3 grid(ceil_div(c.ox1 - c.ox0, 256), c.oy1 - c.oy0);
    // (the window widths of the BASELINE configs with the sub-pixel extension get the compile-time form)
#define WS_REFINE(SSD, CEN)                                                                                       \
    switch (c.ww) {                                                                                               \
    case 7: hipLaunchKernelGGL((ws_refine_planes_kernel<SSD, CEN, 7>), grid, dim3(256), 0, s, g); break;          \
    case 9: hipLaunchKernelGGL((ws_refine_planes_kernel<SSD, CEN, 9>), grid, dim3(256), 0, s, g); break;          \
    default: hipLaunchKernelGGL((ws_refine_planes_kernel<SSD, CEN, 0>), grid, dim3(256), 0, s, g); break;         \
    }
    if (!c.ssd) { WS_REFINE(false, false) }
    else if (g.centred) { WS_REFINE(true, true) }
    else { WS_REFINE(true, false) }
#undef WS_REFINE
    return hipGetLastError();
}

hipError_t launch_refine(const GenericArgs &g, hipStream_t s)
{
    const int ow = g.view == 0 ? g.w1 : g.w2, oh = g.view == 0 ? g.h1 : g.h2;
    dim3 grid(ceil_div(ow, 256), oh);
    hipLaunchKernelGGL(ws_refine_kernel, grid, dim3(256), 0, s, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// varBlock (BlockSearch.cpp:125-145, right view): while the window's centred norm is below
// `thres` the block grows by 4; then the search runs with that pixel's own window.  One wavefront
// per pixel: the lanes share the window pixels for the texture test; where the window grew (no
// sliding sums: it differs from pixel to pixel) they split the disparities for the search, wave-wide
// sums / min by shuffles.  Pixels whose window did not grow keep the ordinary search's result,
// which is in the map already.
// cv::mean / cv::subtract / cv::norm semantics (OpenCV 4.x restated; the library is
// un-vendored): double mean per channel, saturate_cast<uchar>(round-half-even((float)p - (float)mean))
// -- the float32 path OpenCV takes for a u8 Mat minus a non-integer Scalar --, L2 norm.
// Growth stops when the window no longer changes (the reference would loop forever there).
// ------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) ws_varblock_kernel(const GenericArgs g, double thres,
                                                          int16_t *__restrict__ bs_plane, int bs_pitch,
                                                          int *__restrict__ max_block)
{
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long long)g.w2 * g.h2) return; // uniform per wave
    const int x = (int)(pix % g.w2), y = (int)(pix / g.w2);
    const int height = min(g.h1, g.h2);
    float val = 0.0f;
    int bs = g.block_size;
    if (y < height && !black3(g.R + (size_t)y * g.s2 + 3 * x)) {
        int hb = (bs - 1) / 2;
        int left = min(x, hb), right = min(g.w2 - x - 1, hb), up = min(y, hb), down = min(g.h2 - y - 1, hb);
        for (;;) {
            const int ww = left + right, wh = up + down, n = ww * wh;
            double nrm = 0.0;
            if (n > 0) {
                const uint8_t *w0 = g.R + (size_t)(y - up) * g.s2 + 3 * (x - left);
                unsigned long long s0 = 0, s1 = 0, s2 = 0;
                for (int i = lane; i < n; i += 64) {
                    const uint8_t *p = w0 + (size_t)(i / ww) * g.s2 + 3 * (i % ww);
                    s0 += p[0]; s1 += p[1]; s2 += p[2];
                }
                const double area = (double)ww * (double)wh;
                const double m0 = (double)wave_sum_u64(s0) / area, m1 = (double)wave_sum_u64(s1) / area,
                             m2 = (double)wave_sum_u64(s2) / area;
                // cv::subtract(Mat_u8, Scalar) with a non-integer Scalar works in float32 (the Scalar narrowed to
                // float, the pixel widened), then saturate_cast<uchar>(cvRound(.)), round half to even
                const float f0 = (float)m0, f1 = (float)m1, f2 = (float)m2;
                unsigned long long acc = 0;
                for (int i = lane; i < n; i += 64) {
                    const uint8_t *p = w0 + (size_t)(i / ww) * g.s2 + 3 * (i % ww);
                    const int v0 = min(255, max(0, (int)rintf((float)p[0] - f0)));
                    const int v1 = min(255, max(0, (int)rintf((float)p[1] - f1)));
                    const int v2 = min(255, max(0, (int)rintf((float)p[2] - f2)));
                    acc += (unsigned long long)(v0 * v0 + v1 * v1 + v2 * v2);
                }
                nrm = sqrt((double)wave_sum_u64(acc));
            }
            if (!(nrm < thres)) break;
            bs += 4;
            hb = (bs - 1) / 2;
            const int l2 = min(x, hb), r2 = min(g.w2 - x - 1, hb), u2 = min(y, hb), d2 = min(g.h2 - y - 1, hb);
            if (l2 == left && r2 == right && u2 == up && d2 == down) break; // cannot grow any more
            left = l2; right = r2; up = u2; down = d2;
        }
        // A window that did not grow is the ordinary right-view window: the marching / ring kernels
        // have searched it already (launch order in ws_capi.cpp), only the block size is recorded.
        if (bs == g.block_size) {
            if (lane == 0) bs_plane[(size_t)y * bs_pitch + x] = (int16_t)bs;
            return;
        }
        // the search with this pixel's window: lanes over d, d ascending inside a lane
        const int ww = left + right, wh = up + down;
        unsigned long long bcost = ~0ull;
        int bd = 0x7fffffff;
        if (ww > 0 && wh > 0) {
            const uint8_t *rw = g.R + (size_t)(y - up) * g.s2 + 3 * (x - left);
            const int d_end = min(g.max_d - 1, g.w1 - right - x - 1);
            for (int d = g.min_d + lane; d <= d_end; d += 64) {
                const uint8_t *lw = g.L + (size_t)(y - up) * g.s1 + 3 * (x + d - left);
                const unsigned long long c = window_cost64(lw, g.s1, rw, g.s2, ww, wh, g.ssd);
                if (c < bcost) { bcost = c; bd = d; }
            }
        }
        for (int off = 32; off >= 1; off >>= 1) { // wave-wide lexicographic min of (cost, d)
            const unsigned long long oc = __shfl_xor(bcost, off, 64);
            const int od = __shfl_xor(bd, off, 64);
            if (oc < bcost || (oc == bcost && od < bd)) { bcost = oc; bd = od; }
        }
        val = bd == 0x7fffffff ? -(float)x : (float)bd;
        if (lane == 0 && bs > g.block_size) atomicMax(max_block, bs);
    }
    if (lane == 0) {
        g.out[(size_t)y * g.out_pitch + x] = val;
        bs_plane[(size_t)y * bs_pitch + x] = (int16_t)min(bs, 32767);
    }
}

hipError_t launch_varblock(const GenericArgs &g, double thres, int16_t *bs_plane, int bs_pitch, int *max_block,
                           hipStream_t s)
{
    hipError_t e = hipMemsetAsync(max_block, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    const long long n = (long long)g.w2 * g.h2;
    hipLaunchKernelGGL(ws_varblock_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, g, thres, bs_plane, bs_pitch,
                       max_block);
    return hipGetLastError();
}

} // namespace wsamd
