// ws_smooth.hip -- smoothFactor != 1: raster-order passes for the three views
// Part of the gfx950 kernels of the WindowSearch hot path; overview in ws_march.hip.
#include "ws_device.h"

namespace wsamd {

// ------------------------------------------------------------------------------------------
// smoothFactor != 1 for the right view and LinearSearch
//
// There the reference compares a neighbour's stored value (>= 0, or the negative fallback) with
// x - cx = -d (BlockSearch.cpp:160-165, LinearSearch.cpp:39-44), so the factor can only ever hit
// d = 0 next to a neighbour whose stored value is 0 (the fallback cases cannot coincide with a
// pixel that still has candidates).  Since d = 0 is tried first, the pixel's value is
//     0            if  !(c1 < c0 * s^k)      k = [up == 0] + [left == 0]
//     argmin_{d>=1}  otherwise
// with c0 / c1 the reference's doubles (sqrt, / area, successive multiplications).  The search for
// d >= 1 is the ordinary data-parallel search; what is left is a boolean recurrence in raster
// order, solved row by row with a scan over function composition.
// ------------------------------------------------------------------------------------------
constexpr uint8_t kSelFixed = 0x80; // value in the map is final; otherwise bits 0..2 = t_0..t_2
constexpr uint8_t kSelZero = 0x40;  // with kSelFixed: that final value is 0

// The three outcomes of a pixel, t_k = "the value is NOT 0 when k of its two neighbours hold 0", replaying
// the reference's running minimum (BlockSearch.cpp:160-171, LinearSearch.cpp:39-49): d = 0 comes first with
// dist0 = e0 * s (* s) -- successive multiplications -- and is accepted when dist0 < DBL_MAX, so an infinite
// or NaN product (s = +-inf, 0 * inf, overflow) REFUSES d = 0; afterwards the best d >= 1 (distance e1,
// no factor ever reaches it) replaces it when e1 < dist0.  Without any d >= 1 candidate (has_d1 false) the
// refused d = 0 leaves minimumCorrespondX = 0, i.e. the fallback -x the d >= 1 search already stored.
__device__ __forceinline__ uint8_t smooth_bits(double e0, double e1, bool has_d1, double s)
{
    uint8_t code = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!(e0 < 1.7976931348623157e308) || (has_d1 && e1 < e0)) code |= (uint8_t)(1u << k);
        e0 *= s;
    }
    return code;
}

__global__ void __launch_bounds__(256) ws_smooth_prepare_kernel(const GenericArgs g, double s,
                                                                uint8_t *__restrict__ sel, int sel_pitch)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= g.w2 || y >= g.h2) return;
    float *o = g.out + (size_t)y * g.out_pitch + x;
    uint8_t code = kSelFixed;
    float val = 0.0f;
    const int height = min(g.h1, g.h2);
    if (g.view == 1) {
        if (y < height && !black3(g.R + (size_t)y * g.s2 + 3 * x)) {
            const int bsz = g.bs_plane ? (int)g.bs_plane[(size_t)y * g.bs_pitch + x] : g.block_size;
            const int half = (bsz - 1) / 2;
            const int left = min(x, half), right = min(g.w2 - x - 1, half);
            const int up = min(y, half), down = min(g.h2 - y - 1, half);
            const int ww = left + right, wh = up + down;
            const bool any = ww > 0 && wh > 0 && g.max_d > 0 && x + right < g.w1;
            if (!any) {
                val = -(float)x; // no candidate at all: stores -x (BlockSearch.cpp:174)
            } else {
                // d = 0 may be the only candidate: then the d >= 1 search left its fallback -x in the map,
                // which is also what the reference stores when d = 0 is refused (below)
                const bool has_d1 = g.max_d > 1 && x + 1 + right < g.w1;
                const uint8_t *rw = g.R + (size_t)(y - up) * g.s2 + 3 * (x - left);
                const int d1 = has_d1 ? (int)*o : 0;
                const unsigned long long c0 = window_cost64(g.L + (size_t)(y - up) * g.s1 + 3 * (x - left), g.s1, rw, g.s2, ww, wh, g.ssd);
                const unsigned long long c1 = has_d1 ? window_cost64(g.L + (size_t)(y - up) * g.s1 + 3 * (x + d1 - left), g.s1, rw, g.s2, ww, wh, g.ssd) : 0ull;
                const double area = (double)(ww * wh);
                const double e1 = (g.ssd ? sqrt((double)c1) : (double)c1) / area;
                const double e0 = (g.ssd ? sqrt((double)c0) : (double)c0) / area;
                code = smooth_bits(e0, e1, has_d1, s);
                val = has_d1 ? (float)d1 : -(float)x;
                if (!has_d1 && x == 0) { code = kSelFixed; val = 0.0f; } // refused or not, column 0 stores 0 - 0
            }
        }
    } else { // LinearSearch: black test on the left pixel, distance of single pixels
        if (y < g.h1 && !(x < g.w1 && black3(g.L + (size_t)y * g.s1 + 3 * x))) {
            if (!(x < g.w1)) {
                val = -(float)x;
            } else {
                const bool has_d1 = g.linear_range > 1 && x + 1 < g.w1;
                const uint8_t *pr = g.R + (size_t)y * g.s2 + 3 * x;
                const int d1 = has_d1 ? (int)*o : 0;
                const uint32_t c0 = window_cost(pr, 0, g.L + (size_t)y * g.s1 + 3 * x, 0, 1, 1, 1);
                const uint32_t c1 = has_d1 ? window_cost(pr, 0, g.L + (size_t)y * g.s1 + 3 * (x + d1), 0, 1, 1, 1) : 0u;
                code = smooth_bits(sqrt((double)c0), sqrt((double)c1), has_d1, s);
                val = has_d1 ? (float)d1 : -(float)x;
                if (!has_d1 && x == 0) { code = kSelFixed; val = 0.0f; } // refused or not, column 0 stores 0 - 0
            }
        }
    }
    *o = val;
    if ((code & kSelFixed) && val == 0.0f) code |= kSelZero;
    sel[(size_t)y * sel_pitch + x] = code;
}

// The same for the right view when the marching kernel ran, on the packed, mirrored planes.  The
// winner's cost c1 comes from the search itself (cost plane; for SSD without the sum of a^2), so
// what is left per pixel is the cost of d = 0 and, for SSD, the sum of a^2 over the window: two
// box sums.  Interior (complete windows): separable box filter through LDS, a workgroup per
// 64 x 32 tile.  Border ring (clipped windows): one thread per ring pixel.
struct PreparePlanesArgs {
    const uint32_t *A;
    const uint32_t *B;
    int pitch_a, pad_a, pitch_b, pad_b;
    int wa, ha, wb, height, half, max_d; // ring: plane sizes, min(h1,h2), (bs-1)/2, maxDisparity
    int ww, wh, wx0, wy0, boff;
    int d_hi, b_lo;
    int ox0, ox1, oy0, oy1;                 // marching interior, canonical
    int skip_x0, skip_x1, skip_y0, skip_y1; // the same, ORIGINAL coordinates
    double s;
    float *out;
    int out_pitch;
    const int32_t *cost;
    int cost_pitch;
    uint8_t *sel;
    int sel_pitch;
};

// the three outcomes t_k = [c1 < c0 * s^k] with the reference's doubles
template <bool SSD>
__device__ __forceinline__ uint8_t smooth_code(long long c0, long long c1, bool has_d1, double area, double s)
{
    const double e1 = (SSD ? sqrt((double)c1) : (double)c1) / area;
    const double e0 = (SSD ? sqrt((double)c0) : (double)c0) / area;
    return smooth_bits(e0, e1, has_d1, s);
}

constexpr int kBoxRows = 16, kBoxMaxW = 16; // tile rows (32: 18.6 us, 16: 15.3, 8: 15.3 at 900 x 750, 17 x 17); widest / tallest right-view marching window
constexpr int kBoxBatch = 4;                // rows a thread has in flight while a tile's pixel terms are staged

template <bool SSD, bool CENTRED>
__global__ void __launch_bounds__(256) ws_smooth_prepare_box_kernel(const PreparePlanesArgs g)
{
    __shared__ uint32_t e0[kBoxRows + kBoxMaxW - 1][64 + kBoxMaxW];
    __shared__ uint32_t ea[SSD ? kBoxRows + kBoxMaxW - 1 : 1][64 + kBoxMaxW];
    __shared__ uint32_t h0[kBoxRows + kBoxMaxW - 1][64];
    __shared__ uint32_t ha[SSD ? kBoxRows + kBoxMaxW - 1 : 1][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int xt = g.ox0 + blockIdx.x * 64, y0 = g.oy0 + blockIdx.y * kBoxRows;
    const int y1 = min(y0 + kBoxRows, g.oy1);
    const int nrows = (y1 - y0) + g.wh - 1, ncols = 64 + g.ww - 1;
    // (1) per pixel: cost against the d = 0 partner, and a^2.  A wave takes every 4th row, kBoxBatch of them at a
    // time with all their loads in flight (clamped addresses, no branches): with one or two workgroups on a CU the
    // tile is bound by the trips to memory, not by their width.
    for (int cx = tx; cx < ncols; cx += 64) {
        const int ca = xt + g.wx0 + cx + g.pad_a, cb = xt + g.wx0 + cx + g.boff + g.pad_b;
        const bool ina = ca >= 0 && ca < g.pitch_a, inb = cb >= 0 && cb < g.pitch_b;
        for (int k0 = ty; k0 < nrows; k0 += 4 * kBoxBatch) {
            uint32_t a[kBoxBatch], b[kBoxBatch];
#pragma unroll
            for (int i = 0; i < kBoxBatch; ++i) {
                const size_t row = (size_t)(y0 + g.wy0 + min(k0 + 4 * i, nrows - 1));
                a[i] = g.A[row * g.pitch_a + (ina ? ca : 0)];
                b[i] = g.B[row * g.pitch_b + (inb ? cb : 0)];
            }
#pragma unroll
            for (int i = 0; i < kBoxBatch; ++i) {
                const int k = k0 + 4 * i;
                if (k >= nrows) break;
                const uint32_t pa = ina ? a[i] : 0u, pb = inb ? b[i] : 0u;
                if constexpr (SSD) {
                    const uint32_t aa = pix_dot<CENTRED>(pa, pa, 0u);
                    e0[k][cx] = aa + pix_dot<CENTRED>(pb, pb, 0u) - 2u * pix_dot<CENTRED>(pa, pb, 0u); // (a-b)^2 >= 0
                    ea[k][cx] = aa;
                } else {
                    e0[k][cx] = pix_sad(pa, pb, 0u);
                }
            }
        }
    }
    // (what the decisions of stage 3 read from memory: asked for now, used after two barriers)
    constexpr int seg = kBoxRows / 4;
    const int x = xt + tx; // canonical (mirrored) column
    const int ya = y0 + ty * seg, yb = min(ya + seg, y1);
    const int xo = g.wa - 1 - x; // original column
    const bool mine = x < g.ox1 && ya < yb;
    uint32_t apx[seg];
    int32_t cst[seg];
    float prev[seg];
#pragma unroll
    for (int i = 0; i < seg; ++i) {
        const int y = mine ? min(ya + i, yb - 1) : g.oy0, xx = mine ? x : g.ox0, xxo = g.wa - 1 - xx;
        apx[i] = g.A[(size_t)y * g.pitch_a + xx + g.pad_a];
        cst[i] = g.cost[(size_t)y * g.cost_pitch + xxo];
        prev[i] = g.out[(size_t)y * g.out_pitch + xxo];
    }
    __syncthreads();
    // (2) horizontal window sums
    for (int k = ty; k < nrows; k += 4) {
        uint32_t s0 = 0, sa = 0;
#pragma unroll 8
        for (int i = 0; i < g.ww; ++i) {
            s0 += e0[k][tx + i];
            if constexpr (SSD) sa += ea[k][tx + i];
        }
        h0[k][tx] = s0;
        if constexpr (SSD) ha[k][tx] = sa;
    }
    __syncthreads();
    // (3) vertical sliding sums and the decision
    if (!mine) return;
    uint32_t c0 = 0, sa = 0;
    for (int k = 0; k < g.wh; ++k) {
        c0 += h0[ya - y0 + k][tx];
        if constexpr (SSD) sa += ha[ya - y0 + k][tx];
    }
    const double area = (double)(g.ww * g.wh);
    const int xb0 = x + g.boff; // target centre of d = 0
#pragma unroll
    for (int i = 0; i < seg; ++i) {
        const int y = ya + i;
        if (y >= yb) break;
        float *o = g.out + (size_t)y * g.out_pitch + xo;
        uint8_t code = kSelFixed;
        float val = 0.0f;
        if (apx[i] != (CENTRED ? kCentre : 0u)) {
            if (!(g.d_hi >= 0 && xb0 >= g.b_lo)) {
                val = -(float)xo; // no candidate at all
            } else {
                // the search's d >= 1 winner -- or, when d = 0 is the only candidate, its fallback -x --
                // stays unless the recurrence says 0
                const bool has_d1 = g.d_hi >= 1 && xb0 - 1 >= g.b_lo;
                val = prev[i];
                const long long c1 = has_d1 ? (long long)cst[i] + (SSD ? (long long)sa : 0LL) : 0LL;
                code = smooth_code<SSD>((long long)c0, c1, has_d1, area, g.s);
            }
        }
        *o = val;
        if ((code & kSelFixed) && val == 0.0f) code |= kSelZero;
        g.sel[(size_t)y * g.sel_pitch + xo] = code;
        const int k = y - y0;
        if (y + 1 < yb) {
            c0 += h0[k + g.wh][tx] - h0[k][tx];
            if constexpr (SSD) sa += ha[k + g.wh][tx] - ha[k][tx];
        }
    }
}

// The same for 32-bit terms (modulo 2^32: exact, signed or not, whenever the total fits), all 64 lanes active: six DPP
// additions -- Hillis-Steele inside the rows of 16 lanes, the row totals across the rows -- instead of six trips through
// the LDS crossbar; the total is lane 63's.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // lane 15 of rows 0, 2 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // lane 31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// (ring_pixel's enumeration in 32-bit arithmetic, for callers whose pixel count fits: a 64-bit division is ~100 instructions)
__device__ __forceinline__ bool ring_pixel32(const GenericArgs &g, int ow, int oh, uint32_t idx, int *px, int *py)
{
    const uint32_t w = (uint32_t)ow;
    const uint32_t n_top = (uint32_t)g.skip_y0 * w, n_bot = (uint32_t)(oh - g.skip_y1) * w;
    const uint32_t side = (uint32_t)(g.skip_x0 + (ow - g.skip_x1));
    const uint32_t n_side = (uint32_t)(g.skip_y1 - g.skip_y0) * side;
    if (idx < n_top) {
        *py = (int)(idx / w);
        *px = (int)(idx % w);
    } else if (idx < n_top + n_bot) {
        idx -= n_top;
        *py = g.skip_y1 + (int)(idx / w);
        *px = (int)(idx % w);
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

template <bool SSD, bool CENTRED>
__global__ void __launch_bounds__(256) ws_smooth_prepare_ring_kernel(const PreparePlanesArgs g)
{
    // one wave per ring pixel: the lanes share the (clipped) window's pixels, wave-wide sums
    // (26 k waves; 16 pixels per wave in turn, with one thread per pixel for the decision, measured
    // slower: 26 vs 18 us -- the window sums are latency-bound and want the parallelism)
    GenericArgs e{}; // only the skip rectangle is used by ring_pixel
    e.skip_x0 = g.skip_x0; e.skip_x1 = g.skip_x1; e.skip_y0 = g.skip_y0; e.skip_y1 = g.skip_y1;
    const int lane = threadIdx.x & 63;
    const uint32_t pix = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (launch_smooth: the ring's pixel count is an int)
    int x, y; // original coordinates
    if (!ring_pixel32(e, g.wa, g.ha, pix, &x, &y)) return; // uniform per wave
    const int xm = g.wa - 1 - x;
    float *o = g.out + (size_t)y * g.out_pitch + x;
    uint8_t code = kSelFixed;
    float val = 0.0f;
    if (y < g.height && g.A[(size_t)y * g.pitch_a + xm + g.pad_a] != (CENTRED ? kCentre : 0u)) {
        const int left = min(x, g.half), right = min(g.wa - x - 1, g.half);
        const int up = min(y, g.half), down = min(g.ha - y - 1, g.half);
        const int ww = left + right, wh = up + down;
        const bool any = ww > 0 && wh > 0 && g.max_d > 0 && x + right < g.wb;
        if (!any) {
            val = -(float)x; // no candidate at all: stores -x (BlockSearch.cpp:174)
        } else {
            const bool has_d1 = g.max_d > 1 && x + 1 + right < g.wb; // else d = 0 is the only candidate
            const int ca = xm - right + 1; // first window column, canonical
            const uint32_t *pa0 = g.A + (size_t)(y - up) * g.pitch_a + ca + g.pad_a;
            const uint32_t *pb0 = g.B + (size_t)(y - up) * g.pitch_b + ca + g.boff + g.pad_b;
            uint32_t aa = 0, bb = 0, ab = 0; // a lane sees at most (16 * 16) / 64 pixels: 32 bits hold
            // the window is at most 16 x 16 (launch_smooth): lane = (row % 4, column), four rows each, their loads in
            // flight together (clamped, not branched around: the wave waits for memory once, not four times)
            const int c = lane & 15, r4 = lane >> 4;
            uint32_t a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = c < ww && r4 + 4 * i < wh;
                const int r = ok ? r4 + 4 * i : 0, cc = ok ? c : 0;
                a[i] = pa0[(size_t)r * g.pitch_a + cc];
                b[i] = pb0[(size_t)r * g.pitch_b + cc];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!(c < ww && r4 + 4 * i < wh)) continue;
                if constexpr (SSD) {
                    aa = pix_dot<CENTRED>(a[i], a[i], aa);
                    bb = pix_dot<CENTRED>(b[i], b[i], bb);
                    ab = pix_dot<CENTRED>(a[i], b[i], ab);
                } else {
                    ab = pix_sad(a[i], b[i], ab);
                }
            }
            // (a window is at most 16 x 16 pixels: every total -- at most 256 * 3 * 255^2 -- fits 32 bits with room to spare)
            long long c0, sa = 0;
            if constexpr (SSD) {
                const long long taa = (long long)(int32_t)wave_sum_u32(aa), tbb = (long long)(int32_t)wave_sum_u32(bb);
                c0 = taa + tbb - 2LL * (long long)(int32_t)wave_sum_u32(ab);
                sa = taa;
            } else {
                c0 = (long long)wave_sum_u32(ab);
            }
            val = *o; // (without a d >= 1 candidate: the search's fallback -x)
            const long long c1 = has_d1 ? (long long)g.cost[(size_t)y * g.cost_pitch + x] + sa : 0LL;
            code = smooth_code<SSD>(c0, c1, has_d1, (double)(ww * wh), g.s);
            if (!has_d1 && x == 0) { code = kSelFixed; val = 0.0f; } // refused or not, column 0 stores 0 - 0
        }
    }
    if (lane == 0) {
        *o = val;
        if ((code & kSelFixed) && val == 0.0f) code |= kSelZero;
        g.sel[(size_t)y * g.sel_pitch + x] = code;
    }
}

// compose two maps {0,1}->{0,1} stored as bit0 = f(0), bit1 = f(1):  (b o a)(v) = b(a(v))
__device__ __forceinline__ uint32_t compose2(uint32_t a, uint32_t b)
{
    const uint32_t r0 = (b >> (a & 1)) & 1, r1 = (b >> ((a >> 1) & 1)) & 1;
    return r0 | (r1 << 1);
}

// One workgroup walks the rows in order.  Per row every thread owns `per` consecutive columns,
// composes their transition maps, the workgroup scans the compositions, and each thread replays
// its columns with the incoming "left neighbour is 0" bit.
__global__ void __launch_bounds__(1024) ws_smooth_resolve_kernel(float *out, int out_pitch, int w, int rows,
                                                                 const uint8_t *__restrict__ sel, int sel_pitch)
{
    extern __shared__ uint8_t zrow[]; // zero flags of the previous row, then 16 words of wave totals
    uint32_t *wave_tot = reinterpret_cast<uint32_t *>(zrow + ((w + 3) & ~3));
    const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (w + nt - 1) / nt;
    const int x0 = tid * per, x1 = min(x0 + per, w);
    for (int x = tid; x < w; x += nt) zrow[x] = 0; // y = 0 has no upper neighbour
    __syncthreads();
    for (int y = 0; y < rows; ++y) {
        const uint8_t *srow = sel + (size_t)y * sel_pitch;
        float *orow = out + (size_t)y * out_pitch;
        // 1. composition of this thread's columns
        uint32_t f = 2; // identity: f(0)=0, f(1)=1
        for (int x = x0; x < x1; ++x) {
            const uint32_t c = srow[x];
            uint32_t gmap;
            if (c & kSelFixed) {
                const uint32_t z = orow[x] == 0.0f;
                gmap = z | (z << 1);
            } else {
                const uint32_t zu = zrow[x];
                gmap = (((c >> zu) & 1) ^ 1) | ((((c >> (zu + 1)) & 1) ^ 1) << 1);
            }
            f = compose2(f, gmap);
        }
        // 2. exclusive scan of the compositions over the workgroup
        uint32_t incl = f;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t prev = __shfl_up(incl, off, 64);
            if (lane >= off) incl = compose2(prev, incl);
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        uint32_t before = 2; // maps of all earlier waves
        for (int k = 0; k < wv; ++k) before = compose2(before, wave_tot[k]);
        uint32_t excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 2;
        excl = compose2(before, excl);
        uint32_t b = excl & 1; // column 0 has no left neighbour: start from 0
        // 3. replay
        for (int x = x0; x < x1; ++x) {
            const uint32_t c = srow[x];
            uint32_t z;
            if (c & kSelFixed) {
                z = orow[x] == 0.0f;
            } else {
                const uint32_t k = zrow[x] + b;
                z = ((c >> k) & 1) ^ 1;
                if (z) orow[x] = 0.0f;
            }
            b = z;
        }
        __syncthreads(); // everyone has read zrow / wave_tot of this row
        b = excl & 1;
        for (int x = x0; x < x1; ++x) { // store this row's flags for the next one
            const uint32_t c = srow[x];
            const uint32_t z = orow[x] == 0.0f;
            (void)c;
            zrow[x] = (uint8_t)z;
        }
        __syncthreads();
    }
}

// The same recurrence for images up to 64 * PER columns wide, on ONE wavefront: every lane owns
// PER consecutive columns, keeps the previous row's zero flags in a 64-bit mask, and the row scan
// is six shuffles -- no barrier.  The codes arrive in chunks of rows by LDS-DMA, one chunk ahead,
// so the serial walk over the rows never waits for HBM.
template <int PER>
__global__ void __launch_bounds__(64) ws_smooth_resolve_wave_kernel(float *out, int out_pitch, int w, int rows,
                                                                    const uint8_t *__restrict__ sel, int sel_pitch,
                                                                    int chunk_rows)
{
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    extern __shared__ uint4 ws_smem4[];
    uint8_t *lds = reinterpret_cast<uint8_t *>(ws_smem4);
    const int lane = threadIdx.x;
    const int x0 = lane * PER;
    const int chunk_bytes = chunk_rows * sel_pitch; // multiple of 1024: sel_pitch % 64 == 0, chunk_rows % 16 == 0
    const int nchunks = (rows + chunk_rows - 1) / chunk_rows;
    // the sel plane is allocated with (rows rounded up to chunk_rows) rows, so whole chunks may be copied
    for (int o = lane * 16; o < chunk_bytes; o += 1024)
        __builtin_amdgcn_global_load_lds((glb_void *)(sel + o), (lds_void *)(lds + (o - lane * 16)), 16, 0, 0);
    unsigned long long zprev = 0;
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // chunk c has landed
        const uint8_t *cur_buf = lds + (c & 1) * chunk_bytes;
        if (c + 1 < nchunks) {
            const uint8_t *src = sel + (size_t)(c + 1) * chunk_bytes;
            uint8_t *dst = lds + ((c + 1) & 1) * chunk_bytes;
            for (int o = lane * 16; o < chunk_bytes; o += 1024)
                __builtin_amdgcn_global_load_lds((glb_void *)(src + o), (lds_void *)(dst + (o - lane * 16)), 16, 0, 0);
        }
        const int y_end = min((c + 1) * chunk_rows, rows);
        for (int y = c * chunk_rows; y < y_end; ++y) {
            const uint8_t *srow = cur_buf + (y - c * chunk_rows) * sel_pitch + x0;
            uint32_t cur[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) cur[k] = (x0 + k < w) ? srow[k] : kSelFixed;
            uint32_t f = 2;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t cc = cur[k];
                const uint32_t zu = (uint32_t)(zprev >> k) & 1u;
                const uint32_t zf = (cc >> 6) & 1u;
                const uint32_t gm = (cc & kSelFixed) ? (zf | (zf << 1))
                                                     : ((((cc >> zu) & 1u) ^ 1u) | ((((cc >> (zu + 1)) & 1u) ^ 1u) << 1));
                f = compose2(f, gm);
            }
            uint32_t incl = f;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t prev = __shfl_up(incl, off, 64);
                if (lane >= off) incl = compose2(prev, incl);
            }
            uint32_t excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 2;
            uint32_t b = excl & 1u; // column 0 has no left neighbour
            unsigned long long znew = 0;
            float *orow = out + (size_t)y * out_pitch + x0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t cc = cur[k];
                uint32_t z;
                if (cc & kSelFixed) {
                    z = (cc >> 6) & 1u;
                } else {
                    const uint32_t kk = ((uint32_t)(zprev >> k) & 1u) + b;
                    z = ((cc >> kk) & 1u) ^ 1u;
                    if (z) orow[k] = 0.0f;
                }
                znew |= (unsigned long long)z << k;
                b = z;
            }
            zprev = znew;
        }
    }
}

// ------------------------------------------------------------------------------------------
// smoothFactor != 1 for the LEFT view (BlockSearch.cpp:68-73): there the factor reaches any
// candidate d that equals the upper / left neighbour's stored value -- a true dependency on the
// neighbours' values in raster order.  Only those two candidates are ever touched, so:
//   * 0 <= s <= 1: a discounted candidate only gets cheaper, the winner is one of
//     { d1 = the undiscounted argmin, up, left } (every other d is no better than d1 and loses the
//     tie by the reference's own rule).  The ordinary data-parallel search gives d1.
//   * any other s can make a candidate DEARER: the winner is up, left or one of the THREE best
//     untouched candidates in the reference's order (cost ascending, then d descending), kept per
//     pixel by ws_left_top3_kernel.  For s >= 1 an unlisted neighbour value cannot win at all.
// Pixel (y, x) needs (y-1, x) and (y, x-1): all pixels of an anti-diagonal x + y = k are
// independent.  Bands of 64 rows, one wave each on its own CU, walk the diagonals
// (ws_smooth_left_bands_kernel): the upper neighbour's value arrives from the lane above one step
// earlier, the left one is the lane's own previous result, and the window distance of a neighbour's
// value is a SLIDING sum (the cost at x-1 plus one window column entering, one leaving -- or the cost
// at y-1 plus one row entering, one leaving), so a step costs O(bs) pixel operations, not O(bs^2).
// No iteration, any image size.
// ------------------------------------------------------------------------------------------
struct SmoothLeftArgs {
    const uint8_t *L;
    const uint8_t *R;
    // packed planes of the marching kernel (null: compute the distances from the bytes)
    const uint32_t *A;
    const uint32_t *B;
    int pitch_a, pad_a, pitch_b, pad_b, centred;
    int w1, h1, s1, w2, h2, s2;
    int block_size, max_d, ssd;
    double s;
    float *out; // holds d1 on entry, the final map on exit
    int out_pitch;
    int spin_limit;         // polls a band makes for a hand-off word before it gives up (kBandSpinLimit)
    unsigned int *gave_up;  // host-visible word of the context: set when a band gave up (the map is then not valid)
};

__device__ __forceinline__ bool left_candidate_ok(const SmoothLeftArgs &g, int x, int d, int half)
{
    return d >= 1 && d <= g.max_d && x - d >= half && x - d < g.w2 - half;
}

template <bool CENTRED>
__device__ __forceinline__ long long left_ssd_planes(const SmoothLeftArgs &g, int x, int y, int d, int half)
{
    long long c = 0; // sum (a-b)^2 = sum a^2 + sum b^2 - 2 sum ab on the dword planes, row by row in 32 bits
    for (int r = 0; r < g.block_size; ++r) {
        const uint32_t *pa = g.A + (size_t)(y - half + r) * g.pitch_a + (x - half + g.pad_a);
        const uint32_t *pb = g.B + (size_t)(y - half + r) * g.pitch_b + (x - d - half + g.pad_b);
        uint32_t aa = 0, bb = 0, ab = 0;
        for (int i = 0; i < g.block_size; ++i) {
            const uint32_t a = pa[i], b = pb[i];
            aa = pix_dot<CENTRED>(a, a, aa);
            bb = pix_dot<CENTRED>(b, b, bb);
            ab = pix_dot<CENTRED>(a, b, ab);
        }
        c += (long long)(int32_t)aa + (int32_t)bb - 2LL * (int32_t)ab;
    }
    return c;
}

// One window column: pixels (xc, y-half .. y+half) of the left image against (xc - d, ..) of the
// right one.  The distance of a fixed d at consecutive x is a sliding sum of these.
__device__ __forceinline__ uint32_t left_col_cost(const SmoothLeftArgs &g, int xc, int y, int d, int half)
{
    if (g.A) {
        const uint32_t *pa = g.A + (size_t)(y - half) * g.pitch_a + (xc + g.pad_a);
        const uint32_t *pb = g.B + (size_t)(y - half) * g.pitch_b + (xc - d + g.pad_b);
        if (!g.ssd) {
            uint32_t acc = 0;
            for (int r = 0; r < g.block_size; ++r) acc = pix_sad(pa[(size_t)r * g.pitch_a], pb[(size_t)r * g.pitch_b], acc);
            return acc;
        }
        uint32_t aa = 0, bb = 0, ab = 0;
        if (g.centred) {
            for (int r = 0; r < g.block_size; ++r) {
                const uint32_t a = pa[(size_t)r * g.pitch_a], b = pb[(size_t)r * g.pitch_b];
                aa = pix_dot<true>(a, a, aa); bb = pix_dot<true>(b, b, bb); ab = pix_dot<true>(a, b, ab);
            }
        } else {
            for (int r = 0; r < g.block_size; ++r) {
                const uint32_t a = pa[(size_t)r * g.pitch_a], b = pb[(size_t)r * g.pitch_b];
                aa = pix_dot<false>(a, a, aa); bb = pix_dot<false>(b, b, bb); ab = pix_dot<false>(a, b, ab);
            }
        }
        return (uint32_t)((int32_t)aa + (int32_t)bb - 2 * (int32_t)ab);
    }
    const uint8_t *lw = g.L + (size_t)(y - half) * g.s1 + 3 * xc;
    const uint8_t *rw = g.R + (size_t)(y - half) * g.s2 + 3 * (xc - d);
    return window_cost(lw, g.s1, rw, g.s2, 1, g.block_size, g.ssd);
}

// the whole window, row by row
__device__ __forceinline__ uint32_t left_cost_int(const SmoothLeftArgs &g, int x, int y, int d, int half)
{
    if (g.A) {
        if (g.ssd) return (uint32_t)(g.centred ? left_ssd_planes<true>(g, x, y, d, half) : left_ssd_planes<false>(g, x, y, d, half));
        uint32_t acc = 0;
        for (int r = 0; r < g.block_size; ++r) {
            const uint32_t *pa = g.A + (size_t)(y - half + r) * g.pitch_a + (x - half + g.pad_a);
            const uint32_t *pb = g.B + (size_t)(y - half + r) * g.pitch_b + (x - d - half + g.pad_b);
            for (int i = 0; i < g.block_size; ++i) acc = pix_sad(pa[i], pb[i], acc);
        }
        return acc;
    }
    const uint8_t *lw = g.L + (size_t)(y - half) * g.s1 + 3 * (x - half);
    const uint8_t *rw = g.R + (size_t)(y - half) * g.s2 + 3 * (x - d - half);
    return window_cost(lw, g.s1, rw, g.s2, g.block_size, g.block_size, g.ssd);
}

__device__ __forceinline__ double left_dist_of(const SmoothLeftArgs &g, uint32_t c)
{
    return g.ssd ? sqrt((double)c) : (double)c; // cv::norm(NORM_L2) of the window (BlockSearch.cpp:66)
}

__device__ __forceinline__ double left_dist(const SmoothLeftArgs &g, int x, int y, int d, int half)
{
    return left_dist_of(g, left_cost_int(g, x, y, d, half));
}

// candidate (dist, d) beats (bd, bdist) in the reference's iteration (d descending, strict <)
__device__ __forceinline__ bool left_better(double dist, int d, double bdist, int bd)
{
    return dist < bdist || (dist == bdist && d > bd);
}

// the three best candidates of a pixel: (cost, d, distance) x 3, kTopNone = no such candidate.  The distance --
// cv::norm of the window, sqrt(cost) in double for SSD (BlockSearch.cpp:66) -- is taken here, by the data-parallel
// pre-pass, so that the raster pass does not spend its one wave's issue slots on three double square roots a step.
constexpr uint32_t kTopNone = 0xffffffffu; // above any window cost (63 * 63 * 3 * 255^2 < 2^30)
constexpr int kTopWords = 12;              // dwords per pixel: 3 x {cost, d | table bits, distance lo, distance hi}
// The d words: disparity in bits 0-19; bits 20-31 of the three together hold the pixel's 32-bit WINNER TABLE (12 + 12 + 8
// bits): for each of the 4 x 4 cases "the upper neighbour's value is entry 0 / 1 / 2 / none of them" x "the left one's is
// ...", which entry the reference's running minimum ends with (3: none) -- left_store_entry plays the 16 tournaments, with
// the reference's doubles, on every CU; the one wave per band looks the answer up.
constexpr int kTopTableShift = 20;
constexpr uint32_t kTopDMask = (1u << kTopTableShift) - 1u;
constexpr uint32_t kTopDNone = kTopDMask; // an empty entry's disparity field: no neighbour value equals it (d <= 65536)

__device__ __forceinline__ void left_store_entry(const SmoothLeftArgs &g, uint32_t *t, uint32_t c0, int d0, uint32_t c1, int d1,
                                                 uint32_t c2, int d2)
{
    const uint32_t c[3] = {c0, c1, c2};
    const int d[3] = {d0, d1, d2};
    double m[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) m[i] = c[i] == kTopNone ? 0.0 : left_dist_of(g, c[i]);
    uint32_t table = 0;
#pragma unroll
    for (int iu = 0; iu < 4; ++iu) {
#pragma unroll
        for (int il = 0; il < 4; ++il) {
            double dist = 1.7976931348623157e308; // (LeftBest, below: the running minimum of BlockSearch.cpp:50-79)
            int bd = -1, w = 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (c[i] == kTopNone) continue;
                double v = m[i];
                if (iu == i) v *= g.s; // the upper factor first (:68-70)
                if (il == i) v *= g.s; // the left factor second (:71-73)
                if (v < dist || (v == dist && bd >= 0 && d[i] > bd)) { dist = v; bd = d[i]; w = i; }
            }
            table |= (uint32_t)w << (2 * (iu * 4 + il));
        }
    }
    const uint32_t bits[3] = {table & 0xfffu, (table >> 12) & 0xfffu, table >> 24};
#pragma unroll
    for (int i = 0; i < 3; ++i)
        reinterpret_cast<uint4 *>(t)[i] = make_uint4(c[i], (c[i] == kTopNone ? kTopDNone : (uint32_t)d[i] & kTopDMask) | (bits[i] << kTopTableShift),
                                                     (uint32_t)__double2loint(m[i]), (uint32_t)__double2hiint(m[i]));
}

__global__ void __launch_bounds__(256) ws_left_top3_kernel(const SmoothLeftArgs g, uint32_t *__restrict__ top, int top_pitch)
{
    const int half = (g.block_size - 1) / 2;
    const int x = half + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = half + blockIdx.y;
    if (x >= g.w1 - half) return;
    uint32_t c0 = kTopNone, c1 = kTopNone, c2 = kTopNone;
    int d0 = 0, d1 = 0, d2 = 0;
    if (!black3(g.L + (size_t)y * g.s1 + 3 * x)) {
        const int d_hi = min(g.max_d, x - half);
        const int d_lo = max(1, x - (g.w2 - half) + 1);
        for (int d = d_hi; d >= d_lo; --d) {
            const uint32_t c = left_cost_int(g, x, y, d, half);
            if (c < c2) { // strict: an equal cost met later (smaller d) stays behind
                if (c < c1) {
                    c2 = c1; d2 = d1;
                    if (c < c0) { c1 = c0; d1 = d0; c0 = c; d0 = d; }
                    else { c1 = c; d1 = d; }
                } else { c2 = c; d2 = d; }
            }
        }
    }
    left_store_entry(g, top + ((size_t)y * top_pitch + x) * kTopWords, c0, d0, c1, d1, c2, d2);
}

// ---- the same three, separably (round 3) ----------------------------------------------------------------------
// ws_left_top3_kernel sums a whole window per (pixel, disparity): bs^2 pixel costs each, 13 ms at 17 x 17, D = 200,
// 900 x 750.  A window cost is a box sum of per-pixel costs, so: for a slab of kTopSlab disparities (descending, the
// reference's order) ws_left_top3_rows_kernel writes the bs-wide ROW sums of every (row, column, d) of the slab,
// ws_left_top3_cols_kernel adds bs of them down a column and feeds the pixel's running best three -- kept in the
// top-3 buffer between slabs, the distances taken after the last one.  2 bs instead of bs^2 pixel costs per
// candidate; the same integers, so the same three candidates in the same order.
constexpr int kTopSlab = 8;

// cost of the bs pixels (xl - half .. xl + half, row r) of the left image against the same run at xl - d of the right
__device__ __forceinline__ uint32_t left_row_cost(const SmoothLeftArgs &g, int xl, int r, int d, int half)
{
    if (g.A) {
        const uint32_t *pa = g.A + (size_t)r * g.pitch_a + (xl - half + g.pad_a);
        const uint32_t *pb = g.B + (size_t)r * g.pitch_b + (xl - d - half + g.pad_b);
        if (g.ssd) {
            uint32_t aa = 0, bb = 0, ab = 0;
            if (g.centred) {
                for (int i = 0; i < g.block_size; ++i) {
                    const uint32_t a = pa[i], b = pb[i];
                    aa = pix_dot<true>(a, a, aa); bb = pix_dot<true>(b, b, bb); ab = pix_dot<true>(a, b, ab);
                }
            } else {
                for (int i = 0; i < g.block_size; ++i) {
                    const uint32_t a = pa[i], b = pb[i];
                    aa = pix_dot<false>(a, a, aa); bb = pix_dot<false>(b, b, bb); ab = pix_dot<false>(a, b, ab);
                }
            }
            return (uint32_t)((int32_t)aa + (int32_t)bb - 2 * (int32_t)ab); // (>= 0: a sum of squares)
        }
        uint32_t acc = 0;
        for (int i = 0; i < g.block_size; ++i) acc = pix_sad(pa[i], pb[i], acc);
        return acc;
    }
    return window_cost(g.L + (size_t)r * g.s1 + 3 * (xl - half), g.s1, g.R + (size_t)r * g.s2 + 3 * (xl - d - half), g.s2,
                       g.block_size, 1, g.ssd);
}

// vol[(s * rows + r) * w1 + x] = row sum of (row r, column x, disparity d_top - s); grid (columns, rows, slab)
__global__ void __launch_bounds__(256) ws_left_top3_rows_kernel(const SmoothLeftArgs g, uint32_t *__restrict__ vol, int rows,
                                                                int d_top)
{
    const int half = (g.block_size - 1) / 2;
    const int x = half + blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y, d = d_top - (int)blockIdx.z;
    if (x >= g.w1 - half || !left_candidate_ok(g, x, d, half)) return; // (nobody reads the sums of a candidate that is none)
    vol[((size_t)blockIdx.z * rows + r) * g.w1 + x] = left_row_cost(g, x, r, d, half);
}

__global__ void __launch_bounds__(256) ws_left_top3_cols_kernel(const SmoothLeftArgs g, const uint32_t *__restrict__ vol, int rows,
                                                                int d_top, int nslab, int first, int last,
                                                                uint32_t *__restrict__ top, int top_pitch)
{
    const int half = (g.block_size - 1) / 2;
    const int x = half + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = half + blockIdx.y;
    if (x >= g.w1 - half) return;
    uint32_t *t = top + ((size_t)y * top_pitch + x) * kTopWords;
    uint32_t c0 = kTopNone, c1 = kTopNone, c2 = kTopNone;
    int d0 = 0, d1 = 0, d2 = 0;
    if (!first) {
        c0 = t[0]; d0 = (int)t[1]; c1 = t[4]; d1 = (int)t[5]; c2 = t[8]; d2 = (int)t[9];
    }
    if (!black3(g.L + (size_t)y * g.s1 + 3 * x)) {
        const int d_hi = min(g.max_d, x - half);
        const int d_lo = max(1, x - (g.w2 - half) + 1);
        for (int sl = 0; sl < nslab; ++sl) {
            const int d = d_top - sl; // descending: the reference's order
            if (d > d_hi) continue;
            if (d < d_lo) break;
            const uint32_t *col = vol + ((size_t)sl * rows + (y - half)) * g.w1 + x;
            uint32_t c = 0;
            for (int k = 0; k < g.block_size; ++k) c += col[(size_t)k * g.w1];
            if (c < c2) { // strict: an equal cost met later (smaller d) stays behind
                if (c < c1) {
                    c2 = c1; d2 = d1;
                    if (c < c0) { c1 = c0; d1 = d0; c0 = c; d0 = d; }
                    else { c1 = c; d1 = d; }
                } else { c2 = c; d2 = d; }
            }
        }
    }
    if (last) {
        left_store_entry(g, t, c0, d0, c1, d1, c2, d2);
    } else {
        t[0] = c0; t[1] = (uint32_t)d0; t[4] = c1; t[5] = (uint32_t)d1; t[8] = c2; t[9] = (uint32_t)d2;
    }
}

// the reference's running minimum: 'min' starts at DBL_MAX with no winner, d descending, strict '<'
struct LeftBest {
    double dist;
    int d;
    __device__ __forceinline__ void consider(double m, int cd)
    {
        if (m < dist || (m == dist && d >= 0 && cd > d)) { dist = m; d = cd; }
    }
};

// 0 <= s <= 1 needs only the best candidate: d1 is in the map, its cost is summed here (all CUs),
// in the layout of the top-3 buffer
__global__ void __launch_bounds__(256) ws_left_cost_kernel(const SmoothLeftArgs g, uint32_t *__restrict__ top, int top_pitch)
{
    const int half = (g.block_size - 1) / 2;
    const int x = half + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = half + blockIdx.y;
    if (x >= g.w1 - half) return;
    uint32_t c[3] = {kTopNone, kTopNone, kTopNone};
    int dd[3] = {0, 0, 0};
    if (!black3(g.L + (size_t)y * g.s1 + 3 * x)) {
        const float *o = g.out + (size_t)y * g.out_pitch + x;
        const int d = (int)o[0];
        if (left_candidate_ok(g, x, d, half)) { // otherwise: no candidate at all, the value x stays
            dd[0] = d;
            c[0] = left_cost_int(g, x, y, d, half);
            // Two more TRUE candidates with their true costs: what the upper and the left neighbour
            // hold now.  Listing more candidates never changes the minimum; when a neighbour keeps
            // its value (most do) the raster pass finds its cost here instead of summing a window.
            int n = 1;
            const float nb[2] = {y >= 1 ? o[-(ptrdiff_t)g.out_pitch] : 0.0f, x >= 1 ? o[-1] : 0.0f};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = (int)nb[k];
                if ((float)e == nb[k] && e != d && (n == 1 || e != dd[1]) && left_candidate_ok(g, x, e, half)) {
                    dd[n] = e;
                    c[n] = left_cost_int(g, x, y, e, half);
                    ++n;
                }
            }
        }
    }
    left_store_entry(g, top + ((size_t)y * top_pitch + x) * kTopWords, c[0], dd[0], c[1], dd[1], c[2], dd[2]);
}

// Sliding window sums.  The cost of (x, y, d) follows from the cost of (x-1, y, d) -- one window
// column enters, one leaves -- or from the cost of (x, y-1, d) -- one window row enters, one leaves.
// The raster pass is one wave per band whose step time is the latency of these sums, so on the planes
// the loads of both lines are issued in batches of 8 pixels (clamped index, the tail masked) rather
// than one dependent load after the other.
template <int MODE> // 0 SAD, 1 SSD, 2 SSD on centred planes
__device__ __forceinline__ uint32_t left_two_lines(uint32_t c_prev, const uint32_t *a_in, const uint32_t *b_in,
                                                   const uint32_t *a_out, const uint32_t *b_out, int n, int sa, int sb)
{
    uint32_t in = 0, out = 0;
    for (int r0 = 0; r0 < n; r0 += 8) {
        uint32_t ai[8], bi[8], ao[8], bo[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = min(r0 + k, n - 1);
            ai[k] = a_in[r * sa]; bi[k] = b_in[r * sb];
            ao[k] = a_out[r * sa]; bo[k] = b_out[r * sb];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (r0 + k < n) {
                if constexpr (MODE == 0) {
                    in = pix_sad(ai[k], bi[k], in);
                    out = pix_sad(ao[k], bo[k], out);
                } else { // (a-b)^2 = a^2 + b^2 - 2ab per pixel, exact in 32 bits
                    in += pix_dot<MODE == 2>(ai[k], ai[k], 0u) + pix_dot<MODE == 2>(bi[k], bi[k], 0u) - 2u * pix_dot<MODE == 2>(ai[k], bi[k], 0u);
                    out += pix_dot<MODE == 2>(ao[k], ao[k], 0u) + pix_dot<MODE == 2>(bo[k], bo[k], 0u) - 2u * pix_dot<MODE == 2>(ao[k], bo[k], 0u);
                }
            }
        }
    }
    return c_prev + in - out;
}

__device__ __forceinline__ uint32_t left_two_lines(const SmoothLeftArgs &g, uint32_t c_prev, const uint32_t *a_in, const uint32_t *b_in,
                                                   const uint32_t *a_out, const uint32_t *b_out, int sa, int sb)
{
    if (!g.ssd) return left_two_lines<0>(c_prev, a_in, b_in, a_out, b_out, g.block_size, sa, sb);
    return g.centred ? left_two_lines<2>(c_prev, a_in, b_in, a_out, b_out, g.block_size, sa, sb)
                     : left_two_lines<1>(c_prev, a_in, b_in, a_out, b_out, g.block_size, sa, sb);
}

// cost of (x, y, d) from the cost of (x - 1, y, d)
__device__ __forceinline__ uint32_t left_slide(const SmoothLeftArgs &g, uint32_t c_prev, int x, int y, int d, int half)
{
    if (g.A) {
        const uint32_t *pa = g.A + (size_t)(y - half) * g.pitch_a + g.pad_a;
        const uint32_t *pb = g.B + (size_t)(y - half) * g.pitch_b + (g.pad_b - d);
        return left_two_lines(g, c_prev, pa + (x + half), pb + (x + half), pa + (x - 1 - half), pb + (x - 1 - half), g.pitch_a, g.pitch_b);
    }
    return c_prev + left_col_cost(g, x + half, y, d, half) - left_col_cost(g, x - 1 - half, y, d, half);
}

// cost of (x, y, d) from the cost of (x, y - 1, d)
__device__ __forceinline__ uint32_t left_slide_down(const SmoothLeftArgs &g, uint32_t c_above, int x, int y, int d, int half)
{
    if (g.A) {
        const uint32_t *pa = g.A + (x - half + g.pad_a), *pb = g.B + (x - half - d + g.pad_b);
        return left_two_lines(g, c_above, pa + (size_t)(y + half) * g.pitch_a, pb + (size_t)(y + half) * g.pitch_b,
                              pa + (size_t)(y - 1 - half) * g.pitch_a, pb + (size_t)(y - 1 - half) * g.pitch_b, 1, 1);
    }
    const uint8_t *l0 = g.L + 3 * (x - half), *r0 = g.R + 3 * (x - d - half);
    return c_above + window_cost(l0 + (size_t)(y + half) * g.s1, g.s1, r0 + (size_t)(y + half) * g.s2, g.s2, g.block_size, 1, g.ssd) -
           window_cost(l0 + (size_t)(y - 1 - half) * g.s1, g.s1, r0 + (size_t)(y - 1 - half) * g.s2, g.s2, g.block_size, 1, g.ssd);
}

// ---- the two planes' moving windows in LDS -------------------------------------------------
// A step of the raster pass is a chain of dependent loads (which neighbour value, then that value's window
// line) with one wave per CU: its length is load latency.  So each band keeps the part of both dword planes
// its diagonal can touch in LDS: rows [band's first row - half - 1, last row + half], and a circular window of
// columns -- plane A the 2 half + 66 + kBandFill columns up to the diagonal's head, plane B max_d more to the left.
// Column-major (a column's rows are contiguous, the lanes of a diagonal -- one row down, one column left each
// -- fall on different banks because rp - 1 is odd).  Every step one new column per plane is requested kBandFill
// steps ahead by LDS-DMA (global_load_lds_dword: lane = row, strided source, contiguous destination) and a
// column that leaves the window is overwritten cw columns later.
struct LeftLds {
    uint32_t *A, *B;  // element (plane column c, plane row r) at [phys(c) * rp + (r - row0)]
    int rp, cwa, cwb; // dwords per column, columns per window
    int row0;         // plane row of LDS row 0
    int wa, wb;       // a multiple of cwa / cwb not above the window's lowest column (so c - w < 2 cw)
    __device__ __forceinline__ int pa(int c) const { const int p = c - wa; return p >= cwa ? p - cwa : p; }
    __device__ __forceinline__ int pb(int c) const { const int p = c - wb; return p >= cwb ? p - cwb : p; }
    // dword offset of a column (24-bit multiply: full rate, and everything here is far below 2^24)
    __device__ __forceinline__ int oa(int c) const { return __mul24(pa(c), rp); }
    __device__ __forceinline__ int ob(int c) const { return __mul24(pb(c), rp); }
};

template <int MODE> // 0 SAD, 1 SSD, 2 SSD on centred planes
__device__ __forceinline__ uint32_t left_pix_cost(uint32_t a, uint32_t b)
{
    if constexpr (MODE == 0) return pix_sad(a, b, 0u);
    else return pix_dot<MODE == 2>(a, a, 0u) + pix_dot<MODE == 2>(b, b, 0u) - 2u * pix_dot<MODE == 2>(a, b, 0u); // (a-b)^2, exact in 32 bits
}

// n rows of column ca of plane A against column cb of plane B, from LDS row r on
template <int MODE>
__device__ __forceinline__ uint32_t lds_col_cost(const LeftLds &w, int ca, int cb, int r, int n)
{
    const uint32_t *pa = w.A + w.oa(ca) + r, *pb = w.B + w.ob(cb) + r;
    uint32_t acc = 0;
#pragma unroll 4
    for (int i = 0; i < n; ++i) acc += left_pix_cost<MODE>(pa[i], pb[i]);
    return acc;
}

// n columns from ca0 / cb0 on, LDS row r
template <int MODE>
__device__ __forceinline__ uint32_t lds_row_cost(const LeftLds &w, int ca0, int cb0, int r, int n)
{
    uint32_t acc = 0;
#pragma unroll 4
    for (int i = 0; i < n; ++i) acc += left_pix_cost<MODE>(w.A[w.oa(ca0 + i) + r], w.B[w.ob(cb0 + i) + r]);
    return acc;
}

// one plane column into its LDS slot: lane = LDS row (two instructions for up to 128 rows), rows clamped to the plane
__device__ __forceinline__ void lds_fill_column(const uint32_t *plane, int pitch, int plane_rows, int col, uint32_t *slot, int row0,
                                                int rows, int lane)
{
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const int c = min(max(col, 0), pitch - 1);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int r = part * 64 + lane;
        if (r < rows) {
            const uint32_t *src = plane + (size_t)min(max(row0 + r, 0), plane_rows - 1) * pitch + c;
            const uint32_t dst = (uint32_t)(uintptr_t)(lds_u32 *)(slot + part * 64);
            uint32_t saved_m0; // (M0 put back inside the statement: see stage_row_async in ws_march.hip)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(saved_m0) : "v"(src), "s"(dst) : "memory");
        }
    }
}

// The raster pass.  Rows are cut into bands of 64, one single-wave workgroup (on its own CU) per band, lane = row.
// Lane t works on column k - t at step k, so the lanes of a wave sit on an anti-diagonal; the upper neighbour's
// value and window cost are lane t-1's results of the previous step (one cross-lane move each, no LDS, no
// barrier), the left ones are the lane's own.  Between bands the last row of band b hands every finished pixel to
// the first row of band b+1 as ONE naturally aligned 8-byte word {value, cost} written with a device-scope store
// and polled with device-scope loads (the word itself is the flag: it starts as all ones); band b+1 asks for a
// column's word one step before it needs it, so in steady state -- every band runs the same program at the same
// pace, 64 steps and a hand-off latency behind its predecessor -- the hand-off is off the critical path.
// Band numbers are tickets drawn at start-up: a band's predecessor has then certainly started, whatever order
// the workgroups are dispatched in.  A poll that never succeeds (it cannot, short of a bug) gives up after a
// bounded number of tries and flags it instead of hanging the device: in the scratch (ctrl[1]) and in a host-visible
// word of the context that every synchronising boundary call checks (ws_capi.cpp: check_device_status; ws_device_status
// for callers of ws_search_device).
// Round 4: TWO lanes per pixel.  A third of a step's instructions were the two sliding sums -- the upper neighbour's
// value slid down a row, the left one's slid along the row: 8 BS LDS reads, their circular-window addresses, 2 x 6 BS dot
// products -- in ONE lane.  Now a band is 32 rows, lane 2r takes row r's DOWN slide and lane 2r + 1 its RIGHT slide, as one
// instruction stream over (base, stride, wrap) of "its" lines; the two sums are swapped by one DPP move each and both
// lanes replay the same decision.  A step has ~100 instructions fewer (of ~700 at 7 x 7), the chain of steps is 5 %
// longer (more, shorter bands): 3.22 -> 3.12 ms at 7 x 7, 5.05 -> 4.74 ms at 17 x 17 (profiles/r04/left_smooth.txt) --
// a step's time is mostly NOT its instruction count (see there for what else was tried).
#ifdef WS_BAND_STAMPS
// Development build only (tools/band_stamps.py): s_memtime at five points of a step, steps [kStampStep0, +kStampSteps) of
// every band; read back through ws_debug_band_stamps.  Each stamp waits for the scalar unit's counter (lgkmcnt(0)), so a
// stamped step is a little longer than a plain one.
constexpr int kStampBands = 32, kStampStep0 = 300, kStampSteps = 128, kStampPoints = 8; // 0-4: the step's segments; 5-7: inside the decision
__device__ unsigned long long ws_band_stamps[kStampBands][kStampSteps][kStampPoints];
#define WS_STAMP(p)                                                                                        \
    do {                                                                                                   \
        if (band < kStampBands && k >= kStampStep0 && k < kStampStep0 + kStampSteps) {                     \
            const unsigned long long now_ = __builtin_readcyclecounter();                                  \
            if (lane == 0) ws_band_stamps[band][k - kStampStep0][p] = now_;                                \
        }                                                                                                  \
    } while (0)
#else
#define WS_STAMP(p) do { } while (0)
#endif
constexpr int kBandLanes = 64;            // one wave per band
constexpr int kBandRows = kBandLanes / 2; // rows per band: two lanes per row
constexpr unsigned long long kEdgeNone = ~0ull;
constexpr int kBandSpinLimit = 1 << 20;

constexpr int kBandDepth = 3; // steps a pixel's inputs (candidate list, map value, hand-off word) are requested ahead
#ifndef WS_BAND_LAG
#define WS_BAND_LAG 3 // (development builds: tools/variants.py; 5 -> 3: -2.5 % at 7 x 7, profiles/r04/left_smooth.txt)
#endif
constexpr int kBandLag = WS_BAND_LAG; // columns a band stays behind what its requests need from the band above
constexpr int kBandFill = 5;  // steps a window column is requested ahead of its first use

// BS > 0 (with MODE >= 0): the block size at compile time.  The lines of the two common sliding sums -- the upper
// neighbour's value slid down from the row above, the left neighbour's value slid along the row -- are then read
// from LDS at the TOP of the decision, for every lane, whether or not it will need them (8 BS registers): one wave per
// CU has nothing else to hide an LDS round trip behind, and a step used to make fifteen to twenty of them in a row.
// TW (with BS > 0): a second, ROW-major copy of both windows (+ BS mirrored columns behind the circular ones), so that the
// down slide's window rows are consecutive dwords like the right slide's window columns are in the column-major copy:
// every line of the two sums is then read at compile-time offsets from ONE address -- no stride, no wrap-around test per
// element (those were two thirds of the sums' instructions; profiles/r04/left_smooth.txt, 6.).
template <int MODE, int BS = 0, bool TW = false> // MODE -1: window lines from global memory; 0 SAD / 1 SSD / 2 SSD centred: from the LDS windows
__global__ void __launch_bounds__(kBandLanes) ws_smooth_left_bands_kernel(const SmoothLeftArgs g, const uint32_t *__restrict__ top,
                                                                         int top_pitch, unsigned long long *edge, int edge_pitch,
                                                                         unsigned int *ctrl, int cwa, int cwb)
{
    extern __shared__ uint4 ws_smem4[];
    const int lane = threadIdx.x;
    const int t = lane >> 1;    // the lane's row of the band (two lanes per row)
    const int role = lane & 1;  // 0: slides the upper neighbour's value down, stores the results; 1: slides the left one's along
    unsigned int ticket = 0;
    if (lane == 0) ticket = atomicAdd(&ctrl[0], 1u);
    const int band = (int)__builtin_amdgcn_readfirstlane(ticket);
    const int bsz = BS > 0 ? BS : g.block_size;
    const int half = (bsz - 1) / 2;
    const int height = min(g.h1, g.h2);
    const int iw = g.w1 - 2 * half, ih = height - 2 * half; // interior
    const bool other_can_win = !(g.s >= 1.0); // (see above: for s >= 1 an unlisted neighbour value never wins)
    const bool slide_ok = bsz > 2;   // (a 1-pixel window is cheaper summed than slid)
    const int y = half + band * kBandRows + t;
    const bool row_ok = y < height - half;
    const int nrows = min(kBandRows, ih - band * kBandRows);
    if (nrows <= 0) return; // (uniform; the grid has exactly ceil(ih / 64) workgroups)
    const int nsteps = iw + nrows - 1;
    const unsigned long long *edge_in = edge + (size_t)band * edge_pitch; // written by the band above
    unsigned long long *edge_out = edge + (size_t)(band + 1) * edge_pitch;
    const bool hands_down = t == nrows - 1 && role == 0 && (band + 1) * kBandRows < ih;
    float *orow = g.out + (size_t)min(y, height - 1) * g.out_pitch;
    const uint32_t *trow = top + (size_t)min(y, height - 1) * top_pitch * kTopWords;
    // the lane's running state along its row
    float lv = 0.0f;          // value of (y, x-1); the ring column left of the interior holds 0
    uint32_t lcost = 0;       // window cost of (x-1, lv) ...
    bool lknown = false;      // ... if it is known
    int uv = 0, ux = -2;      // last upper-neighbour value whose cost was needed, at column ux
    uint32_t ucost = 0;
    float vprev = 0.0f;          // what this lane produced in the previous step (for the lane below)
    uint32_t cprev = kTopNone;   // ... and that value's window cost there (kTopNone: not known)
    bool gave_up = false;

    // the LDS windows (MODE >= 0): image columns [0, 2 half + kBandFill) before the first step, then one per step
    LeftLds win{};
    const int lds_rows = kBandRows + 2 * half + 1;
    if constexpr (MODE >= 0) {
        win.rp = lds_rows + 1; // even
        win.cwa = cwa; win.cwb = cwb;
        win.A = reinterpret_cast<uint32_t *>(ws_smem4);
        win.B = win.A + (size_t)cwa * win.rp;
        win.row0 = band * kBandRows - 1; // = (the band's first row) - half - 1
        win.wa = (g.pad_a / cwa) * cwa; win.wb = (g.pad_b / cwb) * cwb;
        if constexpr (!TW) { // (TW: both copies are written from registers, below)
            for (int c = 0; c < 2 * half + kBandFill; ++c) {
                lds_fill_column(g.A, g.pitch_a, g.h1, c + g.pad_a, win.A + win.pa(c + g.pad_a) * win.rp, win.row0, lds_rows, lane);
                lds_fill_column(g.B, g.pitch_b, g.h2, c + g.pad_b, win.B + win.pb(c + g.pad_b) * win.rp, win.row0, lds_rows, lane);
            }
        }
    }
    // the row-major copies (TW): element (plane column c, LDS row r) at [r * tw + phys(c)], and again at [.. + cw] for
    // phys(c) < BS.  With them a column comes from memory by an ordinary load (lane = LDS row) kBandFill steps before
    // its first use and goes to BOTH copies at the top of the next step (no LDS-DMA: the value is in a register anyway,
    // and a store is cheaper than the M0 dance).
    static_assert(!TW || (BS > 0 && MODE >= 0), "the row-major copy serves the compile-time sliding sums");
    constexpr int TB = TW ? BS : 0;
    const int twa = (cwa + TB + 1) & ~1, twb = (cwb + TB + 1) & ~1; // even: lanes one row down, one column left -> odd dword distance
    uint32_t *const tA = reinterpret_cast<uint32_t *>(ws_smem4) + (size_t)(cwa + cwb) * (lds_rows + 1);
    uint32_t *const tB = tA + (size_t)twa * lds_rows;
    const int t_lane = min(lane, lds_rows - 1); // (lanes past the last LDS row repeat it: same value, same place)
    const uint32_t *const t_row_a = g.A + (size_t)min(max(win.row0 + t_lane, 0), g.h1 - 1) * g.pitch_a; // (clamped like lds_fill_column)
    const uint32_t *const t_row_b = g.B + (size_t)min(max(win.row0 + t_lane, 0), g.h2 - 1) * g.pitch_b;
    uint32_t *const t_lane_a = tA + t_lane * twa, *const t_lane_b = tB + t_lane * twb;
    auto t_load = [&](const uint32_t *row, int pitch, int col) -> uint32_t { return row[min(max(col, 0), pitch - 1)]; };
    uint32_t *const c_lane_a = win.A + t_lane, *const c_lane_b = win.B + t_lane; // (the column-major copies: [phys * rp + row])
    auto t_store = [&](uint32_t *t, uint32_t *cm, int cw, int phys, uint32_t v) {
        cm[__mul24(phys, win.rp)] = v;
        t[phys] = v;
        if (phys < TB) t[phys + cw] = v; // (uniform)
    };
    // (per-lane constants of the sliding sums' addresses; r_in / r_out / r_col: the LDS rows of the window row that enters,
    // the one that leaves, the window's first)
    const int tw_r_in = y + half - win.row0, tw_r_out = y - 1 - half - win.row0, tw_r_col = y - half - win.row0;
    const int tw_ta = (int)(tA - win.A), tw_tb = (int)(tB - win.A), tw_cb = (int)(win.B - win.A);
    const int tw_col_in = role ? half : -half, tw_col_out = role ? -1 - half : -half;
    const int tw_mul = role ? win.rp : 1;
    const int tw_a_in = role ? tw_r_col : tw_ta + tw_r_in * twa, tw_a_out = role ? tw_r_col : tw_ta + tw_r_out * twa;
    const int tw_b_in = role ? tw_cb + tw_r_col : tw_tb + tw_r_in * twb, tw_b_out = role ? tw_cb + tw_r_col : tw_tb + tw_r_out * twb;
    uint32_t tva = 0, tvb = 0; // the column requested in the previous step ...
    int tcol = -1;             // ... (image column; -1: none)
    if constexpr (TW) {
        for (int c0 = 0; c0 < 2 * half + kBandFill; c0 += 4) { // (four columns' loads in flight)
            uint32_t va[4], vb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                va[i] = t_load(t_row_a, g.pitch_a, c0 + i + g.pad_a);
                vb[i] = t_load(t_row_b, g.pitch_b, c0 + i + g.pad_b);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (c0 + i >= 2 * half + kBandFill) break; // (uniform)
                t_store(t_lane_a, c_lane_a, cwa, win.pa(c0 + i + g.pad_a), va[i]);
                t_store(t_lane_b, c_lane_b, cwb, win.pb(c0 + i + g.pad_b), vb[i]);
            }
        }
    }
    const int slack = cwa - (2 * half + kBandRows + 2 + kBandFill); // columns the windows keep behind the last one any lane reads
    // (the windows' first columns are on their way or in place before the band waits for its turn: they are the caller's
    // images, not the band above's results)
    // A band starts once the band above is kBandLag columns into its last row: from then on both run the same
    // program at the same pace, and a hand-off word asked for kBandDepth steps early has been written by then.
    if (band > 0) {
        const int xl = half + min(kBandLag, iw - 1);
        unsigned long long w = kEdgeNone;
        for (int spins = 0;; ++spins) {
            if (lane == 0) w = __hip_atomic_load(&edge_in[xl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__builtin_amdgcn_ballot_w64(lane == 0 && w == kEdgeNone)) break;
            if (spins > g.spin_limit) {
                gave_up = true;
                if (lane == 0) {
                    atomicOr(&ctrl[1], 1u);
                    __hip_atomic_store(g.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // the host checks it at its next synchronisation
                }
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }

    // window cost of candidate d at (x, y) from the cost at (x-1, y), from the cost at (x, y-1), or summed whole
    auto cost_slide = [&](uint32_t c_prev, int x, int d) -> uint32_t {
        if constexpr (MODE >= 0) {
            const int r = y - half - win.row0;
            return c_prev + lds_col_cost<MODE>(win, x + half + g.pad_a, x + half - d + g.pad_b, r, bsz) -
                   lds_col_cost<MODE>(win, x - 1 - half + g.pad_a, x - 1 - half - d + g.pad_b, r, bsz);
        } else {
            return left_slide(g, c_prev, x, y, d, half);
        }
    };
    auto cost_slide_down = [&](uint32_t c_up, int x, int d) -> uint32_t {
        if constexpr (MODE >= 0) {
            const int ca = x - half + g.pad_a, cb = x - half - d + g.pad_b;
            return c_up + lds_row_cost<MODE>(win, ca, cb, y + half - win.row0, bsz) -
                   lds_row_cost<MODE>(win, ca, cb, y - 1 - half - win.row0, bsz);
        } else {
            return left_slide_down(g, c_up, x, y, d, half);
        }
    };
    auto cost_full = [&](int x, int d) -> uint32_t {
        if constexpr (MODE >= 0) {
            uint32_t acc = 0;
            for (int i = 0; i < bsz; ++i)
                acc += lds_col_cost<MODE>(win, x - half + i + g.pad_a, x - half + i - d + g.pad_b, y - half - win.row0, bsz);
            return acc;
        } else {
            return left_cost_int(g, x, y, d, half);
        }
    };

    // A step's result leaves at the top of the NEXT step (the map pixel, and for the band's last row the hand-off
    // word: ONE 8-byte device-scope store the band below polls).
    int store_x = -1;
    auto flush_result = [&]() {
        if (store_x >= 0) {
            if (role == 0) orow[store_x] = vprev;
            if (hands_down)
                __hip_atomic_store(&edge_out[store_x], ((unsigned long long)__float_as_uint(vprev) << 32) | cprev, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            store_x = -1;
        }
    };

    // What a pixel starts from -- its candidate list, the map's value there (black / no candidate: fixed) and, for
    // the band's first row, the hand-off word of the pixel above -- is requested kBandDepth steps before its
    // step and queues up in registers: the step never waits for a load it issued itself.  (The window columns
    // are requested BEFORE these loads in every step and waits retire in order, so a column asked for
    // kBandFill > kBandDepth steps ahead has landed once the loads of the same step have been consumed.)
    // The queue is kBandDepth register slots used round-robin by a loop unrolled kBandDepth times (no moves: a move
    // would have to wait for the load it moves).
    struct Slot { uint4 a0, a1, a2; float ao; unsigned long long aw; };
    Slot slot[kBandDepth];
#pragma unroll
    for (int i = 0; i < kBandDepth; ++i) {
        slot[i].a0 = slot[i].a1 = slot[i].a2 = make_uint4(kTopNone, 0u, 0u, 0u);
        slot[i].ao = 0.0f;
        slot[i].aw = kEdgeNone;
    }
    auto step = [&](const int k, Slot &sl) {
        if (k >= nsteps) return; // (uniform)
        const int xs = k - t;
        const bool in = k >= 0 && row_ok && xs >= 0 && xs < iw;
        const int x = half + xs;
        WS_STAMP(0);
        flush_result();
        if constexpr (MODE >= 0) {
            if (k >= 0) {
                // the windows' lowest columns are k - kBandRows - slack and that - max_d (image columns)
                const int lo_a = max(k - kBandRows - slack, 0) + g.pad_a, lo_b = max(k - kBandRows - slack - g.max_d, 0) + g.pad_b;
                if (lo_a >= win.wa + win.cwa) win.wa += win.cwa;
                if (lo_b >= win.wb + win.cwb) win.wb += win.cwb;
                const int cn = k + 2 * half + kBandFill; // first read at step k + kBandFill
                if constexpr (!TW) {
                    lds_fill_column(g.A, g.pitch_a, g.h1, cn + g.pad_a, win.A + win.pa(cn + g.pad_a) * win.rp, win.row0, lds_rows, lane);
                    lds_fill_column(g.B, g.pitch_b, g.h2, cn + g.pad_b, win.B + win.pb(cn + g.pad_b) * win.rp, win.row0, lds_rows, lane);
                }
                if constexpr (TW) {
                    if (tcol >= 0) {
                        t_store(t_lane_a, c_lane_a, cwa, win.pa(tcol + g.pad_a), tva);
                        t_store(t_lane_b, c_lane_b, cwb, win.pb(tcol + g.pad_b), tvb);
                    }
                    tva = t_load(t_row_a, g.pitch_a, cn + g.pad_a);
                    tvb = t_load(t_row_b, g.pitch_b, cn + g.pad_b);
                    tcol = cn;
                }
            }
        }
        // this step's inputs leave their slot, the inputs of step k + kBandDepth take it
        const uint4 e0 = sl.a0, e1 = sl.a1, e2 = sl.a2;
        const float omap = sl.ao;
        unsigned long long w = sl.aw;
        {
            const int xn = xs + kBandDepth; // (clamped, not branched: every step issues the same loads)
            const bool on = row_ok && xn >= 0 && xn < iw;
            const int xc = half + min(max(xn, 0), iw - 1);
            const uint32_t *q = trow + (size_t)xc * kTopWords;
            sl.a0 = reinterpret_cast<const uint4 *>(q)[0];
            if (!on) sl.a0.x = kTopNone;
            sl.a1 = reinterpret_cast<const uint4 *>(q)[1];
            sl.a2 = reinterpret_cast<const uint4 *>(q)[2];
            sl.ao = orow[xc];
            sl.aw = kEdgeNone;
            if (band > 0 && t == 0) sl.aw = __hip_atomic_load(&edge_in[xc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (k < 0) return; // (uniform)
        WS_STAMP(1);
        // the upper neighbour (y-1, x): the lane above finished it in the previous step
        // (wave_shr:1 -- one DPP move each; __shfl_up goes through the LDS crossbar and its latency)
        // (the row above is two lanes up: both of its lanes hold the same result)
        const int v1 = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(vprev), 0x138, 0xf, 0xf, false);
        const int c1 = __builtin_amdgcn_update_dpp(0, (int)cprev, 0x138, 0xf, 0xf, false);
        float upf = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, v1, 0x138, 0xf, 0xf, false));
        uint32_t c_above = (uint32_t)__builtin_amdgcn_update_dpp(0, c1, 0x138, 0xf, 0xf, false);
        if (t == 0) { upf = 0.0f; c_above = kTopNone; } // band 0: the ring row above the interior holds 0
        if (band > 0) { // (uniform) lane 0's upper neighbour belongs to the band above
            for (int spins = 0; !gave_up; ++spins) { // (only if the band above fell behind: it was ahead at start-up)
                const bool wait = t == 0 && in && w == kEdgeNone;
                if (!__builtin_amdgcn_ballot_w64(wait)) break;
                if (spins > g.spin_limit) {
                    gave_up = true;
                    if (lane == 0) {
                        atomicOr(&ctrl[1], 1u);
                        __hip_atomic_store(g.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if (wait) w = __hip_atomic_load(&edge_in[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (t == 0 && in) {
                upf = __uint_as_float((uint32_t)(w >> 32));
                c_above = (uint32_t)w;
            }
        }
        // (BS > 0) the lines of the two common sliding sums, read and summed NOW by both lanes of the pixel, whether or not
        // the decision below will ask for them: lane 0 of the pixel slides the upper neighbour's value down (window rows
        // y + half in / y - 1 - half out: one element per window COLUMN, a stride of rp dwords that wraps around the
        // circular window), lane 1 slides the left neighbour's value along the row (window columns x + half in /
        // x - 1 - half out: consecutive dwords of one column).  One instruction stream: element i of a line sits at
        // base + i * stride, minus the window's span once it runs past the window's end.
        constexpr int NB = BS > 0 ? BS : 1;
        WS_STAMP(2);
        uint32_t pre_down = 0, pre_right = 0; // c_above slid down to (x, y) / lcost slid right to (x, y): valid if those were
        if (MODE >= 0 && BS > 0 && other_can_win) { // (uniform)
            const int xq = half + min(max(xs, 1), iw - 1);                  // (lanes outside their row read somewhere harmless)
            const int du = min(max((int)upf, 1), g.max_d), dl = min(max((int)lv, 1), g.max_d);
            const int r_in = y + half - win.row0, r_out = y - 1 - half - win.row0, r_col = y - half - win.row0;
            const int span_a = win.cwa * win.rp, span_b = win.cwb * win.rp;
            // first elements (dword offsets inside the windows) and strides
            int a_in, b_in, a_out, b_out, stride;
            uint32_t va_in[NB], vb_in[NB], va_out[NB], vb_out[NB];
            if constexpr (TW) {
                // One expression for both roles (a branch per role runs both sides, one after the other): the column a line
                // starts in differs by a per-lane constant, its physical column is scaled by rp (column-major copy, right
                // slide) or by 1 (row-major copy, down slide), the rest of the offset -- which copy, which row -- never
                // changes (tw_* below, dword offsets from win.A: the copies lie one behind the other).
                const int d = role ? dl : du;
                const int ci = xq + tw_col_in, co = xq + tw_col_out;
                a_in = __mul24(win.pa(ci + g.pad_a), tw_mul) + tw_a_in;
                b_in = __mul24(win.pb(ci - d + g.pad_b), tw_mul) + tw_b_in;
                a_out = __mul24(win.pa(co + g.pad_a), tw_mul) + tw_a_out;
                b_out = __mul24(win.pb(co - d + g.pad_b), tw_mul) + tw_b_out;
                (void)stride; (void)span_a; (void)span_b; (void)r_in; (void)r_out; (void)r_col;
                const uint32_t *pai = win.A + a_in, *pbi = win.A + b_in, *pao = win.A + a_out, *pbo = win.A + b_out;
#pragma unroll
                for (int i = 0; i < BS; ++i) {
                    va_in[i] = pai[i]; vb_in[i] = pbi[i];
                    va_out[i] = pao[i]; vb_out[i] = pbo[i];
                }
            } else {
            if (role == 0) {
                a_in = win.oa(xq - half + g.pad_a) + r_in;  b_in = win.ob(xq - half - du + g.pad_b) + r_in;
                a_out = a_in + (r_out - r_in);              b_out = b_in + (r_out - r_in);
                stride = win.rp;
            } else {
                a_in = win.oa(xq + half + g.pad_a) + r_col;      b_in = win.ob(xq + half - dl + g.pad_b) + r_col;
                a_out = win.oa(xq - 1 - half + g.pad_a) + r_col; b_out = win.ob(xq - 1 - half - dl + g.pad_b) + r_col;
                stride = 1;
            }
            // (a column's dwords never wrap; a row's elements do when their column index passes the window's end: then the
            // offset is at least `span` -- columns are rp dwords apart and the rows add less than rp)
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                va_in[i] = win.A[a_in]; vb_in[i] = win.B[b_in];
                va_out[i] = win.A[a_out]; vb_out[i] = win.B[b_out];
                a_in += stride; b_in += stride; a_out += stride; b_out += stride;
                if (a_in >= span_a) a_in -= span_a;
                if (a_out >= span_a) a_out -= span_a;
                if (b_in >= span_b) b_in -= span_b;
                if (b_out >= span_b) b_out -= span_b;
            }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): ONE wait for all the lines instead of one per dword
            uint32_t acc = role == 0 ? c_above : lcost;
            if constexpr (TW) {
                // the lines' sums on accumulators of their own: sum (a-b)^2 = sum a.a + sum b.b - 2 sum a.b, every product one
                // accumulating v_dot4 (a fresh (a-b)^2 per pixel pair costs a v_mov of the 0 per product and two additions)
                constexpr int M = MODE >= 0 ? MODE : 0;
                uint32_t in0 = 0, in1 = 0, in2 = 0, out0 = 0, out1 = 0, out2 = 0;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if constexpr (M == 0) {
                        in0 = pix_sad(va_in[i], vb_in[i], in0);
                        out0 = pix_sad(va_out[i], vb_out[i], out0);
                    } else {
                        in0 = pix_dot<M == 2>(va_in[i], va_in[i], in0);
                        in1 = pix_dot<M == 2>(vb_in[i], vb_in[i], in1);
                        in2 = pix_dot<M == 2>(va_in[i], vb_in[i], in2);
                        out0 = pix_dot<M == 2>(va_out[i], va_out[i], out0);
                        out1 = pix_dot<M == 2>(vb_out[i], vb_out[i], out1);
                        out2 = pix_dot<M == 2>(va_out[i], vb_out[i], out2);
                    }
                }
                acc += (in0 + in1 - 2u * in2) - (out0 + out1 - 2u * out2);
            } else {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                acc += left_pix_cost<(MODE >= 0 ? MODE : 0)>(va_in[i], vb_in[i]) - left_pix_cost<(MODE >= 0 ? MODE : 0)>(va_out[i], vb_out[i]);
            }
            // the pixel's other lane has the other sum: quad_perm [1, 0, 3, 2]
            const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0xb1, 0xf, 0xf, false);
            pre_down = role == 0 ? acc : other;
            pre_right = role == 0 ? other : acc;
        }
        auto pre_slide_down = [&](uint32_t) -> uint32_t { return pre_down; };
        auto pre_slide = [&](uint32_t) -> uint32_t { return pre_right; };
        constexpr bool PRE = MODE >= 0 && BS > 0;
        WS_STAMP(3);
        if (in) {
            float v;
            if (e0.x == kTopNone) {
                v = omap; // black (0) or no candidate at all (x): fixed
                lknown = false;
            } else {
                const int up = (int)upf;
                const bool up_ok = y >= 1 && (float)up == upf && left_candidate_ok(g, x, up, half);
                const int l = (int)lv;
                const bool l_ok = x >= 1 && (float)l == lv && left_candidate_ok(g, x, l, half);
                LeftBest b{1.7976931348623157e308, -1};
                uint32_t bcost = 0; // integer window cost of the running winner
                // the listed candidates: which of them the two neighbours' values are, then the pre-pass's table
                const uint32_t table = (e0.y >> kTopTableShift) | ((e1.y >> kTopTableShift) << 12) | ((e2.y >> kTopTableShift) << 24);
                const int ld0 = (int)(e0.y & kTopDMask), ld1 = (int)(e1.y & kTopDMask), ld2 = (int)(e2.y & kTopDMask);
                // (index of the first entry a value equals, 3 if none, in plain arithmetic: ne_i = min(value ^ d_i, 1) is 0 on a
                // match; a chain of compares, mask logic and selects runs through the scalar registers and costs a lone wave
                // several times as much.  Empty entries hold kTopDNone, a value that may not take part kTopDNone - 1.)
                auto first_match = [&](bool ok, int val) -> int {
                    const uint32_t u = ok ? (uint32_t)val : kTopDNone - 1u;
                    const uint32_t n0 = min(u ^ (uint32_t)ld0, 1u), n1 = min(u ^ (uint32_t)ld1, 1u), n2 = min(u ^ (uint32_t)ld2, 1u);
                    return (int)(n0 * (1u + n1 * (1u + n2)));
                };
                const int iu = first_match(up_ok, up), il = first_match(l_ok, l);
                const int w = (int)((table >> (2 * (iu * 4 + il))) & 3u);
                bool up_listed = iu != 3, l_listed = il != 3;
                b.d = w == 0 ? ld0 : w == 1 ? ld1 : w == 2 ? ld2 : -1;
                bcost = w == 0 ? e0.x : w == 1 ? e1.x : e2.x;
                // (its distance only where an unlisted neighbour value has to be compared with it)
                if (other_can_win && ((up_ok && !up_listed) || (l_ok && !l_listed)) && w != 3) {
                    double m = w == 0 ? __hiloint2double((int)e0.w, (int)e0.z) : w == 1 ? __hiloint2double((int)e1.w, (int)e1.z) : __hiloint2double((int)e2.w, (int)e2.z);
                    if (iu == w) m *= g.s; // the upper factor first (:68-70)
                    if (il == w) m *= g.s; // the left factor second (:71-73)
                    b.dist = m;
                }
                WS_STAMP(5);
                // the listed candidates' costs also serve the sliding sums below
                uint32_t cu = 0;
                bool cu_known = false;
                if (up_ok && !up_listed && other_can_win) {
                    if (slide_ok && c_above != kTopNone) cu = PRE ? pre_slide_down(c_above) : cost_slide_down(c_above, x, up);
                    else if (slide_ok && ux == x - 1 && uv == up) cu = cost_slide(ucost, x, up);
                    else if (slide_ok && lknown && l == up && l_ok) cu = cost_slide(lcost, x, up);
                    else cu = cost_full(x, up);
                    cu_known = true;
                    double m = left_dist_of(g, cu) * g.s;
                    if (l_ok && l == up) { m *= g.s; l_listed = true; }
                    const int before = b.d;
                    b.consider(m, up);
                    if (b.d != before) bcost = cu;
                } else if (up_ok && !up_listed && l_ok && l == up) {
                    l_listed = true; // (s >= 1: neither can win)
                }
                WS_STAMP(6);
                if (l_ok && !l_listed && other_can_win) {
                    const uint32_t cl = slide_ok && lknown ? (PRE ? pre_slide(lcost) : cost_slide(lcost, x, l)) : cost_full(x, l);
                    const int before = b.d;
                    b.consider(left_dist_of(g, cl) * g.s, l);
                    if (b.d != before) bcost = cl;
                }
                WS_STAMP(7);
                // remember the upper value's cost for the next column
                if (up_ok) {
                    if (cu_known) { uv = up; ucost = cu; ux = x; }
                    else if (iu != 3) { uv = up; ucost = iu == 0 ? e0.x : iu == 1 ? e1.x : e2.x; ux = x; } // (listed: its cost is in the list)
                }
                if (b.d >= 0) {
                    v = (float)b.d;
                    lcost = bcost;
                    lknown = true;
                } else {
                    v = (float)x; // nothing below DBL_MAX: minimumCorrespondX stays 0
                    lknown = false;
                }
            }
            lv = v;
            vprev = v;
            cprev = lknown ? lcost : kTopNone;
            store_x = x; // (written at the top of the next step)
        }
        WS_STAMP(4);
    };
    static_assert(kBandDepth == 3, "the loop below is unrolled by hand");
    for (int k = -kBandDepth; k < nsteps; k += kBandDepth) {
        step(k, slot[0]);
        step(k + 1, slot[1]);
        step(k + 2, slot[2]);
    }
    flush_result();
}

// scratch of the raster pass behind the per-pixel candidate lists: 64 control bytes + one row of hand-off words per band
static size_t smooth_left_edge_pitch(int w) { return (size_t)((w + 15) & ~15); }
static size_t smooth_left_sync_bytes(int w, int h) { return 64 + ((size_t)h / kBandRows + 2) * smooth_left_edge_pitch(w) * 8; }

// (+ the row sums of one slab of disparities, for the factors outside [0, 1])
static size_t smooth_left_vol_bytes(int w, int h) { return (size_t)kTopSlab * w * h * sizeof(uint32_t); }
// (the volume last, and only for the factors that run the separable top-3 pass: the pipeline's 0.9 does not, and the
// volume is two thirds more scratch -- 265 MB at 3840 x 2160)
size_t smooth_left_top_bytes(int w, int h, double s)
{
    const bool outside = !(s >= 0.0 && s <= 1.0);
    return (size_t)w * h * kTopWords * sizeof(uint32_t) + smooth_left_sync_bytes(w, h) + (outside ? smooth_left_vol_bytes(w, h) : 0);
}

hipError_t launch_smooth_left(const GenericArgs &g, double s, uint32_t *top3, const Canon *canon, Plane pa, Plane pb,
                              unsigned int *gave_up, hipStream_t st)
{
    SmoothLeftArgs a{};
    static const int spin_limit = [] {
        const char *e = getenv("WS_BAND_SPIN_LIMIT"); // development knob: -1 makes every band below the first give up (tests)
        return e ? atoi(e) : kBandSpinLimit;
    }();
    a.spin_limit = spin_limit;
    a.gave_up = gave_up;
    if (!gave_up) return hipErrorInvalidValue;
    if (canon) { // the marching kernel ran: its planes are the two images, one dword per pixel
        a.A = pa.data; a.B = pb.data;
        a.pitch_a = pa.pitch; a.pad_a = pa.pad; a.pitch_b = pb.pitch; a.pad_b = pb.pad;
        a.centred = march_centred(*canon);
    }
    a.L = g.L; a.R = g.R; a.w1 = g.w1; a.h1 = g.h1; a.s1 = g.s1; a.w2 = g.w2; a.h2 = g.h2; a.s2 = g.s2;
    a.block_size = g.block_size; a.max_d = g.max_d; a.ssd = g.ssd; a.s = s;
    a.out = g.out; a.out_pitch = g.out_pitch;
    if (!top3) return hipErrorInvalidValue;
    const int half = (g.block_size - 1) / 2;
    const int iw = g.w1 - 2 * half, ih = std::min(g.h1, g.h2) - 2 * half;
    if (iw <= 0 || ih <= 0) return hipSuccess;
    if (s >= 0.0 && s <= 1.0)
        hipLaunchKernelGGL(ws_left_cost_kernel, dim3(ceil_div(iw, 256), ih), dim3(256), 0, st, a, top3, g.w1);
    uint8_t *sync = reinterpret_cast<uint8_t *>(top3) + (size_t)g.w1 * g.h1 * kTopWords * sizeof(uint32_t);
    if (!(s >= 0.0 && s <= 1.0)) {
        static const bool whole_windows = [] {
            const char *e = getenv("WS_TOP3_WHOLE"); // development knob: 1 = the round-2 kernel (whole windows per candidate)
            return e && atoi(e) == 1;
        }();
        if (whole_windows) {
            hipLaunchKernelGGL(ws_left_top3_kernel, dim3(ceil_div(iw, 256), ih), dim3(256), 0, st, a, top3, g.w1);
        } else {
            uint32_t *vol = reinterpret_cast<uint32_t *>(sync + smooth_left_sync_bytes(g.w1, g.h1));
            const int rows = ih + 2 * half; // image rows 0 .. rows - 1 carry the windows of the interior
            const int d_max = std::min(g.max_d, g.w1 - 1 - 2 * half); // (beyond: no column has such a candidate)
            int d_top = d_max;
            bool first = true, last = false;
            do { // (no candidate anywhere -- d_max < 1 -- still runs once: the entries say "none")
                const int nslab = std::max(0, std::min(kTopSlab, d_top));
                last = d_top - kTopSlab < 1;
                if (nslab > 0)
                    hipLaunchKernelGGL(ws_left_top3_rows_kernel, dim3(ceil_div(iw, 256), rows, nslab), dim3(256), 0, st, a, vol, rows, d_top);
                hipLaunchKernelGGL(ws_left_top3_cols_kernel, dim3(ceil_div(iw, 256), ih), dim3(256), 0, st, a, vol, rows, d_top, nslab,
                                   first ? 1 : 0, last ? 1 : 0, top3, g.w1);
                first = false;
                d_top -= kTopSlab;
            } while (!last);
        }
    }
    // hand-off words all ones ("not there yet"), ticket and error words zero
    const int nbands = ceil_div(ih, kBandRows);
    const size_t pitch = smooth_left_edge_pitch(g.w1);
    hipError_t e = hipMemsetAsync(sync + 64, 0xff, (size_t)(nbands + 1) * pitch * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(sync, 0, 64, st);
    if (e != hipSuccess) return e;
    unsigned long long *edge = reinterpret_cast<unsigned long long *>(sync + 64);
    unsigned int *ctrl = reinterpret_cast<unsigned int *>(sync);
    // the planes' moving windows in LDS when the marching kernel left its planes behind and they fit
    const int cwa = 2 * half + kBandRows + 34 + kBandFill, cwb = g.max_d + 2 * half + kBandRows + 34 + kBandFill, rp = kBandRows + 2 * half + 2;
    const size_t lds = (size_t)(cwa + cwb) * rp * sizeof(uint32_t);
    // ... and, for the block sizes with a compile-time form, their row-major copies behind them (4 columns of slack
    // instead of 32, or 17 x 17 at D = 200 would not fit both)
    const int bs = g.block_size;
    const int cwa_t = 2 * half + kBandRows + 6 + kBandFill, cwb_t = g.max_d + cwa_t;
    const size_t lds_t = (size_t)(cwa_t + cwb_t) * rp * sizeof(uint32_t) +
                         (size_t)(((cwa_t + bs + 1) & ~1) + ((cwb_t + bs + 1) & ~1)) * (rp - 1) * sizeof(uint32_t);
    static const bool no_tw = [] {
        const char *e = getenv("WS_LEFT_TW"); // development knob: 0 = without the row-major copies
        return e && atoi(e) == 0;
    }();
    // (only the factors below 1 run the sliding sums at all: above, an unlisted neighbour value never wins)
    const bool tw = !no_tw && !(s >= 1.0) && lds_t <= 158 * 1024 && rp - 1 <= kBandLanes;
    auto launch = [&](auto kernel, size_t bytes, int ca, int cb) -> hipError_t {
        if (bytes > 48 * 1024) {
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (err != hipSuccess) return err;
        }
        hipLaunchKernelGGL(kernel, dim3(nbands), dim3(kBandLanes), bytes, st, a, top3, g.w1, edge, (int)pitch, ctrl, ca, cb);
        return hipGetLastError();
    };
    if (canon && lds <= 152 * 1024 && 2 * half + 1 + kBandRows <= 128) {
        // (block sizes of the BASELINE configs and of the reference's own call get the compile-time form)
#define WS_LEFT_BS_ONE(MODE, N)                                                                       \
        case N: return tw ? launch(ws_smooth_left_bands_kernel<MODE, N, true>, lds_t, cwa_t, cwb_t)  \
                          : launch(ws_smooth_left_bands_kernel<MODE, N>, lds, cwa, cwb);
#define WS_LEFT_BS(MODE)                                                                              \
        switch (g.block_size) {                                                                       \
        WS_LEFT_BS_ONE(MODE, 5)                                                                       \
        WS_LEFT_BS_ONE(MODE, 7)                                                                       \
        WS_LEFT_BS_ONE(MODE, 9)                                                                       \
        WS_LEFT_BS_ONE(MODE, 17)                                                                      \
        default: return launch(ws_smooth_left_bands_kernel<MODE, 0>, lds, cwa, cwb);                  \
        }
        if (!a.ssd) { WS_LEFT_BS(0) }
        if (a.centred) { WS_LEFT_BS(2) }
        WS_LEFT_BS(1)
#undef WS_LEFT_BS
#undef WS_LEFT_BS_ONE
    }
    return launch(ws_smooth_left_bands_kernel<-1>, 0, cwa, cwb);
}

// ---- bit-parallel form for 0 <= smoothFactor <= 1 -------------------------------------------
// There c0 * s^k does not grow with k, so t_0 >= t_1 >= t_2 and a pixel is one of: always 0
// ("generate"), never 0 ("kill"), or 0 exactly when its left neighbour is ("propagate") -- a
// carry chain.  With the codes packed into bit planes one lane resolves a word of columns with a
// single addition (A = g|p, B = g: the carries of A+B are the chain).  Words hold 31 columns,
// so the carry out of a word is the top bit of the sum; a lane takes one word, or two for
// images wider than 1984 (3968 at most; wider ones go to the other resolvers).
// The words of a row are skewed in time: lane l works on row t - l at step t, so the carry into
// its word is what lane l-1 produced one step earlier (one DPP shift) and the flags of the row
// above are its own previous result -- a systolic array in one wave, no scalar unit, no ballots.
// A lone wave issues an instruction every ~8 ns whatever it is, so the step is counted in
// instructions: 6 of arithmetic, one LDS read, and a wait / store / address update per 4 steps.
// The planes are stored the way the lanes walk them: per LDS chunk, per word, the skewed rows
// (word l of image row y is skewed row y + l) one entry after the other.
template <int NS> struct alignas(16) PlaneEntry { uint32_t v[NS][4]; }; // per sub-word: n0, n1, n2, padding (one b128 LDS read)

// NS = 31-column sub-words per lane: 1, or 2 for images wider than 1984 (the second one takes the first one's
// carry in the same step: 4 more instructions a step, against 64-bit adds at a quarter of the rate).
template <int NS> struct BitsLayout {
    static constexpr int kBits = 31 * NS; // columns per lane
    int nw, steps, chunk_rows, nchunks;
    size_t plane_entries, z_words;
    __host__ BitsLayout(int w, int rows)
    {
        nw = ceil_div(w, kBits);
        steps = round_up(rows + nw, 8);
        // two chunks of chunk_rows entries (+ 16 bytes) per lane in ~128 KB of LDS; the odd 16 bytes keep the
        // lanes' b128 reads on different banks (an even stride in 16-byte units: 8 lanes per bank group, measured
        // 118 instead of 60 ns a step at two sub-words per lane)
        chunk_rows = (int)((65536 - 16 * (size_t)nw) / ((size_t)nw * sizeof(PlaneEntry<NS>))) / 8 * 8;
        if (chunk_rows > 64) chunk_rows = 64;
        if (chunk_rows < 8) chunk_rows = 8;
        nchunks = ceil_div(steps, chunk_rows);
        plane_entries = ((size_t)nchunks * nw * lane_bytes() + sizeof(PlaneEntry<NS>) - 1) / sizeof(PlaneEntry<NS>);
        z_words = ((size_t)nw * nchunks * chunk_rows + 8) * NS; // + the idle lanes' scratch
    }
    __host__ int z_pitch() const { return nchunks * chunk_rows; }
    __host__ size_t lane_bytes() const { return (size_t)chunk_rows * sizeof(PlaneEntry<NS>) + 16; }
    __host__ size_t bytes() const { return plane_entries * sizeof(PlaneEntry<NS>) + z_words * 4 + 64; }
    __host__ size_t chunk_bytes() const { return (size_t)nw * lane_bytes(); }
    __host__ size_t lds_bytes() const { return 2 * chunk_bytes() + 8 * sizeof(PlaneEntry<NS>); }
};
constexpr int kBitsMaxWidth1 = 31 * 64, kBitsMaxWidth2 = 62 * 64;

template <int NS>
__global__ void __launch_bounds__(256) ws_smooth_planes_kernel(const uint8_t *__restrict__ sel, int sel_pitch, int w, int rows,
                                                               PlaneEntry<NS> *__restrict__ planes, int nw, int chunk_rows)
{
    // a wave takes 62 columns = two sub-words (lanes 0..30 and 31..61): two lanes' words, of two image rows
    // (NS = 1), or the two halves of one lane's word (NS = 2)
    const int lane = threadIdx.x & 63, sub = lane / 31;
    const int gsw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + sub; // sub-word of the row
    const int word = gsw / NS;                                       // = the lane that resolves it
    const int x = gsw * 31 + (lane - sub * 31);
    const int y = (int)blockIdx.y - word; // blockIdx.y is the skewed row
    // beyond the row: fixed, non-zero; beyond the image (the skew's two triangles): the same, all planes 0
    const bool inside = sub < 2 && word < nw && x < w && y >= 0 && y < rows;
    const uint32_t c = inside ? sel[(size_t)y * sel_pitch + x] : kSelFixed;
    const bool fixed = c & kSelFixed;
    // n_k = "this pixel is 0 when k of its two neighbours are": !t_k for a free pixel, the fixed value's
    // zero flag otherwise (the planes the resolver selects from with the upper neighbour's flags)
    const bool zf = fixed && (c & kSelZero);
    const unsigned long long b0 = __ballot(fixed ? zf : !(c & 1)), b1 = __ballot(fixed ? zf : !(c & 2)),
                             b2 = __ballot(fixed ? zf : !(c & 4));
    if (sub < 2 && lane == sub * 31 && word < nw) {
        const uint32_t mask = 0x7fffffffu;
        const uint32_t n0 = (uint32_t)(b0 >> (sub * 31)) & mask;
        const uint32_t n1 = ((uint32_t)(b1 >> (sub * 31)) & mask) | n0; // t_0 >= t_1 >= t_2, spelled out: the resolver's
        const uint32_t n2 = ((uint32_t)(b2 >> (sub * 31)) & mask) | n1; // generate is n0 without "& n1"
        const int chunk = blockIdx.y / chunk_rows, r = blockIdx.y - chunk * chunk_rows;
        // per chunk and lane: chunk_rows entries + 16 bytes (see BitsLayout)
        const size_t lane_bytes = (size_t)chunk_rows * sizeof(PlaneEntry<NS>) + 16;
        uint8_t *q = reinterpret_cast<uint8_t *>(planes) + ((size_t)chunk * nw + word) * lane_bytes +
                     ((size_t)r * NS + (gsw - word * NS)) * 16;
        *reinterpret_cast<uint4 *>(q) = make_uint4(n0, n1, n2, 0u);
    }
}

// (s0 & s1) | (~s0 & s2) in one instruction (the compiler splits the pattern once ~s0 has a second use)
__device__ __forceinline__ uint32_t bfi(uint32_t s0, uint32_t s1, uint32_t s2)
{
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2));
    return d;
}

// B | (A & (S ^ A ^ B)) -- the carry out of every column of S = A + B + carry-in -- as one three-input bit
// operation (truth table over (A, B, S), A the high index bit: 0xdc); the compiler finds it for one sub-word
// per lane and spends three instructions on it for two.
__device__ __forceinline__ uint32_t carry_out_bits(uint32_t A, uint32_t B, uint32_t S)
{
    uint32_t z;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xdc" : "=v"(z) : "v"(A), "v"(B), "v"(S));
    return z;
}

// Four plane entries (four steps of one lane) from LDS, issued where they are written and waited for where
// they are used: the compiler gathers plain LDS reads in front of one s_waitcnt, which puts the LDS latency
// back on the chain.  LDS returns in order, so "at most the four younger entries still in flight" means
// these have arrived; the wait takes the registers as in/out operands so that no use can move in front of it.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NS> struct PlaneRegs4;
template <> struct PlaneRegs4<1> {
    u32x4 e[4][1];
    __device__ __forceinline__ void read(uint32_t addr)
    {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"
                     "ds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
                     : "=&v"(e[0][0]), "=&v"(e[1][0]), "=&v"(e[2][0]), "=&v"(e[3][0])
                     : "v"(addr));
    }
    __device__ __forceinline__ void arrived()
    {
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(e[0][0]), "+v"(e[1][0]), "+v"(e[2][0]), "+v"(e[3][0]));
    }
};
template <> struct PlaneRegs4<2> {
    u32x4 e[4][2];
    __device__ __forceinline__ void read(uint32_t addr)
    {
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\t"
                     "ds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                     "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
                     "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112"
                     : "=&v"(e[0][0]), "=&v"(e[0][1]), "=&v"(e[1][0]), "=&v"(e[1][1]), "=&v"(e[2][0]), "=&v"(e[2][1]),
                       "=&v"(e[3][0]), "=&v"(e[3][1])
                     : "v"(addr));
    }
    __device__ __forceinline__ void arrived()
    {
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(e[0][0]), "+v"(e[0][1]), "+v"(e[1][0]), "+v"(e[1][1]), "+v"(e[2][0]), "+v"(e[2][1]), "+v"(e[3][0]),
                       "+v"(e[3][1]));
    }
};

template <int NS>
__global__ void __launch_bounds__(64) ws_smooth_resolve_bits_kernel(const PlaneEntry<NS> *__restrict__ planes, int nw,
                                                                    int nchunks, uint32_t *__restrict__ zplane, int chunk_rows)
{
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    typedef __attribute__((address_space(1))) const void glb_void;
    constexpr int E = (int)sizeof(PlaneEntry<NS>);
    extern __shared__ uint4 ws_smem4[];
    uint8_t *lds = reinterpret_cast<uint8_t *>(ws_smem4);
    const int lane = threadIdx.x;
    const int lane_bytes = chunk_rows * E + 16; // one lane's entries of a chunk (+ the odd 16 bytes)
    const int chunk_bytes = nw * lane_bytes;     // a multiple of 16
    const uint8_t *src0 = reinterpret_cast<const uint8_t *>(planes);
    for (int o = lane * 16; o < chunk_bytes; o += 1024)
        __builtin_amdgcn_global_load_lds((glb_void *)(src0 + o), (lds_void *)(lds + (o - lane * 16)), 16, 0, 0);
    uint32_t zprev[NS] = {};
    uint32_t cflag = 0;
    // Branch-free steps: lanes beyond the image's words read zeroed entries behind the two chunks (so they
    // produce no carry) and store to scratch words behind the resolved plane.
    const bool active = lane < nw;
    if (lane < 8 * E / 16) reinterpret_cast<uint4 *>(lds + 2 * chunk_bytes)[lane] = make_uint4(0, 0, 0, 0);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8 *)lds;
    const uint32_t rstep = active ? 4u * E : 0u;
    const int z_pitch = nchunks * chunk_rows;
    uint32_t *zp = zplane + (size_t)(active ? lane : nw) * z_pitch * NS;
    const int zstep = active ? 4 * NS : 0;
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // the chunk is there (and the zero entries)
        if (c + 1 < nchunks) {
            const uint8_t *src = src0 + (size_t)(c + 1) * chunk_bytes;
            uint8_t *dst = lds + ((c + 1) & 1) * chunk_bytes;
            for (int o = lane * 16; o < chunk_bytes; o += 1024)
                __builtin_amdgcn_global_load_lds((glb_void *)(src + o), (lds_void *)(dst + (o - lane * 16)), 16, 0, 0);
        }
        uint32_t row = active ? lds0 + (uint32_t)((c & 1) * chunk_bytes + lane * lane_bytes) : lds0 + (uint32_t)(2 * chunk_bytes);
        asm("" : "+v"(row)); // a running address, one addition per four steps, not re-derived from the base
        // Entries are read four steps ahead of the chain through zprev (the four read behind a lane's last
        // one are dropped: they are the next lane's, the other chunk's or the zero entries).
        PlaneRegs4<NS> pa, pb;
        pa.read(row);
        auto step4 = [&](PlaneRegs4<NS> &p) {
            p.arrived();
            uint32_t z[4][NS];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // the carry out of the lane to the left, one step ago: the same image row (lane 0: none)
                uint32_t carry = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cflag, 0x138, 0xf, 0xf, true);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    // zero when the left neighbour is not / is zero, given the upper neighbour's flag:
                    // B = generate, A = generate | propagate (n0 is inside n1: the planes kernel saw to it)
                    const uint32_t B = bfi(zprev[k], p.e[i][k].y, p.e[i][k].x), A = bfi(zprev[k], p.e[i][k].z, p.e[i][k].y);
                    const uint32_t S = A + B + carry;
                    carry = S >> 31; // into the next sub-word, or the next lane's first one a step later
                    // S ^ A ^ B: bit j = carry into column j = "the left neighbour is zero"; the carry out of
                    // column j: the pixel is zero
                    z[i][k] = carry_out_bits(A, B, S);
                    zprev[k] = z[i][k];
                }
                cflag = carry;
            }
            uint4 *q = reinterpret_cast<uint4 *>(zp);
            if constexpr (NS == 1) {
                q[0] = make_uint4(z[0][0], z[1][0], z[2][0], z[3][0]);
            } else {
                q[0] = make_uint4(z[0][0], z[0][1], z[1][0], z[1][1]);
                q[1] = make_uint4(z[2][0], z[2][1], z[3][0], z[3][1]);
            }
            zp += zstep;
        };
        for (int t = 0; t < chunk_rows; t += 8) { // chunk_rows is a multiple of 8
            row += rstep; pb.read(row);
            step4(pa);
            row += rstep; pa.read(row);
            step4(pb);
        }
    }
}

template <int NS>
__global__ void __launch_bounds__(256) ws_smooth_apply_kernel(float *out, int out_pitch, int w, int rows,
                                                              const uint8_t *__restrict__ sel, int sel_pitch,
                                                              const uint32_t *__restrict__ zplane, int z_pitch)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= rows) return;
    if (sel[(size_t)y * sel_pitch + x] & kSelFixed) return;
    const int sw = x / 31, word = sw / NS; // sub-word of the row, lane that resolved it (skewed rows)
    if ((zplane[((size_t)word * z_pitch + y + word) * NS + (sw - word * NS)] >> (x - sw * 31)) & 1)
        out[(size_t)y * out_pitch + x] = 0.0f;
}

template <int NS>
static hipError_t launch_smooth_bits(const GenericArgs &g, const uint8_t *sel, int sel_pitch, unsigned long long *buf, int rows,
                                     hipStream_t st)
{
    const BitsLayout<NS> lay(g.w2, rows);
    PlaneEntry<NS> *planes = reinterpret_cast<PlaneEntry<NS> *>(buf);
    uint32_t *zplane = reinterpret_cast<uint32_t *>(planes + lay.plane_entries);
    hipLaunchKernelGGL(ws_smooth_planes_kernel<NS>, dim3(ceil_div(lay.nw * NS, 8), lay.nchunks * lay.chunk_rows), dim3(256), 0, st,
                       sel, sel_pitch, g.w2, rows, planes, lay.nw, lay.chunk_rows);
    if (lay.lds_bytes() > 48 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(ws_smooth_resolve_bits_kernel<NS>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.lds_bytes());
        if (err != hipSuccess) return err;
    }
    hipLaunchKernelGGL(ws_smooth_resolve_bits_kernel<NS>, dim3(1), dim3(64), lay.lds_bytes(), st, planes, lay.nw, lay.nchunks,
                       zplane, lay.chunk_rows);
    hipLaunchKernelGGL(ws_smooth_apply_kernel<NS>, dim3(ceil_div(g.w2, 256), rows), dim3(256), 0, st, g.out, g.out_pitch, g.w2, rows,
                       sel, sel_pitch, zplane, lay.z_pitch());
    return hipGetLastError();
}

int smooth_sel_rows(int rows) { return (rows + 15) / 16 * 16 + 16; }

size_t smooth_planes_bytes(int w, int h)
{
    if (w <= kBitsMaxWidth1) return BitsLayout<1>(w, h).bytes();
    if (w <= kBitsMaxWidth2) return BitsLayout<2>(w, h).bytes();
    return 64; // wider images take the other resolvers
}

hipError_t launch_smooth(const GenericArgs &g, double s, uint8_t *sel, int sel_pitch, unsigned long long *planes,
                         const Canon *canon, Plane pa, Plane pb, const int32_t *cost, int cost_pitch, hipStream_t st)
{
    if (canon) { // right view after the marching kernel: everything on the packed planes + the cost plane
        PreparePlanesArgs a{};
        a.A = pa.data; a.B = pb.data;
        a.pitch_a = pa.pitch; a.pad_a = pa.pad; a.pitch_b = pb.pitch; a.pad_b = pb.pad;
        a.wa = canon->wa; a.ha = canon->ha; a.wb = canon->wb; a.height = std::min(canon->ha, canon->hb);
        a.half = canon->wh / 2; a.max_d = g.max_d;
        a.ww = canon->ww; a.wh = canon->wh; a.wx0 = canon->wx0; a.wy0 = canon->wy0;
        a.boff = canon->boff; a.d_hi = g.max_d - 1; a.b_lo = canon->b_lo;
        a.ox0 = canon->ox0; a.ox1 = canon->ox1; a.oy0 = canon->oy0; a.oy1 = canon->oy1;
        a.skip_x0 = g.skip_x0; a.skip_x1 = g.skip_x1; a.skip_y0 = g.skip_y0; a.skip_y1 = g.skip_y1;
        a.s = s; a.out = g.out; a.out_pitch = g.out_pitch; a.cost = cost; a.cost_pitch = cost_pitch;
        a.sel = sel; a.sel_pitch = sel_pitch;
        if (!cost || canon->ww > kBoxMaxW || canon->wh > kBoxMaxW) return hipErrorInvalidValue;
        dim3 gi(ceil_div(canon->ox1 - canon->ox0, 64), ceil_div(canon->oy1 - canon->oy0, kBoxRows));
        const long long inside = (long long)(g.skip_x1 - g.skip_x0) * (g.skip_y1 - g.skip_y0);
        const long long nring = (long long)canon->wa * canon->ha - (inside > 0 ? inside : 0);
        dim3 gr((unsigned)ceil_div((int)nring, 4)); // a wave per ring pixel, 4 per workgroup
        if (!canon->ssd) {
            hipLaunchKernelGGL((ws_smooth_prepare_box_kernel<false, false>), gi, dim3(256), 0, st, a);
            if (nring > 0) hipLaunchKernelGGL((ws_smooth_prepare_ring_kernel<false, false>), gr, dim3(256), 0, st, a);
        } else if (march_centred(*canon)) {
            hipLaunchKernelGGL((ws_smooth_prepare_box_kernel<true, true>), gi, dim3(256), 0, st, a);
            if (nring > 0) hipLaunchKernelGGL((ws_smooth_prepare_ring_kernel<true, true>), gr, dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL((ws_smooth_prepare_box_kernel<true, false>), gi, dim3(256), 0, st, a);
            if (nring > 0) hipLaunchKernelGGL((ws_smooth_prepare_ring_kernel<true, false>), gr, dim3(256), 0, st, a);
        }
    } else {
        dim3 grid(ceil_div(g.w2, 256), g.h2);
        hipLaunchKernelGGL(ws_smooth_prepare_kernel, grid, dim3(256), 0, st, g, s, sel, sel_pitch);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int rows = std::min(g.h1, g.h2);
    if (s >= 0.0 && s <= 1.0 && planes && g.w2 <= kBitsMaxWidth2)
        return g.w2 <= kBitsMaxWidth1 ? launch_smooth_bits<1>(g, sel, sel_pitch, planes, rows, st)
                                      : launch_smooth_bits<2>(g, sel, sel_pitch, planes, rows, st);
    const int per = ceil_div(g.w2, 64);
    // rows per LDS chunk: two chunks in at most 64 KB, a multiple of 16 rows (whole 1 KB DMA pieces)
    int chunk = (32768 / sel_pitch) / 16 * 16;
    if (chunk > 64) chunk = 64;
    const size_t wl = (size_t)2 * chunk * sel_pitch;
    if (per <= 8 && chunk >= 16)
        hipLaunchKernelGGL(ws_smooth_resolve_wave_kernel<8>, dim3(1), dim3(64), wl, st, g.out, g.out_pitch, g.w2, rows, sel, sel_pitch, chunk);
    else if (per <= 16 && chunk >= 16)
        hipLaunchKernelGGL(ws_smooth_resolve_wave_kernel<16>, dim3(1), dim3(64), wl, st, g.out, g.out_pitch, g.w2, rows, sel, sel_pitch, chunk);
    else if (per <= 32 && chunk >= 16)
        hipLaunchKernelGGL(ws_smooth_resolve_wave_kernel<32>, dim3(1), dim3(64), wl, st, g.out, g.out_pitch, g.w2, rows, sel, sel_pitch, chunk);
    else {
        const size_t lds = (size_t)((g.w2 + 3) & ~3) + 16 * sizeof(uint32_t);
        hipLaunchKernelGGL(ws_smooth_resolve_kernel, dim3(1), dim3(1024), lds, st, g.out, g.out_pitch, g.w2, rows, sel,
                           sel_pitch);
    }
    return hipGetLastError();
}

} // namespace wsamd

#ifdef WS_BAND_STAMPS
extern "C" int ws_debug_band_stamps(unsigned long long *dst, int *bands, int *steps, int *points)
{
    *bands = wsamd::kStampBands; *steps = wsamd::kStampSteps; *points = wsamd::kStampPoints;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wsamd::ws_band_stamps), sizeof(wsamd::ws_band_stamps));
}
#endif
