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
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace wsamd {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// LDS regions of the target-image rows: a thread's run there starts at quad (X/4)*r + (ND/4)*k, so the
// region of its m-th quad is only fixed at compile time if the region count divides both
__host__ __device__ constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }
__host__ __device__ constexpr int march_nreg_b(int x, int nd) { return gcd_c(x / 4, nd / 4); }

// dwords per LDS region for a row of n dwords split into nreg regions (+1 quad: runs may over-read)
__host__ __device__ constexpr int march_region_dwords(int n, int nreg)
{
    return 4 * ((((n + 3) >> 2) + nreg - 1) / nreg + 1);
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

// ------------------------------------------------------------------------------------------
// the marching kernel
// ------------------------------------------------------------------------------------------
struct MarchArgs {
    const uint32_t *A;
    const uint32_t *B;
    const int32_t *bias; // SSD only
    float *out;
    int pitch_a, pad_a, pitch_b, pad_b, pitch_bi, pad_bi, out_pitch;
    int wa;
    int nxr, nch;
    int wx0, wy0, boff;
    int d_lo, d_hi, b_lo, b_hi;
    int d_first;   // first disparity of chunk 0 in THIS launch (d_lo + pass * chunks * ND)
    int pass_mode; // 0 = the only pass, 1 = first, 2 = middle, 3 = last of several d-group passes
    void *keys;    // several passes: plane of the best keys so far (slot_t per pixel)
    int keys_pitch;
    int ox0, ox1, oy0, oy1;
    int strip_rows, tiles, strips;
    int prefer_large, mirror, fallback_neg;
    int tag_bits; // SAD: keys are (cost << tag_bits) | global tie tag
};

// LDS row layout.  A thread reads runs of consecutive pixels starting at column X*r; with a
// plain row-major row the 16 lanes that share a ds_read_b128 cycle sit 4*X bytes apart and fall
// on every (X/4)-th bank group only.  So a row is stored as NREG = X/4 regions: region j holds
// the quads (16-byte groups of 4 pixels) whose index is j mod NREG, densely.  Lane r's m-th quad
// is then quad r + m/NREG of region m%NREG: consecutive lanes read consecutive 16-byte slots and
// every read is conflict free.  `ro` = dwords per region.
template <int NREG>
__device__ __forceinline__ int lds_phys(int q, int ro)
{
    const int quad = q >> 2;
    return (quad % NREG) * ro + (quad / NREG) * 4 + (q & 3);
}

// N consecutive logical dwords starting at a quad this thread's run starts with
// (base = row + 4 * first quad index inside region 0).
template <int N, int NREG>
__device__ __forceinline__ void lds_run(uint32_t (&dst)[N], const uint32_t *base, int ro)
{
    constexpr int Q = (N + 3) / 4;
#pragma unroll
    for (int m = 0; m < Q; ++m) {
        const uint4 v = *reinterpret_cast<const uint4 *>(base + (m % NREG) * ro + (m / NREG) * 4);
        if (4 * m + 0 < N) dst[4 * m + 0] = v.x;
        if (4 * m + 1 < N) dst[4 * m + 1] = v.y;
        if (4 * m + 2 < N) dst[4 * m + 2] = v.z;
        if (4 * m + 3 < N) dst[4 * m + 3] = v.w;
    }
}

// Asynchronous HBM -> LDS copy of one row (n dwords, 16-byte aligned source) into the region
// layout, by the whole workgroup: global_load_lds_dwordx4, no VGPR staging.  The LDS address of
// an LDS-DMA is wave-uniform base + lane * 16, so consecutive lanes fill consecutive quads of one
// region and each lane fetches the quad that belongs there (the source address carries the
// permutation).  Completion is covered by the vmcnt(0) hipcc places before the barrier.
template <int NREG>
__device__ __forceinline__ void stage_row_async(uint32_t *row, int ro, const uint32_t *gsrc, int n,
                                                int tid, int nt)
{
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    const int lane = tid & 63;
    const int nquads = (n + 3) >> 2;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const int nidx = (nquads - j + NREG - 1) / NREG; // quads of this region
        for (int idx = tid; idx < nidx; idx += nt)
            __builtin_amdgcn_global_load_lds((glb_void *)(gsrc + 4 * (idx * NREG + j)),
                                             (lds_void *)(row + j * ro + 4 * (idx - lane)), 16, 0, 0);
    }
}

// One row entering (SIGN=+1) or leaving (SIGN=-1) the window of every (column, disparity) this
// thread owns.
//   SAD: V = (window sum << shift) + global tie tag          key = V
//   SSD: V = local tie tag - (2 * cross sum << LT)           key = bias[xb] + V
//        (bias = box sum of the squared target pixels << LT, or poison for an invalid centre)
// With KEY the candidate keys are folded into best[] (signed min; equal costs go to the smaller
// tag, i.e. to the disparity the reference's strict '<' keeps).
template <int X, int ND, int WW, bool SSD, bool CENTRED, int SIGN, bool KEY>
__device__ __forceinline__ void march_row(int32_t (&V)[X][ND], int32_t (&best)[X],
                                          const uint32_t *runA, int ro_a, const uint32_t *runB,
                                          int ro_b, const int32_t *runBias, int ro_bi, int shift)
{
    constexpr int NREG = X / 4, NREGB = march_nreg_b(X, ND);
    constexpr int NA = X + WW - 1;
    constexpr int NB = NA + ND - 1;
    constexpr int NBI = X + ND - 1;
    // SAD accumulates +cost, SSD accumulates -2*cross: flip the sign of the update for SSD
    constexpr bool ADD = ((SIGN > 0) != SSD);

    uint32_t pa[NA], pb[NB];
    lds_run<NA, NREG>(pa, runA, ro_a);
    lds_run<NB, NREGB>(pb, runB, ro_b);
    uint32_t bi[NBI];
    if constexpr (KEY && SSD) lds_run<NBI, NREGB>(bi, reinterpret_cast<const uint32_t *>(runBias), ro_bi);

    // two disparities at a time: two independent prefix chains interleave in the issue stream
    // (a v_dot4 needs a wait state before its result can feed the next v_dot4's accumulator)
#pragma unroll
    for (int j = 0; j < ND; j += 2) {
        uint32_t S0[NA], S1[NA];
        uint32_t s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const uint32_t b0 = pb[i - j + ND - 1], b1 = pb[i - j + ND - 2];
            s0 = SSD ? pix_dot<CENTRED>(pa[i], b0, s0) : pix_sad(pa[i], b0, s0);
            s1 = SSD ? pix_dot<CENTRED>(pa[i], b1, s1) : pix_sad(pa[i], b1, s1);
            S0[i] = s0;
            S1[i] = s1;
        }
#pragma unroll
        for (int x = 0; x < X; ++x) {
            const uint32_t w0 = ADD ? S0[x + WW - 1] - (x ? S0[x - 1] : 0u) : (x ? S0[x - 1] : 0u) - S0[x + WW - 1];
            const uint32_t w1 = ADD ? S1[x + WW - 1] - (x ? S1[x - 1] : 0u) : (x ? S1[x - 1] : 0u) - S1[x + WW - 1];
            V[x][j] = (int32_t)((w0 << shift) + (uint32_t)V[x][j]);
            V[x][j + 1] = (int32_t)((w1 << shift) + (uint32_t)V[x][j + 1]);
            if constexpr (KEY) {
                const int32_t k0 = SSD ? (int32_t)bi[x - j + ND - 1] + V[x][j] : V[x][j];
                const int32_t k1 = SSD ? (int32_t)bi[x - j + ND - 2] + V[x][j + 1] : V[x][j + 1];
                best[x] = min(best[x], min(k0, k1));
            }
        }
    }
}

template <int X, int ND, int WW, int WH, bool SSD, int MAXT>
__global__ void __launch_bounds__(MAXT) ws_march_kernel(const MarchArgs g)
{
    static_assert(X % 4 == 0 && ND % 4 == 0, "runs start on 16-byte quads");
    constexpr int NREG = X / 4, NREGB = march_nreg_b(X, ND);
    constexpr int LT = ilog2c(ND);
    constexpr bool CENTRED = SSD && ssd_needs_centring(WW, WH, ND);
    constexpr int NR = WH + 2; // ring rows: WH+1 in use by a step, 1 being filled for the next
    typedef typename std::conditional<SSD, unsigned long long, int32_t>::type slot_t;
    const slot_t kEmpty = SSD ? (slot_t)~0ull : (slot_t)INT_MAX;

    extern __shared__ uint4 ws_smem4[];
    uint32_t *smem = reinterpret_cast<uint32_t *>(ws_smem4);

    const int NT = blockDim.x, tid = threadIdx.x;
    const int tx = g.nxr * X, dt = g.nch * ND;
    const int n_a = tx + WW - 1, n_b = tx + WW + dt - 2, n_bi = tx + dt - 1;
    const int ro_a = march_region_dwords(n_a, NREG), ro_b = march_region_dwords(n_b, NREGB);
    const int ro_bi = SSD ? march_region_dwords(n_bi, NREGB) : 0;
    const int a_w = NREG * ro_a, b_w = NREGB * ro_b, bi_w = NREGB * ro_bi;
    uint32_t *ringA = smem;
    uint32_t *ringB = ringA + NR * a_w;
    int32_t *biasr = reinterpret_cast<int32_t *>(ringB + NR * b_w);
    slot_t *slots = reinterpret_cast<slot_t *>(biasr + 2 * bi_w);

    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (each with its own
    // L2), so ids b and b+8 share one.  Give every XCD a contiguous range of (strip, tile) pairs:
    // the tiles of a strip overlap in the target-image columns they read and then hit the same L2.
    const int nblk = gridDim.x; // padded to a multiple of 8 by the launcher
    const int logical = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    if (logical >= g.tiles * g.strips) return; // uniform per workgroup
    const int tile_x0 = g.ox0 + (logical % g.tiles) * tx;
    const int ys = g.oy0 + (logical / g.tiles) * g.strip_rows;
    const int ye = min(ys + g.strip_rows, g.oy1);
    if (ys >= ye) return; // uniform per workgroup

    const int dhi_t = g.d_first + dt - 1;
    const uint32_t *gA = g.A + (tile_x0 + g.wx0 + g.pad_a);
    const uint32_t *gB = g.B + (tile_x0 + g.wx0 + g.boff - dhi_t + g.pad_b);
    const uint32_t *gBi = SSD ? reinterpret_cast<const uint32_t *>(g.bias) + (tile_x0 + g.boff - dhi_t + g.pad_bi) : nullptr;

    for (int k = tid; k < 2 * tx; k += NT) slots[k] = kEmpty;

    const int r = tid % g.nxr, c = tid / g.nxr;
    const bool worker = c < g.nch;
    // run starts (dword offset inside region 0): A at column X*r, B / bias at column X*r + ND*(nch-1-c)
    const int ia = 4 * r;
    const int ib = 4 * (((X / 4) * r + (ND / 4) * (g.nch - 1 - (worker ? c : 0))) / NREGB);
    const int d0 = g.d_first + c * ND; // first disparity of this thread's chunk
    const int shift = SSD ? LT + 1 : g.tag_bits;
    // global tie tag of local tag jt is ctag + jt (SSD merge)
    const int ctag = g.prefer_large ? g.d_hi - d0 - (ND - 1) : d0 - g.d_lo;

    int32_t V[X][ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int d = d0 + j;
        if constexpr (SSD) {
            // local tag: the preferred disparity of a tie gets the smaller tag
            const int tag = g.prefer_large ? (ND - 1 - j) : j;
            const int32_t init = (d <= g.d_hi) ? tag : (kPoison + tag);
#pragma unroll
            for (int x = 0; x < X; ++x) V[x][j] = init;
        } else {
            // global tag; both validity rules (d range, target centre range) fold into V
            const int tag = g.prefer_large ? g.d_hi - d : d - g.d_lo;
#pragma unroll
            for (int x = 0; x < X; ++x) {
                const int xb = tile_x0 + r * X + x - d + g.boff;
                V[x][j] = (d <= g.d_hi && xb >= g.b_lo && xb <= g.b_hi) ? tag : kPoison;
            }
        }
    }

    const int ra0 = ys + g.wy0; // first window row of the first output row
    const int nsteps = (ye - ys) + WH - 1;

    // prologue: row ra0 (and the bias row of step 0 when the window is one row high)
    stage_row_async<NREG>(ringA, ro_a, gA + (size_t)ra0 * g.pitch_a, n_a, tid, NT);
    stage_row_async<NREGB>(ringB, ro_b, gB + (size_t)ra0 * g.pitch_b, n_b, tid, NT);
    if (SSD && WH == 1)
        stage_row_async<NREGB>(reinterpret_cast<uint32_t *>(biasr), ro_bi, gBi + (size_t)ys * g.pitch_bi, n_bi, tid, NT);
    __syncthreads();

    int add_slot = 0;      // ring slot of the row entering at this step   (a     mod NR)
    int sub_slot = 2 % NR; // ring slot of the row leaving at this step    (a-WH  mod NR)
    // image row of the output flushed at step a (row ys + a - WH) sits in slot (a - WH - wy0) mod NR
    int out_slot = ((-WH - g.wy0) % NR + NR) % NR;
    for (int a = 0; a <= nsteps; ++a) {
        const int oi = a - (WH - 1); // output row index inside the strip produced by this step

        // 1. hand the row finished in the previous step to HBM (a == nsteps: only this)
        if (oi >= 1) {
            const int y = ys + oi - 1;
            slot_t *sl = slots + ((oi - 1) & 1) * tx;
            const uint32_t *rowA = ringA + out_slot * a_w;
            for (int k = tid; k < tx; k += NT) {
                const int si = (k % X) * g.nxr + k / X; // slots are stored [x][r]
                slot_t key = sl[si];
                sl[si] = kEmpty;
                const int x = tile_x0 + k;
                if (x < g.ox1 && g.pass_mode != 0) {
                    // disparity ranges too wide for one tile run as several d-group passes that
                    // meet in a plane of keys (same keys, same ordering: min is the merge)
                    slot_t *kp = static_cast<slot_t *>(g.keys) + (size_t)y * g.keys_pitch + x;
                    if (g.pass_mode != 1) key = min(key, *kp);
                    if (g.pass_mode != 3) {
                        *kp = key;
                        continue;
                    }
                }
                if (x < g.ox1) {
                    const int xo = g.mirror ? g.wa - 1 - x : x;
                    float val;
                    if (key == kEmpty) {
                        val = g.fallback_neg ? -(float)xo : (float)xo;
                    } else {
                        const int gtag = SSD ? (int)(uint32_t)key : ((int)key & ((1 << g.tag_bits) - 1));
                        val = (float)(g.prefer_large ? g.d_hi - gtag : g.d_lo + gtag);
                    }
                    // black pixel (BlockSearch.cpp:41, :105): image row y, column x, from the ring
                    if (rowA[lds_phys<NREG>(k - g.wx0, ro_a)] == (CENTRED ? kCentre : 0u)) val = 0.0f;
                    g.out[(size_t)y * g.out_pitch + xo] = val;
                }
            }
        }
        if (a == nsteps) break;

        // 2. start the copy of the next step's rows into the ring slot nobody reads this step
        int nxt_slot = add_slot + 1;
        if (nxt_slot == NR) nxt_slot = 0;
        if (a + 1 < nsteps) {
            stage_row_async<NREG>(ringA + nxt_slot * a_w, ro_a, gA + (size_t)(ra0 + a + 1) * g.pitch_a, n_a, tid, NT);
            stage_row_async<NREGB>(ringB + nxt_slot * b_w, ro_b, gB + (size_t)(ra0 + a + 1) * g.pitch_b, n_b, tid, NT);
            if (SSD && oi + 1 >= 0)
                stage_row_async<NREGB>(reinterpret_cast<uint32_t *>(biasr + ((oi + 1) & 1) * bi_w), ro_bi,
                                      gBi + (size_t)(ys + oi + 1) * g.pitch_bi, n_bi, tid, NT);
        }

        // 3. arithmetic
        if (worker) {
            int32_t best[X];
#pragma unroll
            for (int x = 0; x < X; ++x) best[x] = INT_MAX;
            if (a >= WH)
                march_row<X, ND, WW, SSD, CENTRED, -1, false>(V, best, ringA + sub_slot * a_w + ia, ro_a,
                                                     ringB + sub_slot * b_w + ib, ro_b, nullptr, 0, shift);
            if (oi >= 0) {
                march_row<X, ND, WW, SSD, CENTRED, +1, true>(V, best, ringA + add_slot * a_w + ia, ro_a,
                                                    ringB + add_slot * b_w + ib, ro_b,
                                                    biasr + (oi & 1) * bi_w + ib, ro_bi, shift);
                slot_t *sl = slots + (oi & 1) * tx + r;
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    const int32_t bk = best[x];
                    if constexpr (SSD) {
                        const int32_t t = bk >> LT;
                        if (t < (kValidKeyBound >> LT)) {
                            const uint32_t gtag = (uint32_t)(ctag + (bk & (ND - 1)));
                            const unsigned long long key =
                                ((unsigned long long)((uint32_t)t ^ 0x80000000u) << 32) | gtag;
                            atomicMin(sl + x * g.nxr, key); // ds_min_u64, lanes on consecutive slots
                        }
                    } else {
                        if (bk < kValidKeyBound) atomicMin(sl + x * g.nxr, bk); // ds_min_i32
                    }
                }
            } else {
                march_row<X, ND, WW, SSD, CENTRED, +1, false>(V, best, ringA + add_slot * a_w + ia, ro_a,
                                                     ringB + add_slot * b_w + ib, ro_b, nullptr, 0, shift);
            }
        }

        __syncthreads(); // also waits for the asynchronous row copies (vmcnt(0))
        add_slot = nxt_slot;
        if (++sub_slot == NR) sub_slot = 0;
        if (++out_slot == NR) out_slot = 0;
    }
}

// ---- instantiation table -----------------------------------------------------------------
#ifndef WS_X
#define WS_X 8
#endif
#ifndef WS_ND
#define WS_ND 8
#endif
#ifndef WS_MAXT
#define WS_MAXT 512
#endif
constexpr int kX = WS_X, kND = WS_ND, kMaxT = WS_MAXT; // build-time tuning (tools/variants.py)
constexpr int kMaxChunks = 64; // at most 512 disparities per tile and pass (tools/time_calls.py)
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
    // left view: bs x bs                       right view: (bs-1) x (bs-1)
    WS_MARCH_ENTRY(3, 3), WS_MARCH_ENTRY(5, 5), WS_MARCH_ENTRY(7, 7), WS_MARCH_ENTRY(9, 9),
    WS_MARCH_ENTRY(11, 11), WS_MARCH_ENTRY(13, 13), WS_MARCH_ENTRY(15, 15), WS_MARCH_ENTRY(17, 17),
    WS_MARCH_ENTRY(2, 2), WS_MARCH_ENTRY(4, 4), WS_MARCH_ENTRY(6, 6), WS_MARCH_ENTRY(8, 8),
    WS_MARCH_ENTRY(10, 10), WS_MARCH_ENTRY(12, 12), WS_MARCH_ENTRY(14, 14), WS_MARCH_ENTRY(16, 16),
};

static const MarchEntry *find_march(const Canon &c)
{
    for (const MarchEntry &e : kMarchTable)
        if (e.ww == c.ww && e.wh == c.wh && e.ssd == c.ssd) return &e;
    return nullptr;
}

static int tag_bits_for(const Canon &c)
{
    int bits = 1;
    while ((1 << bits) < c.d_hi - c.d_lo + 1) ++bits;
    return bits;
}

int march_centred(const Canon &c) { return c.ssd && ssd_needs_centring(c.ww, c.wh, kND); }

bool march_supported(const Canon &c)
{
    if (!find_march(c)) return false;
    if (c.ox1 <= c.ox0 || c.oy1 <= c.oy0) return false;
    const int dcount = c.d_hi - c.d_lo + 1;
    if (dcount < 1) return false;
    // keys must stay inside (-2^28, 2^28)
    //   SSD: (2 * cross sum) << log2(ND)        SAD: window sum << tag bits
    const long long worst = c.ssd ? 2LL * c.ww * c.wh * 3 * (march_centred(c) ? 128 * 128 : 255 * 255) * kND
                                  : ((long long)c.ww * c.wh * 3 * 255) << tag_bits_for(c);
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
    // d-chunks per tile.  One tile holds at most kMaxChunks chunks (wider disparity ranges would
    // leave too few columns per tile); beyond that the range is cut into equal d-group passes that
    // meet in a plane of keys.
    const int nch_total = ceil_div(dcount, kND);
    static const int max_chunks = [] {
        const char *e = getenv("WS_MAX_CHUNKS"); // development knob
        const int v = e ? atoi(e) : 0;
        return v >= 8 && v <= kMaxT / kMinXRuns ? v : kMaxChunks;
    }();
    m.passes = ceil_div(nch_total, max_chunks);
    m.nch = ceil_div(nch_total, m.passes);
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
    const int nreg = kX / 4, nregb = march_nreg_b(kX, kND);
    const int a_w = nreg * march_region_dwords(tx + c.ww - 1, nreg),
              b_w = nregb * march_region_dwords(tx + c.ww + dt - 2, nregb),
              bi_w = nregb * march_region_dwords(tx + dt - 1, nregb);
    const int nr = c.wh + 2;
    m.lds_bytes = c.ssd ? (size_t)(nr * a_w + nr * b_w + 2 * bi_w) * 4 + (size_t)2 * tx * 8
                        : (size_t)(nr * a_w + nr * b_w) * 4 + (size_t)2 * tx * 4;
    if (m.lds_bytes > 160 * 1024) return false;
    *out = m;
    return true;
}

static int aligned_pad(int base)
{
    // smallest pad >= max(0, -base) that puts column `base` of the image on a 16-byte boundary
    int pad = base < 0 ? -base : 0;
    while (((base + pad) & 3) != 0) ++pad;
    return pad;
}

void march_plane_geometry(const Canon &c, const MarchLaunch &m, Plane *a, Plane *b, Plane *bias)
{
    const int tx = m.nxr * m.x_per_thread, dt = m.nch * m.nd_per_thread;
    const int dhi_t = c.d_lo + m.passes * dt - 1; // the last pass reaches furthest to the left
    const int n_a = tx + c.ww - 1, n_b = tx + c.ww + m.passes * dt - 2, n_bi = tx + m.passes * dt - 1;
    // first column each tile row copy starts at (tile 0); tiles advance by tx (a multiple of 8)
    const int base_a = c.ox0 + c.wx0;
    const int base_b = c.ox0 + c.wx0 + c.boff - dhi_t;
    const int base_bi = c.ox0 + c.boff - dhi_t;
    const int last = (m.tiles - 1) * tx;
    a->pad = aligned_pad(base_a);
    a->pitch = round_up(std::max(base_a + last + round_up(n_a, 4), c.wa) + a->pad + 4, 64);
    b->pad = aligned_pad(base_b);
    b->pitch = round_up(std::max(base_b + last + round_up(n_b, 4), c.wb) + b->pad + 4, 64);
    bias->pad = aligned_pad(base_bi);
    bias->pitch = round_up(base_bi + last + round_up(n_bi, 4) + bias->pad + 4, 64);
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

const char *march_kernel_name(const Canon &c, const MarchLaunch &)
{
    const MarchEntry *e = find_march(c);
    return e ? e->name : "";
}

hipError_t launch_march(const Canon &c, const MarchLaunch &m, Plane a, Plane b, Plane bias,
                        float *out, int out_pitch, void *keys, int keys_pitch, hipStream_t s)
{
    const MarchEntry *e = find_march(c);
    if (!e) return hipErrorInvalidValue;
    MarchArgs g{};
    g.A = a.data;
    g.B = b.data;
    g.bias = reinterpret_cast<const int32_t *>(bias.data);
    g.pitch_bi = bias.pitch;
    g.pad_bi = bias.pad;
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
    g.b_lo = c.b_lo;
    g.b_hi = c.b_hi;
    g.tag_bits = tag_bits_for(c);
    g.ox0 = c.ox0;
    g.ox1 = c.ox1;
    g.oy0 = c.oy0;
    g.oy1 = c.oy1;
    g.strip_rows = m.strip_rows;
    g.tiles = m.tiles;
    g.strips = m.strips;
    g.prefer_large = c.prefer_large;
    g.mirror = c.mirror;
    g.fallback_neg = c.fallback_neg;
    if (m.lds_bytes > 48 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(e->fn),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)m.lds_bytes);
        if (err != hipSuccess) return err;
    }
    dim3 grid(round_up(m.tiles * m.strips, 8));
    g.keys = keys;
    g.keys_pitch = keys_pitch;
    for (int pass = 0; pass < m.passes; ++pass) {
        g.d_first = c.d_lo + pass * m.nch * m.nd_per_thread;
        g.pass_mode = m.passes == 1 ? 0 : pass == 0 ? 1 : pass == m.passes - 1 ? 3 : 2;
        hipLaunchKernelGGL(e->fn, grid, dim3(m.threads), m.lds_bytes, s, g);
    }
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

__global__ void __launch_bounds__(256) ws_generic_kernel(const GenericArgs g)
{
    const int ow = g.view == 0 ? g.w1 : g.w2, oh = g.view == 0 ? g.h1 : g.h2;
    int x, y;
    if (!ring_pixel(g, ow, oh, (long long)blockIdx.x * blockDim.x + threadIdx.x, &x, &y)) return;
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
    g.out[(size_t)y * g.out_pitch + x] = val;
}

// ------------------------------------------------------------------------------------------
// right-view border ring: clipped windows (BlockSearch.cpp:116-123), one wave per pixel
//
// The marching kernel owns the pixels whose (bs-1)^2 window is complete.  On the ring the window
// is clipped by the image border, so its size changes from pixel to pixel; there are only
// ~2*half*(W+H) such pixels.  One wavefront takes one pixel: the 64 lanes take 64 consecutive
// disparities at a time (consecutive lanes read consecutive target pixels), every lane sums its
// own window on the packed planes, keeps its best candidate, and a wave-wide min over
// (cost, d) keys -- butterfly of shuffles -- picks the winner with the reference's tie rule.
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
    int out_pitch;
};

__global__ void __launch_bounds__(256) ws_ring_kernel(const RingArgs g)
{
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    GenericArgs e{}; // only the skip rectangle is used by ring_pixel
    e.skip_x0 = g.skip_x0; e.skip_x1 = g.skip_x1; e.skip_y0 = g.skip_y0; e.skip_y1 = g.skip_y1;
    int x, y;
    if (!ring_pixel(e, g.wa, g.ha, pix, &x, &y)) return; // uniform per wave
    const int xm = g.wa - 1 - x; // canonical (mirrored) column
    float val = 0.0f;
    const uint32_t black = g.centred ? kCentre : 0u;
    if (y < g.height && g.A[(size_t)y * g.pitch_a + xm + g.pad_a] != black) {
        const int left = min(x, g.half), right = min(g.wa - x - 1, g.half);
        const int up = min(y, g.half), down = min(g.ha - y - 1, g.half);
        const int ww = left + right, wh = up + down;
        // candidates: d_lo <= d <= d_hi with x + d + right < w1 (BlockSearch.cpp:147-149)
        const int d_end = min(g.d_hi, g.wb - right - x - 1);
        long long best = LLONG_MAX;
        if (ww > 0 && wh > 0) {
            const int c0 = xm - right + 1; // first window column, canonical
            const uint32_t *arow0 = g.A + (size_t)(y - up) * g.pitch_a + c0 + g.pad_a;
            for (int d = g.d_lo + lane; d <= d_end; d += 64) {
                const uint32_t *brow0 = g.B + (size_t)(y - up) * g.pitch_b + (c0 - d + g.boff + g.pad_b);
                int32_t cost = 0;
                for (int r = 0; r < wh; ++r) {
                    const uint32_t *pa = arow0 + (size_t)r * g.pitch_a;
                    const uint32_t *pb = brow0 + (size_t)r * g.pitch_b;
                    if (!g.ssd) {
                        uint32_t acc = 0;
                        for (int i = 0; i < ww; ++i) acc = pix_sad(pa[i], pb[i], acc);
                        cost += (int32_t)acc;
                    } else if (g.centred) {
                        uint32_t bb = 0, ab = 0; // sum (a-b)^2 = sum a^2 + [sum b^2 - 2 sum ab]; sum a^2 is the same for every d
                        for (int i = 0; i < ww; ++i) {
                            bb = pix_dot<true>(pb[i], pb[i], bb);
                            ab = pix_dot<true>(pa[i], pb[i], ab);
                        }
                        cost += (int32_t)bb - 2 * (int32_t)ab;
                    } else {
                        uint32_t bb = 0, ab = 0;
                        for (int i = 0; i < ww; ++i) {
                            bb = pix_dot<false>(pb[i], pb[i], bb);
                            ab = pix_dot<false>(pa[i], pb[i], ab);
                        }
                        cost += (int32_t)bb - 2 * (int32_t)ab;
                    }
                }
                const long long key = ((long long)cost << 32) | (uint32_t)d; // ties: smaller d
                best = min(best, key);
            }
        }
        // wave-wide min: butterfly over the 64 lanes
        for (int off = 32; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
        val = best == LLONG_MAX ? -(float)x : (float)(uint32_t)(best & 0xffffffffll);
    }
    if (lane == 0) g.out[(size_t)y * g.out_pitch + x] = val;
}

hipError_t launch_ring(const Canon &c, Plane a, Plane b, const GenericArgs &skip, float *out, int out_pitch,
                       hipStream_t s)
{
    RingArgs g{};
    g.A = a.data; g.B = b.data;
    g.pitch_a = a.pitch; g.pad_a = a.pad; g.pitch_b = b.pitch; g.pad_b = b.pad;
    g.wa = c.wa; g.ha = c.ha; g.wb = c.wb;
    g.height = std::min(c.ha, c.hb);
    g.half = c.wh / 2; // right view: window (bs-1)^2 = (2*half)^2
    g.boff = c.boff; g.d_lo = c.d_lo; g.d_hi = c.d_hi;
    g.ssd = c.ssd; g.centred = march_centred(c);
    g.skip_x0 = skip.skip_x0; g.skip_x1 = skip.skip_x1; g.skip_y0 = skip.skip_y0; g.skip_y1 = skip.skip_y1;
    g.out = out; g.out_pitch = out_pitch;
    const long long inside = (long long)(g.skip_x1 - g.skip_x0) * (g.skip_y1 - g.skip_y0);
    const long long n = (long long)c.wa * c.ha - (inside > 0 ? inside : 0);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ws_ring_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, g);
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
            } else if (!(g.max_d > 1 && x + 1 + right < g.w1)) {
                val = 0.0f; // d = 0 is the only candidate
            } else {
                const uint8_t *rw = g.R + (size_t)(y - up) * g.s2 + 3 * (x - left);
                const int d1 = (int)*o;
                const unsigned long long c0 = window_cost64(g.L + (size_t)(y - up) * g.s1 + 3 * (x - left), g.s1, rw, g.s2, ww, wh, g.ssd);
                const unsigned long long c1 = window_cost64(g.L + (size_t)(y - up) * g.s1 + 3 * (x + d1 - left), g.s1, rw, g.s2, ww, wh, g.ssd);
                const double area = (double)(ww * wh);
                const double e1 = (g.ssd ? sqrt((double)c1) : (double)c1) / area;
                double e0 = (g.ssd ? sqrt((double)c0) : (double)c0) / area;
                code = 0;
                if (e1 < e0) code |= 1;
                e0 *= s;
                if (e1 < e0) code |= 2;
                e0 *= s;
                if (e1 < e0) code |= 4;
                val = (float)d1;
            }
        }
    } else { // LinearSearch: black test on the left pixel, distance of single pixels
        if (y < g.h1 && !(x < g.w1 && black3(g.L + (size_t)y * g.s1 + 3 * x))) {
            if (!(x < g.w1)) {
                val = -(float)x;
            } else if (!(g.linear_range > 1 && x + 1 < g.w1)) {
                val = 0.0f;
            } else {
                const uint8_t *pr = g.R + (size_t)y * g.s2 + 3 * x;
                const int d1 = (int)*o;
                const uint32_t c0 = window_cost(pr, 0, g.L + (size_t)y * g.s1 + 3 * x, 0, 1, 1, 1);
                const uint32_t c1 = window_cost(pr, 0, g.L + (size_t)y * g.s1 + 3 * (x + d1), 0, 1, 1, 1);
                const double e1 = sqrt((double)c1);
                double e0 = sqrt((double)c0);
                code = 0;
                if (e1 < e0) code |= 1;
                e0 *= s;
                if (e1 < e0) code |= 2;
                e0 *= s;
                if (e1 < e0) code |= 4;
                val = (float)d1;
            }
        }
    }
    *o = val;
    if ((code & kSelFixed) && val == 0.0f) code |= kSelZero;
    sel[(size_t)y * sel_pitch + x] = code;
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
// smoothFactor in [0,1) for the LEFT view (BlockSearch.cpp:68-73): there the factor reaches any
// candidate d that equals the upper / left neighbour's stored value -- a true dependency on the
// neighbours' values in raster order.  With 0 <= s <= 1 a discounted candidate only gets cheaper,
// so the winner is always one of { d1 = the undiscounted argmin, up, left } (every other d is no
// better than d1 and loses the tie by the reference's own rule).  So: the ordinary data-parallel
// search gives d1; one workgroup then walks the rows in order and, inside a row, iterates
//     v_x <- F_x(v_{x-1})          F_x(l) = lexmin over {d1, up, l} of (distance * s^matches, -d)
// from the left-independent guess until nothing changes (the fixed point is the sequential
// result; the iteration count is the longest run a left neighbour's value actually propagates).
// ------------------------------------------------------------------------------------------
struct SmoothLeftArgs {
    const uint8_t *L;
    const uint8_t *R;
    int w1, h1, s1, w2, h2, s2;
    int block_size, max_d, ssd;
    double s;
    float *out; // holds d1 on entry, the final map on exit
    int out_pitch;
};

__device__ __forceinline__ bool left_candidate_ok(const SmoothLeftArgs &g, int x, int d, int half)
{
    return d >= 1 && d <= g.max_d && x - d >= half && x - d < g.w2 - half;
}

__device__ __forceinline__ double left_dist(const SmoothLeftArgs &g, int x, int y, int d, int half)
{
    const uint8_t *lw = g.L + (size_t)(y - half) * g.s1 + 3 * (x - half);
    const uint8_t *rw = g.R + (size_t)(y - half) * g.s2 + 3 * (x - d - half);
    const uint32_t c = window_cost(lw, g.s1, rw, g.s2, g.block_size, g.block_size, g.ssd);
    return g.ssd ? sqrt((double)c) : (double)c;
}

// candidate (dist, d) beats (bd, bdist) in the reference's iteration (d descending, strict <)
__device__ __forceinline__ bool left_better(double dist, int d, double bdist, int bd)
{
    return dist < bdist || (dist == bdist && d > bd);
}

constexpr int kSmoothLeftPer = 4; // columns per thread: images up to 4096 wide

__global__ void __launch_bounds__(1024) ws_smooth_left_kernel(const SmoothLeftArgs g)
{
    extern __shared__ float sl_rows[]; // [3][w1]: previous row, current guess, next guess
    __shared__ int changed;
    float *prev = sl_rows, *cur = sl_rows + g.w1, *nxt = sl_rows + 2 * g.w1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int half = (g.block_size - 1) / 2;
    const int height = min(g.h1, g.h2);
    for (int x = tid; x < g.w1; x += nt) prev[x] = 0.0f; // row above the first interior row: border zeros
    if (half > 0) // (for half == 0 the first row has no upper neighbour at all: zeros never match d >= 1)
        for (int x = tid; x < g.w1; x += nt) prev[x] = g.out[(size_t)(half - 1) * g.out_pitch + x];
    __syncthreads();
    for (int y = half; y < height - half; ++y) {
        float *orow = g.out + (size_t)y * g.out_pitch;
        // per column: the fixed ingredients of F_x
        int d1[kSmoothLeftPer], du[kSmoothLeftPer], lastl[kSmoothLeftPer];
        double u1[kSmoothLeftPer], uu[kSmoothLeftPer], ul[kSmoothLeftPer];
        bool act[kSmoothLeftPer];
#pragma unroll
        for (int k = 0; k < kSmoothLeftPer; ++k) {
            const int x = tid + k * nt;
            act[k] = false;
            d1[k] = du[k] = 0; lastl[k] = -1;
            u1[k] = uu[k] = ul[k] = 0.0;
            if (x < g.w1) {
                const float v = orow[x];
                cur[x] = v;
                if (x >= half && x < g.w1 - half && !black3(g.L + (size_t)y * g.s1 + 3 * x)) {
                    const int d = (int)v;
                    if (left_candidate_ok(g, x, d, half)) { // otherwise: no candidate at all, value x stays
                        act[k] = true;
                        d1[k] = d;
                        u1[k] = left_dist(g, x, y, d, half);
                        const int up = (int)prev[x];
                        const bool up_int = (float)up == prev[x];
                        if (y >= 1 && up_int && up != d && left_candidate_ok(g, x, up, half)) {
                            du[k] = up;
                            uu[k] = left_dist(g, x, y, up, half) * g.s;
                        } else if (y >= 1 && up_int && up == d) {
                            du[k] = -1; // d1 itself is the upper neighbour's value
                            u1[k] *= g.s;
                        }
                        // the guess without a left neighbour
                        float gx = (float)d;
                        if (du[k] > 0 && left_better(uu[k], du[k], u1[k], d)) gx = (float)du[k];
                        cur[x] = gx;
                    }
                }
            }
        }
        __syncthreads();
        for (int it = 0; it < g.w1 + 1; ++it) {
            if (tid == 0) changed = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kSmoothLeftPer; ++k) {
                const int x = tid + k * nt;
                if (x >= g.w1) continue;
                float res = cur[x];
                if (act[k]) {
                    // start from d1 (its upper-neighbour factor already in u1), then up, then left
                    double bdist = u1[k];
                    int bd = d1[k];
                    const float lf = x >= 1 ? cur[x - 1] : 0.0f;
                    const int l = (int)lf;
                    const bool l_ok = x >= 1 && (float)l == lf && left_candidate_ok(g, x, l, half);
                    if (l_ok && l == d1[k]) bdist = bdist * g.s; // the left factor comes second (BlockSearch.cpp:71-73)
                    if (du[k] > 0) {
                        double e = uu[k];
                        if (l_ok && l == du[k]) e = e * g.s;
                        if (left_better(e, du[k], bdist, bd)) { bdist = e; bd = du[k]; }
                    }
                    if (l_ok && l != d1[k] && l != du[k]) {
                        if (lastl[k] != l) { // distance at the left neighbour's value: cached per column
                            ul[k] = left_dist(g, x, y, l, half);
                            lastl[k] = l;
                        }
                        const double e = ul[k] * g.s;
                        if (left_better(e, l, bdist, bd)) { bdist = e; bd = l; }
                    }
                    res = (float)bd;
                }
                nxt[x] = res;
                if (res != cur[x]) changed = 1;
            }
            __syncthreads();
            float *t = cur; cur = nxt; nxt = t;
            const int any = changed;
            __syncthreads();
            if (!any) break;
        }
        for (int x = tid; x < g.w1; x += nt) {
            orow[x] = cur[x];
        }
        __syncthreads();
        { float *t = prev; prev = cur; cur = t; }
    }
}

hipError_t launch_smooth_left(const GenericArgs &g, double s, hipStream_t st)
{
    SmoothLeftArgs a{};
    a.L = g.L; a.R = g.R; a.w1 = g.w1; a.h1 = g.h1; a.s1 = g.s1; a.w2 = g.w2; a.h2 = g.h2; a.s2 = g.s2;
    a.block_size = g.block_size; a.max_d = g.max_d; a.ssd = g.ssd; a.s = s;
    a.out = g.out; a.out_pitch = g.out_pitch;
    if (g.w1 > kSmoothLeftPer * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ws_smooth_left_kernel, dim3(1), dim3(1024), (size_t)3 * g.w1 * sizeof(float), st, a);
    return hipGetLastError();
}

// ---- bit-parallel form for 0 <= smoothFactor <= 1 -------------------------------------------
// There c0 * s^k does not grow with k, so t_0 >= t_1 >= t_2 and a pixel is one of: always 0
// ("generate"), never 0 ("kill"), or 0 exactly when its left neighbour is ("propagate") -- a
// carry chain.  With the codes packed into bit planes (64 columns per word) one lane resolves 64
// columns with a single 64-bit addition (A = g|p, B = g: the carries of A+B are the chain), and
// the carries between the lanes' words come from the same addition on two ballot masks in the
// scalar unit.  ~40 bit operations per row for the whole image width.
__global__ void __launch_bounds__(256) ws_smooth_planes_kernel(const uint8_t *__restrict__ sel, int sel_pitch, int w,
                                                               unsigned long long *__restrict__ planes, int nwp)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const uint32_t c = x < w ? sel[(size_t)y * sel_pitch + x] : kSelFixed; // beyond the row: fixed, non-zero
    const bool fixed = c & kSelFixed;
    const unsigned long long t0 = __ballot(!fixed && (c & 1)), t1 = __ballot(!fixed && (c & 2)),
                             t2 = __ballot(!fixed && (c & 4)), fx = __ballot(fixed),
                             zf = __ballot(fixed && (c & kSelZero));
    if ((threadIdx.x & 63) == 0 && (x >> 6) < nwp) {
        unsigned long long *row = planes + (size_t)y * 5 * nwp + (x >> 6);
        row[0] = t0; row[nwp] = t1; row[2 * nwp] = t2; row[3 * nwp] = fx; row[4 * nwp] = zf;
    }
}

__global__ void __launch_bounds__(64) ws_smooth_resolve_bits_kernel(const unsigned long long *__restrict__ planes, int nwp,
                                                                    int rows, unsigned long long *__restrict__ zplane,
                                                                    int chunk_rows)
{
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    extern __shared__ uint4 ws_smem4[];
    uint8_t *lds = reinterpret_cast<uint8_t *>(ws_smem4);
    const int lane = threadIdx.x;
    const int row_bytes = 5 * nwp * 8; // nwp is even: a multiple of 16 bytes
    const int chunk_bytes = chunk_rows * row_bytes;
    const int nchunks = (rows + chunk_rows - 1) / chunk_rows;
    const uint8_t *src0 = reinterpret_cast<const uint8_t *>(planes);
    for (int o = lane * 16; o < chunk_bytes; o += 1024)
        __builtin_amdgcn_global_load_lds((glb_void *)(src0 + o), (lds_void *)(lds + (o - lane * 16)), 16, 0, 0);
    unsigned long long zprev = 0;
    const bool active = lane < nwp;
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint8_t *cur = lds + (c & 1) * chunk_bytes;
        if (c + 1 < nchunks) {
            const uint8_t *src = src0 + (size_t)(c + 1) * chunk_bytes;
            uint8_t *dst = lds + ((c + 1) & 1) * chunk_bytes;
            for (int o = lane * 16; o < chunk_bytes; o += 1024)
                __builtin_amdgcn_global_load_lds((glb_void *)(src + o), (lds_void *)(dst + (o - lane * 16)), 16, 0, 0);
        }
        const int y_end = min((c + 1) * chunk_rows, rows);
        for (int y = c * chunk_rows; y < y_end; ++y) {
            const unsigned long long *row = reinterpret_cast<const unsigned long long *>(cur + (size_t)(y - c * chunk_rows) * row_bytes);
            unsigned long long g = 0, p = 0;
            if (active) {
                const unsigned long long t0 = row[lane], t1 = row[nwp + lane], t2 = row[2 * nwp + lane],
                                         fx = row[3 * nwp + lane], zf = row[4 * nwp + lane];
                // zero when the left neighbour is not / is zero, given the upper neighbour's flag
                unsigned long long n0 = ~((t0 & ~zprev) | (t1 & zprev)), n1 = ~((t1 & ~zprev) | (t2 & zprev));
                n0 = (n0 & ~fx) | zf;
                n1 = (n1 & ~fx) | zf;
                g = n0 & n1;
                p = n1 & ~n0;
            }
            const unsigned long long A = g | p, B = g, S0 = A + B;
            const bool cout0 = S0 < A, cout1 = cout0 || S0 == ~0ull;
            // carries between the lanes' words: the same adder on the ballots (scalar unit)
            const unsigned long long gw = __ballot(cout0), pw = __ballot(cout1 && !cout0);
            const unsigned long long aw = gw | pw, sw = aw + gw, cw = sw ^ aw ^ gw; // bit l = carry into lane l
            const unsigned long long cin = (cw >> lane) & 1ull; // column 0 of the row: no left neighbour
            const unsigned long long S = S0 + cin;
            const unsigned long long carries = S ^ A ^ B; // bit k = carry into column k
            const unsigned long long cout = cin ? (unsigned long long)cout1 : (unsigned long long)cout0;
            const unsigned long long z = (carries >> 1) | (cout << 63);
            if (active) zplane[(size_t)y * nwp + lane] = z;
            zprev = z;
        }
    }
}

__global__ void __launch_bounds__(256) ws_smooth_apply_kernel(float *out, int out_pitch, int w, int rows,
                                                              const uint8_t *__restrict__ sel, int sel_pitch,
                                                              const unsigned long long *__restrict__ zplane, int nwp)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= rows) return;
    if (sel[(size_t)y * sel_pitch + x] & kSelFixed) return;
    if ((zplane[(size_t)y * nwp + (x >> 6)] >> (x & 63)) & 1ull) out[(size_t)y * out_pitch + x] = 0.0f;
}

int smooth_sel_rows(int rows) { return (rows + 15) / 16 * 16 + 16; }

size_t smooth_planes_bytes(int w, int h)
{
    const int nwp = round_up(ceil_div(w, 64), 2);
    return (size_t)(h + 64) * 6 * nwp * 8; // 5 code planes + the resolved plane
}

hipError_t launch_smooth(const GenericArgs &g, double s, uint8_t *sel, int sel_pitch, unsigned long long *planes,
                         hipStream_t st)
{
    dim3 grid(ceil_div(g.w2, 256), g.h2);
    hipLaunchKernelGGL(ws_smooth_prepare_kernel, grid, dim3(256), 0, st, g, s, sel, sel_pitch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int rows = std::min(g.h1, g.h2);
    const int nwords = ceil_div(g.w2, 64);
    if (s >= 0.0 && s <= 1.0 && nwords <= 64 && planes) {
        const int nwp = round_up(nwords, 2);
        unsigned long long *zplane = planes + (size_t)(g.h2 + 64) * 5 * nwp;
        hipLaunchKernelGGL(ws_smooth_planes_kernel, dim3(ceil_div(g.w2, 256), rows), dim3(256), 0, st, sel, sel_pitch,
                           g.w2, planes, nwp);
        int chunk = 24576 / (5 * nwp * 8);
        if (chunk > 64) chunk = 64;
        if (chunk < 1) chunk = 1;
        hipLaunchKernelGGL(ws_smooth_resolve_bits_kernel, dim3(1), dim3(64), (size_t)2 * chunk * 5 * nwp * 8, st, planes,
                           nwp, rows, zplane, chunk);
        hipLaunchKernelGGL(ws_smooth_apply_kernel, dim3(ceil_div(g.w2, 256), rows), dim3(256), 0, st, g.out, g.out_pitch,
                           g.w2, rows, sel, sel_pitch, zplane, nwp);
        return hipGetLastError();
    }
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

// ------------------------------------------------------------------------------------------
// varBlock (BlockSearch.cpp:125-145, right view): while the window's centred norm is below
// `thres` the block grows by 4; then the search runs with that pixel's own window.  Windows differ
// from pixel to pixel, so no sliding sums: one wavefront per pixel, lanes share the window pixels
// for the texture test and split the disparities for the search; wave-wide sums / min by shuffles.
// cv::mean / cv::subtract / cv::norm semantics (OpenCV 4.x restated; the library is
// un-vendored): double mean per channel, saturate_cast<uchar>(round-half-even(p - mean)), L2 norm.
// Growth stops when the window no longer changes (the reference would loop forever there).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

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
                unsigned long long acc = 0;
                for (int i = lane; i < n; i += 64) {
                    const uint8_t *p = w0 + (size_t)(i / ww) * g.s2 + 3 * (i % ww);
                    const double v0 = fmin(255.0, fmax(0.0, rint((double)p[0] - m0)));
                    const double v1 = fmin(255.0, fmax(0.0, rint((double)p[1] - m1)));
                    const double v2 = fmin(255.0, fmax(0.0, rint((double)p[2] - m2)));
                    acc += (unsigned long long)(v0 * v0) + (unsigned long long)(v1 * v1) + (unsigned long long)(v2 * v2);
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
