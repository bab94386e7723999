// ws_march.hip -- the hot kernel of the WindowSearch path on gfx950 (CDNA4), plus the overview of
// all device code:  ws_prepass.hip (pack + bias + border in one launch), ws_march.hip (marching kernel, tiling plan),
// ws_border.hip (brute force, border ring, refine, varBlock), ws_smooth.hip (smoothFactor passes),
// ws_consumers.hip (warp, Reconstruction-side maps), ws_device.h (shared helpers).
//
// What the reference computes (BlockSearch.cpp:24-179): for every pixel, for every candidate
// disparity, the L2 norm of the absolute difference of two bs x bs x 3 windows, and the
// candidate with the strictly smallest value.  It re-sums the window for every (pixel, d).
//
// What runs here instead (same integers, same winner):
//   ws_prepare_kernel  one launch of independent workgroups before the hot kernel:
//                    pack: BGR bytes -> one dword per pixel (B | G<<8 | R<<16), zero padded plane,
//                          mirrored in x for the right view;
//                    bias: per (row, B column) the validity poison, and for SSD the box sum of the
//                          squared target pixels (the part of sum (a-b)^2 that does not need a);
//                    left view: the pixels outside the marching interior (border zeros).
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
//   ws_ring_kernel   right-view border ring (clipped windows): a small marching kernel, lanes over d.
//   ws_linear_kernel LinearSearch through LDS.
//   ws_generic_kernel  literal per-pixel brute force: window sizes without a marching instantiation
//                    and LinearSearch ranges beyond 4096.
//   ws_refine_*      sub-pixel parabola (extension).
//
// No MFMA: the hot loop is a stencil + reduction on bytes, bounded by VALU issue and LDS, see
// DESIGN.md.  Wave64 throughout; nothing here assumes 32-wide warps.
#include "ws_device.h"

#include <mutex>
#include <utility>
#include <vector>

namespace wsamd {

// ------------------------------------------------------------------------------------------
// the marching kernel
// ------------------------------------------------------------------------------------------
struct MarchArgs {
    const uint32_t *A;
    const uint32_t *B;
    const int32_t *bias; // SSD only
    float *out;
    double *out64; // if set: doubles here instead of floats to `out` (CV_64F output without a widening pass)
    int pitch_a, pad_a, pitch_b, pad_b, pitch_bi, pad_bi, out_pitch;
    int wa;
    int nxr, nch;
    int wx0, wy0, boff;
    int d_lo, d_hi, b_lo, b_hi;
    int d_top;     // d_lo + passes * chunks * ND - 1: the padded upper end of the range (SSD tie tags count from it)
    int d_first;   // first disparity of chunk 0 in THIS launch (d_lo + pass * chunks * ND)
    int pass_mode; // 0 = the only pass, 1 = first, 2 = middle, 3 = last of several d-group passes
    void *keys;    // several passes: plane of the best keys so far (slot_t per pixel)
    int keys_pitch;
    int ox0, ox1, oy0, oy1;
    int strip_rows, tiles, strips;
    int prefer_large, mirror, fallback_neg;
    int tag_bits; // SAD: keys are (cost << tag_bits) | global tie tag
    int32_t *cost_out; // optional (smoothFactor passes): the winner's cost, SSD without the sum of a^2
    int cost_pitch;
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
// layout: global_load_lds_dwordx4, no VGPR staging.  The LDS address of an LDS-DMA is wave-uniform
// base (M0) + lane * 16, so consecutive lanes fill consecutive quads of one region and each lane
// fetches the quad that belongs there (the source address carries the permutation).
//
// The instruction is issued through inline assembly ON PURPOSE: for the builtin the compiler makes
// every later LDS read of the wave wait for vmcnt(0) (it cannot know the copy fills a ring slot nobody
// reads in this step), which exposes the copy's whole latency at the top of the arithmetic; here
// nothing waits until the explicit dma_wait() in front of the step's barrier (A/B on one MI355X,
// config 2: 171 -> 166 us).  Dealing the copies of a step to different waves, with scalar addressing,
// measured SLOWER (186-191 us): the loop below leaves them all to the workgroup's first wave.
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NREG>
__device__ __forceinline__ void stage_row_async(uint32_t *row, int ro, const uint32_t *gsrc, int n, int tid, int nt)
{
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const int lane = tid & 63;
    const int nquads = (n + 3) >> 2;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const int nidx = (nquads - j + NREG - 1) / NREG; // quads of this region
        for (int idx = tid; idx < nidx; idx += nt) {
            // (M0 is written right in front of its use: nothing of the compiler's can sit between the two)
            const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u32 *)(row + j * ro + 4 * (idx - lane)));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :
                         : "v"(gsrc + 4 * (idx * NREG + j)), "s"(la)
                         : "memory");
        }
    }
}

// One row entering (SIGN=+1) or leaving (SIGN=-1) the window of every (column, disparity) this
// thread owns.
//   SAD: V = (window sum << shift) + global tie tag          key = V
//   SSD: V = local tie tag - (2 * cross sum << LT)           key = bias[xb] + V
//        (bias = box sum of the squared target pixels << LT, or poison for an invalid centre)
// With KEY the candidate keys are folded into best[] (signed min; equal costs go to the smaller
// tag, i.e. to the disparity the reference's strict '<' keeps).
template <int X, int ND, int WW, bool SSD, bool KEY>
__device__ __forceinline__ void march_load(uint32_t (&pa)[X + WW - 1], uint32_t (&pb)[X + WW + ND - 2], uint32_t (&bi)[X + ND - 1],
                                           const uint32_t *runA, int ro_a, const uint32_t *runB, int ro_b,
                                           const int32_t *runBias, int ro_bi)
{
    constexpr int NREG = X / 4, NREGB = march_nreg_b(X, ND);
    lds_run<X + WW - 1, NREG>(pa, runA, ro_a);
    lds_run<X + WW + ND - 2, NREGB>(pb, runB, ro_b);
    if constexpr (KEY && SSD) lds_run<X + ND - 1, NREGB>(bi, reinterpret_cast<const uint32_t *>(runBias), ro_bi);
}

template <int X, int ND, int WW, bool SSD, bool CENTRED, int SIGN, bool KEY>
__device__ __forceinline__ void march_compute(int32_t (&V)[X][ND], int32_t (&best)[X], const uint32_t (&pa)[X + WW - 1],
                                              const uint32_t (&pb)[X + WW + ND - 2], const uint32_t (&bi)[X + ND - 1], int shift)
{
    constexpr int NA = X + WW - 1;
    // SAD accumulates +cost, SSD accumulates -2*cross: flip the sign of the update for SSD
    constexpr bool ADD = ((SIGN > 0) != SSD);
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

template <int X, int ND, int WW, bool SSD, bool CENTRED, int SIGN, bool KEY>
__device__ __forceinline__ void march_row(int32_t (&V)[X][ND], int32_t (&best)[X],
                                          const uint32_t *runA, int ro_a, const uint32_t *runB,
                                          int ro_b, const int32_t *runBias, int ro_bi, int shift)
{
    uint32_t pa[X + WW - 1], pb[X + WW + ND - 2], bi[X + ND - 1];
    march_load<X, ND, WW, SSD, KEY>(pa, pb, bi, runA, ro_a, runB, ro_b, runBias, ro_bi);
    march_compute<X, ND, WW, SSD, CENTRED, SIGN, KEY>(V, best, pa, pb, bi, shift);
}

template <int X, int ND, int WW, int WH, bool SSD, int MAXT, bool COST = false>
__global__ void __launch_bounds__(MAXT) ws_march_kernel(const MarchArgs g)
{
    static_assert(X % 4 == 0 && ND % 4 == 0, "runs start on 16-byte quads");
    constexpr int NREG = X / 4, NREGB = march_nreg_b(X, ND);
    constexpr int LT = ilog2c(ND);
    constexpr bool CENTRED = SSD && ssd_needs_centring(WW, WH, ND);
    constexpr int NR = WH + 2; // ring rows: WH+1 in use by a step, 1 being filled for the next
    // merge slots: SSD (cost << LT | 7) : global tie tag as one signed 64-bit key, SAD the 32-bit key itself;
    // a key at or above kValidKeyBound (in its cost word) is "no valid candidate"
    typedef typename std::conditional<SSD, long long, int32_t>::type slot_t;
    const slot_t kEmpty = SSD ? (slot_t)LLONG_MAX : (slot_t)INT_MAX;

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
    // SSD merge: global tie tag = chunk tag | local tag (a multiple of ND, so one v_and_or builds it):
    // the chunk's distance from the preferred end of the padded range [d_lo, d_top]
    const int ctag = g.prefer_large ? g.d_top - d0 - (ND - 1) : d0 - g.d_lo;

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
    dma_wait();
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
            // (the workgroup's first wave issues the row copies: the flush goes to its LAST waves; tx <= NT)
            const int k = tid - (NT - round_up_dev(tx, 64));
            if (k >= 0 && k < tx) {
                const int si = (k % X) * g.nxr + k / X; // slots are stored [x][r]
                slot_t key = sl[si];
                sl[si] = kEmpty;
                const int x = tile_x0 + k;
                if (x < g.ox1 && g.pass_mode != 0) {
                    // disparity ranges too wide for one tile run as several d-group passes that
                    // meet in a plane of keys (same keys, same ordering: min is the merge)
                    slot_t *kp = static_cast<slot_t *>(g.keys) + (size_t)y * g.keys_pitch + x;
                    if (g.pass_mode != 1) key = min(key, *kp);
                    if (g.pass_mode != 3) *kp = key;
                }
                if (x < g.ox1 && (g.pass_mode == 0 || g.pass_mode == 3)) {
                    const int xo = g.mirror ? g.wa - 1 - x : x;
                    float val;
                    const bool none = SSD ? (int32_t)((long long)key >> 32) >= kValidKeyBound : (int32_t)key >= kValidKeyBound;
                    if (none) {
                        val = g.fallback_neg ? -(float)xo : (float)xo;
                    } else {
                        const int gtag = SSD ? (int)(uint32_t)key : ((int)key & ((1 << g.tag_bits) - 1));
                        val = (float)(g.prefer_large ? (SSD ? g.d_top : g.d_hi) - gtag : g.d_lo + gtag);
                    }
                    // black pixel (BlockSearch.cpp:41, :105): image row y, column x, from the ring
                    if (rowA[lds_phys<NREG>(k - g.wx0, ro_a)] == (CENTRED ? kCentre : 0u)) val = 0.0f;
                    if (g.out64) g.out64[(size_t)y * g.out_pitch + xo] = (double)val;
                    else g.out[(size_t)y * g.out_pitch + xo] = val;
                    if (COST && !none) { // (a template flag: the test alone cost the hot kernel 2.7 %)
                        int32_t cst;
                        if constexpr (SSD) cst = (int32_t)((long long)key >> 32) >> LT;
                        else cst = (int32_t)key >> g.tag_bits;
                        g.cost_out[(size_t)y * g.cost_pitch + xo] = cst;
                    }
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
                    // no validity test here: a poisoned key is just a large one, the flush sorts it out
                    if constexpr (SSD) {
                        const uint32_t gtag = (uint32_t)(bk & (ND - 1)) | (uint32_t)ctag; // v_and_or_b32
                        const long long key = (long long)(((unsigned long long)(uint32_t)(bk | (ND - 1)) << 32) | gtag);
                        atomicMin(sl + x * g.nxr, key); // ds_min_i64, lanes on consecutive slots
                    } else {
                        atomicMin(sl + x * g.nxr, bk); // ds_min_i32
                    }
                }
            } else {
                march_row<X, ND, WW, SSD, CENTRED, +1, false>(V, best, ringA + add_slot * a_w + ia, ro_a,
                                                     ringB + add_slot * b_w + ib, ro_b, nullptr, 0, shift);
            }
        }

        dma_wait(); // the row copies issued at the top of this step have long landed
        __syncthreads();
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
constexpr int kNDNarrow = 4;                           // the second instantiation of every window (march_nd)
constexpr int kMaxChunks = 64; // at most 512 disparities per tile and pass (tools/time_calls.py)
constexpr int kMinXRuns = 4; // narrowest tile: 4 x-runs = 32 columns (D up to 1536)

typedef void (*MarchFn)(const MarchArgs);
struct MarchEntry {
    int ww, wh, ssd, nd;
    MarchFn fn;
    MarchFn fn_cost; // the same kernel also writing the winners' costs (right-view window sizes only)
    const char *name;
};
// Every window comes with 8 and with 4 disparities per thread.  8 (64 running sums, ~180 VGPRs, two waves per
// SIMD) is the fastest search of a config-2-like pair on an idle chip; 4 (~125 VGPRs, four waves per SIMD) reads
// more of plane B per hypothesis but lets the workgroups of two searches -- two contexts taking a queue of
// pairs alternately, INTEGRATION.md -- share a CU, and is level or ahead on its own for wide windows and wide
// disparity ranges (tools/two_in_flight.py on both builds: 17x17 D=200 0.143 -> 0.134 ms alone, config 3
// 0.96 -> 0.89 ms and config 5's search 3.20 -> 2.93 ms with two in flight; 7x7 D=256 0.176 -> 0.204 alone).
#define WS_MARCH_ENTRY_ND(W, H, N, TAG)                                                                          \
    {W, H, 0, N, ws_march_kernel<kX, N, W, H, false, kMaxT>, nullptr, "ws_march_kernel<sad," #W "x" #H TAG ">"}, \
    {W, H, 1, N, ws_march_kernel<kX, N, W, H, true, kMaxT>, nullptr, "ws_march_kernel<ssd," #W "x" #H TAG ">"}
#define WS_MARCH_ENTRY_COST_ND(W, H, N, TAG)                                                                   \
    {W, H, 0, N, ws_march_kernel<kX, N, W, H, false, kMaxT>, ws_march_kernel<kX, N, W, H, false, kMaxT, true>, \
     "ws_march_kernel<sad," #W "x" #H TAG ">"},                                                                \
    {W, H, 1, N, ws_march_kernel<kX, N, W, H, true, kMaxT>, ws_march_kernel<kX, N, W, H, true, kMaxT, true>,   \
     "ws_march_kernel<ssd," #W "x" #H TAG ">"}
#define WS_MARCH_ENTRY(W, H) WS_MARCH_ENTRY_ND(W, H, kND, ""), WS_MARCH_ENTRY_ND(W, H, kNDNarrow, ",nd4")
#define WS_MARCH_ENTRY_COST(W, H) WS_MARCH_ENTRY_COST_ND(W, H, kND, ""), WS_MARCH_ENTRY_COST_ND(W, H, kNDNarrow, ",nd4")
static const MarchEntry kMarchTable[] = {
    // left view: bs x bs
    WS_MARCH_ENTRY(3, 3), WS_MARCH_ENTRY(5, 5), WS_MARCH_ENTRY(7, 7), WS_MARCH_ENTRY(9, 9),
    WS_MARCH_ENTRY(11, 11), WS_MARCH_ENTRY(13, 13), WS_MARCH_ENTRY(15, 15), WS_MARCH_ENTRY(17, 17),
    // right view: (bs-1) x (bs-1)
    WS_MARCH_ENTRY_COST(2, 2), WS_MARCH_ENTRY_COST(4, 4), WS_MARCH_ENTRY_COST(6, 6), WS_MARCH_ENTRY_COST(8, 8),
    WS_MARCH_ENTRY_COST(10, 10), WS_MARCH_ENTRY_COST(12, 12), WS_MARCH_ENTRY_COST(14, 14), WS_MARCH_ENTRY_COST(16, 16),
};

// disparities per thread for this search: a function of the canonical problem alone, so that everything that
// reads the marching kernel's planes afterwards (smoothFactor, sub-pixel refine) derives the same key layout
static int march_nd(const Canon &c)
{
    static const int forced = [] {
        const char *e = getenv("WS_MARCH_ND"); // development knob
        const int v = e ? atoi(e) : 0;
        return v == kND || v == kNDNarrow ? v : 0;
    }();
    if (forced) return forced;
    if (kND == kNDNarrow) return kND;
    // The table behind this rule: profiles/r02/nd_grid.txt (tools/two_in_flight.py --grid: windows 5..17, ranges
    // 128 / 256 / 512, both costs at 1500 x 1000, either build alone and with two pairs in flight).
    const int dcount = c.d_hi - c.d_lo + 1;
    if (c.ww <= 9) return !c.ssd || dcount >= 448 ? kNDNarrow : kND; // two workgroups a CU: SAD -8..-15 % in flight;
                                                                      // SSD -8 % from 512 on, +5 % alone at 320..384
    if (c.ww <= 14) return c.ssd ? kNDNarrow : kND;                   // SSD -7..-17 % either way, SAD +15..+27 %
    return c.ssd && dcount <= 224 ? kNDNarrow : kND;                  // 15..17 SSD: level at 1500 x 1000, -4 % (-24 % in
                                                                      // flight, right view) at the reference's 900 x 750,
                                                                      // D = 200; +6..+24 % from 256 on; SAD +8 % at 192
}

static const MarchEntry *find_march(const Canon &c)
{
    const int nd = march_nd(c);
    for (const MarchEntry &e : kMarchTable)
        if (e.ww == c.ww && e.wh == c.wh && e.ssd == c.ssd && e.nd == nd) return &e;
    return nullptr;
}

static int tag_bits_for(const Canon &c)
{
    int bits = 1;
    while ((1 << bits) < c.d_hi - c.d_lo + 1) ++bits;
    return bits;
}

bool march_has_cost(const Canon &c)
{
    const MarchEntry *e = find_march(c);
    return e && e->fn_cost;
}

int march_centred(const Canon &c) { return c.ssd && ssd_needs_centring(c.ww, c.wh, march_nd(c)); }

bool march_supported(const Canon &c)
{
    if (!find_march(c)) return false;
    if (c.ox1 <= c.ox0 || c.oy1 <= c.oy0) return false;
    const int dcount = c.d_hi - c.d_lo + 1;
    if (dcount < 1) return false;
    // keys must stay inside (-2^28, 2^28)
    //   SSD: (2 * cross sum) << log2(ND)        SAD: window sum << tag bits
    const long long worst = c.ssd ? 2LL * c.ww * c.wh * 3 * (march_centred(c) ? 128 * 128 : 255 * 255) * march_nd(c)
                                  : ((long long)c.ww * c.wh * 3 * 255) << tag_bits_for(c);
    return worst < (long long)kValidKeyBound;
}

static int march_slots_per_cu(const Canon &c, int nd, int threads)
{
    static const int forced = [] {
        const char *e = getenv("WS_PLAN_SLOTS"); // development knob
        return e ? atoi(e) : 0;
    }();
    if (forced > 0) return forced;
    const MarchEntry *e = find_march(c);
    const int guess = nd <= kNDNarrow && threads <= 512 && c.ww <= 9 ? 2 : 1;
    if (!e) return guess;
    // one question per kernel and block size, asked once (LDS never is the limit at these sizes)
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const MarchEntry *, int>, int>> known;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &k : known)
        if (k.first.first == e && k.first.second == threads) return k.second;
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void *>(e->fn), threads, 0) != hipSuccess ||
        blocks < 1) {
        (void)hipGetLastError();
        return guess; // (no device: not remembered)
    }
    const int slots = blocks > 2 ? 2 : blocks;
    known.push_back({{e, threads}, slots});
    return slots;
}

bool march_plan(const Canon &c, int num_cus, int tune_nxr, int tune_strip_rows, int tune_threads,
                MarchLaunch *out)
{
    if (!march_supported(c)) return false;
    MarchLaunch m{};
    const int nd = march_nd(c);
    m.x_per_thread = kX;
    m.nd_per_thread = nd;
    m.max_threads = kMaxT;
    const int dcount = c.d_hi - c.d_lo + 1;
    const int out_w = c.ox1 - c.ox0, out_h = c.oy1 - c.oy0;
    // d-chunks per tile.  One tile holds at most kMaxChunks chunks (wider disparity ranges would
    // leave too few columns per tile); beyond that the range is cut into equal d-group passes that
    // meet in a plane of keys.
    const int nch_total = ceil_div(dcount, nd);
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
        // One workgroup per CU at a time (its registers fill the CU).  A strip of R rows costs about
        // R + (wh - 1) / 2 + 3 row times (the wh - 1 warm-up rows only add, the prologue is worth ~3 rows) and the
        // chip works through ceil(workgroups / CUs) rounds of them: take the strip count with the cheapest total
        // (config 3's own sweep, profiles/r01/sweep_tiles_config3.csv, has its minimum where this puts it).
        // (With 4 disparities per thread and a window up to 9 x 9 the kernel stays within 128 VGPRs: two
        // workgroups share a CU, four waves per SIMD, and a round is twice as many workgroups -- the runtime's
        // occupancy figure where a device is there to ask, that rule of thumb for ws_plan without one.)
        const int slots = march_slots_per_cu(c, nd, m.threads);
        double best_cost = 0.0;
        strips = 1;
        for (int sc = 1; sc <= out_h; ++sc) {
            const int rows = ceil_div(out_h, sc), st = ceil_div(out_h, rows);
            if (st != sc) continue; // (the same strips as a smaller count already seen)
            const int rounds = ceil_div(m.tiles * st, num_cus * slots);
            const double cost = rounds * (rows + 0.5 * (c.wh - 1) + 3.0);
            if (sc == 1 || cost < best_cost) { best_cost = cost; strips = sc; }
        }
        if (strips > out_h) strips = out_h;
    }
    m.strip_rows = ceil_div(out_h, strips);
    m.strips = ceil_div(out_h, m.strip_rows);
    const int dt = m.nch * nd;
    const int nreg = kX / 4, nregb = march_nreg_b(kX, nd);
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


const char *march_kernel_name(const Canon &c, const MarchLaunch &)
{
    const MarchEntry *e = find_march(c);
    return e ? e->name : "";
}

hipError_t launch_march(const Canon &c, const MarchLaunch &m, Plane a, Plane b, Plane bias,
                        float *out, double *out64, int out_pitch, void *keys, int keys_pitch, int32_t *cost_out, int cost_pitch,
                        hipStream_t s)
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
    g.out64 = out64;
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
    g.d_top = c.d_lo + m.passes * m.nch * m.nd_per_thread - 1;
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
    const MarchFn fn = cost_out ? e->fn_cost : e->fn;
    if (!fn) return hipErrorInvalidValue;
    if (m.lds_bytes > 48 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)m.lds_bytes);
        if (err != hipSuccess) return err;
    }
    dim3 grid(round_up(m.tiles * m.strips, 8));
    g.keys = keys;
    g.keys_pitch = keys_pitch;
    g.cost_out = cost_out;
    g.cost_pitch = cost_pitch;
    for (int pass = 0; pass < m.passes; ++pass) {
        g.d_first = c.d_lo + pass * m.nch * m.nd_per_thread;
        g.pass_mode = m.passes == 1 ? 0 : pass == 0 ? 1 : pass == m.passes - 1 ? 3 : 2;
        hipLaunchKernelGGL(fn, grid, dim3(m.threads), m.lds_bytes, s, g);
    }
    return hipGetLastError();
}

} // namespace wsamd
