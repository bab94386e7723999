// ws_march_kernel.h -- the marching kernel's template (device code), shared by the two translation units that
// instantiate it: ws_march.hip (8 disparities per thread, compiled with the compiler's "max-ilp" scheduling strategy:
// its two interleaved prefix chains then stay interleaved -- 13 instead of 70 s_nop in the 7 x 7 SSD step, no spills at
// any window) and ws_march_nd4.hip (4 per thread, default strategy: max-ilp costs those kernels a few registers and
// with them the 128-VGPR line that lets two workgroups share a CU).  Overview of the device code: ws_march.hip.
#pragma once
#include "ws_device.h"

#include <stddef.h>

namespace wsamd {

// ------------------------------------------------------------------------------------------
// the marching kernel
// ------------------------------------------------------------------------------------------
// What the stages ahead of the chains read (ws_march_kernel, produce): 16 dwords, one scalar load.  They re-read it from
// the kernel's argument segment every step instead of keeping ~40 values derived from it in scalar registers across
// the chains' arithmetic (first version: 60 .. 200 spilled scalar registers per instantiation, a v_readlane each).
struct StageArgs {
    // the caller's CV_8UC3 rows (BlockSearch.cpp:41,46,59 reads the same bytes): A carries the outputs, B the candidates
    const uint8_t *img_a;
    const uint8_t *img_b;
    int stride_a, stride_b; // bytes
    int wa, wb;             // pixels
    int nxr, nch;
    int wx0, boff;
    int d_first; // first disparity of chunk 0 in THIS launch (d_lo + pass * chunks * ND)
    int mirror;
    int b_lo, b_hi;
};
static_assert(sizeof(StageArgs) == 64, "one s_load_dwordx16");

struct MarchArgs {
    StageArgs st;
    float *out;
    int16_t *out16; // if set: 16-bit integers here instead of floats to `out` (the map as it crosses PCIe, ws_capi.cpp: wire format)
    int out_pitch;
    int border;       // left view: this launch also writes the zeros outside [ox0,ox1) x [oy0,oy1) of the out_w x out_h map
    int out_w, out_h;
    int wy0;
    int d_lo, d_hi;
    int d_top;     // d_lo + passes * chunks * ND - 1: the padded upper end of the range (SSD tie tags count from it)
    int pass_mode; // 0 = the only pass, 1 = first, 2 = middle, 3 = last of several d-group passes
    void *keys;    // several passes: plane of the best keys so far (slot_t per pixel)
    int keys_pitch;
    int ox0, ox1, oy0, oy1;
    int strip_rows, tiles, strips;
    int tile_stride; // output columns per tile: nxr * X, or (nxr - 1) * X for the halo-exchange kernels (march_pk_halo)
    int prefer_large, fallback_neg;
    int tag_bits; // SAD: keys are (cost << tag_bits) | global tie tag
    int32_t *cost_out; // optional (smoothFactor passes): the winner's cost, SSD without the sum of a^2
    int cost_pitch;
    int tune_prod_wave;  // development knobs (WS_STAGE_WAVE / WS_FLUSH_WAVE): the first wave that unpacks / that flushes; -1 = default
    int tune_flush_wave;
    int tune_a_gap;
};
static_assert(offsetof(MarchArgs, st) == 0, "produce() reads StageArgs at the start of the kernel's argument segment");

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

// ---- staging the caller's CV_8UC3 rows ---------------------------------------------------------------------------
// The reference reads the caller's Mat directly (BlockSearch.cpp:41,46,59); so does this kernel.  Rounds 1-3 ran a
// pre-pass that rewrote both images as dword planes and box-summed a "bias" plane (8.7 MB in, 20.8 MB out at config 2,
// 15 % of a pair's device time); now a tile's rows travel HBM -> LDS as the bytes they are and the workgroup unpacks
// them itself, two stages ahead of their use, one barrier between stages (the barrier every step ends with anyway):
//
//   step a-2  DMA      the 16-byte blocks that hold the tile's bytes of image row a go to a raw LDS buffer
//                      (global_load_lds_dwordx4, no VGPR staging; the blocks are aligned in HBM, so the tile's first
//                      byte sits at offset s = address & 15 of the buffer -- any row stride, any base pointer);
//   step a-1  PRODUCE  a lane takes 4 pixels = 12 bytes (4 dwords, v_alignbyte by s & 3, three v_perm) to one 16-byte
//                      quad of the dword ring (B | G<<8 | R<<16, the layout the chains read), zero outside the image,
//                      mirrored for the right view.  SSD: the same lane moves the quad's four column sums
//                      G = sum over the window rows of b^2 + the fused chain's correction term (kept in LDS), and --
//                      with the sums of the next lanes, fetched by DPP inside the additions -- turns 4 + WW - 1 of them
//                      into the 4 bias values of the output row of step a (running prefix, differences);
//   step a    the chains read the row.
// What counts is instructions per WAVE and step: a wave issues one instruction every ~5 cycles whatever it is, and the
// step ends when the slowest wave reaches the barrier.  (The first version computed its addresses from the kernel's
// arguments every step -- ~150 scalar instructions in every producing wave -- and had a third stage for the bias rows:
// the kernel took 168 us instead of 117 at config 2.)  Hence: the raw buffers and the column sums sit at COMPILE-TIME
// LDS addresses (a stage area sized for the widest tile row, ws_device.h), a wave's roles are fixed for the kernel's
// life, and what varies per step is a handful of scalars.
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct RawSide {         // one image as a tile sees it (wave-uniform; used while the kernel sets itself up)
    const uint8_t *base; // the caller's rows
    int stride;          // bytes per row
    int c0;              // image column behind raw index 0 (may lie outside the image)
    int v_lo, v_hi;      // raw indices [v_lo, v_hi) are inside the image
    int nq;              // quads per tile row
};

// mirror (right view: canonical x = w - 1 - x): logical p <-> canonical canon0 + p <-> image column c0 + (n4 - 1 - p),
// i.e. raw quad q holds logical quad nq - 1 - q, pixels reversed
__device__ __forceinline__ RawSide raw_side(const uint8_t *base, int stride, int w, int canon0, int n4, int mirror)
{
    RawSide s;
    s.base = base;
    s.stride = stride;
    s.nq = n4 >> 2;
    s.c0 = mirror ? w - canon0 - n4 : canon0;
    s.v_lo = max(0, -s.c0);
    s.v_hi = max(s.v_lo, min(n4, w - s.c0)); // (empty when the tile's columns miss the image altogether)
    return s;
}

// what a wave that issues the copies of one image keeps (scalars)
struct RawDma {
    uintptr_t f0;     // address of the first byte inside the image of the strip's first window row
    uint32_t stride;  // bytes per image row
    uint32_t nbytes;  // bytes inside the image per tile row (0: none)
    uint32_t off3;    // 3 * v_lo: raw index 0 sits that many bytes before f0
};
__device__ __forceinline__ RawDma raw_dma_setup(const RawSide &s, int y0)
{
    RawDma d;
    d.f0 = reinterpret_cast<uintptr_t>(s.base) + (uintptr_t)((long long)y0 * s.stride + 3LL * (s.c0 + s.v_lo));
    d.stride = (uint32_t)s.stride;
    d.nbytes = 3u * (uint32_t)(s.v_hi - s.v_lo);
    d.off3 = 3u * (uint32_t)s.v_lo;
    return d;
}

// Row i of the strip: every 16-byte block that holds a byte of the tile's columns -> the raw buffer at LDS byte address
// raw_lds, so that raw index 0's first byte lands at offset (its address & 15).
//
// (raw_lds: an integer, not a pointer -- a generic pointer cast to the LDS address space at every call carries a null
// check that one of the compiler's scheduling strategies could not encode.)  The instruction is issued through inline
// assembly ON PURPOSE: for the builtin the compiler makes every later LDS read of the wave wait for vmcnt(0) (it cannot
// know the copy fills a buffer nobody reads in this step), which exposes the copy's whole latency at the top of the
// arithmetic; here nothing waits until the explicit dma_wait() in front of the step's barrier (A/B on one MI355X,
// config 2, round 1: 171 -> 166 us).
__device__ __forceinline__ void raw_dma(uint32_t raw_lds, const RawDma &d, int i, int lane)
{
    if (d.nbytes == 0) return; // (uniform: the tile's columns of this image are all outside it)
    const uintptr_t f = d.f0 + (uintptr_t)((uint32_t)i * d.stride);
    const uintptr_t a0 = f & ~(uintptr_t)15;                     // an aligned block that holds a byte of the image
    const int nblk = (int)((((uint32_t)f & 15u) + d.nbytes + 15u) >> 4); // stays inside that byte's page
    const uint32_t dst0 = raw_lds + (((uint32_t)f & ~15u) - (((uint32_t)f - d.off3) & ~15u));
    for (int idx = lane; idx < nblk; idx += 64) {
        // M0 is written right in front of its use and put back behind it, inside one statement: the compiler keeps
        // values of its own in M0 (LDS-DMA builtins, indexed register moves) and is not told otherwise -- M0 is a
        // reserved register, a clobber of it is refused with a warning
        const uint32_t la = __builtin_amdgcn_readfirstlane(dst0 + 16u * (uint32_t)(idx - lane));
        uint32_t saved_m0;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(saved_m0)
                     : "v"(reinterpret_cast<const uint8_t *>(a0) + 16 * (size_t)idx), "s"(la)
                     : "memory");
    }
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t ws_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ws_u32x4 lds_u128;
// LDS accesses by byte address (a 32-bit integer: no generic pointer, no null check, immediates fold into the offset field)
__device__ __forceinline__ const lds_u32 *lds_at32(uint32_t byte_addr) { return reinterpret_cast<const lds_u32 *>((uintptr_t)byte_addr); }
__device__ __forceinline__ uint4 lds_load128(uint32_t byte_addr)
{
    const ws_u32x4 v = *reinterpret_cast<const lds_u128 *>((uintptr_t)byte_addr); // ds_read_b128
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_store128(uint32_t byte_addr, uint4 v)
{
    ws_u32x4 t = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<lds_u128 *>((uintptr_t)byte_addr) = t; // ds_write_b128
}

// 12 bytes at LDS byte address `at` (dword aligned) + byte phase sh (0..3) -> four pixel dwords, in memory order
template <bool CENTRED>
__device__ __forceinline__ void raw_unpack(uint32_t (&px)[4], uint32_t at, uint32_t sh)
{
    const lds_u32 *p = lds_at32(at);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
    const uint32_t a0 = __builtin_amdgcn_alignbyte(d1, d0, sh), a1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                   a2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
    // v_perm_b32: selector bytes 0..3 pick from the second operand, 4..7 from the first, 0x0c is a zero byte
    px[0] = a0 & 0x00ffffffu;
    px[1] = __builtin_amdgcn_perm(a1, a0, 0x0c050403u);
    px[2] = __builtin_amdgcn_perm(a2, a1, 0x0c040302u);
    px[3] = a2 >> 8;
    if constexpr (CENTRED) {
#pragma unroll
        for (int e = 0; e < 4; ++e) px[e] ^= kCentre;
    }
}

// lane + n's value inside a row of 16 lanes (n = 1 .. 15; lanes past the row's end read 0)
template <int N>
__device__ __forceinline__ uint32_t from_lane_plus(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x100 + N, 0xf, 0xf, true); // row_shl:N
}

// One row entering (SIGN=+1) or leaving (SIGN=-1) the window of every (column, disparity) this
// thread owns.
//   SAD: V = (window sum << shift) + global tie tag          key = V
//   SSD: V = local tie tag - (2 * cross sum << LT)           key = bias[xb] + V
//        (bias = box sum of the squared target pixels << LT, or poison for an invalid centre)
// With KEY the candidate keys are folded into best[] (signed min; equal costs go to the smaller
// tag, i.e. to the disparity the reference's strict '<' keeps).
#ifndef WS_FUSE
#define WS_FUSE 1
#endif
constexpr bool kFuseSsd = WS_FUSE != 0; // build-time knob for A/B runs (tools/variants.py); the column sums follow it (kMul)

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

// SSD, steady state: the row entering the window and the row leaving it in ONE prefix chain.  The chain can only add,
// so the leaving row's reference pixels come complemented (qa = ~a): for plain bytes (255 - a) . b = 255 b - a . b, for
// centred bytes (~a = -a - 1) (-a - 1) . b = -b - a . b (the 4th byte of b is 0 either way).  The chain's differences are
// then  (cross sum of the entering row) - (cross sum of the leaving row) + K * (sum of the leaving row's target bytes
// over the window), K = 255 or -1: the last term does not depend on the reference image, only on (row, target
// column), and the stage that sums the bias rows (ws_march_kernel, produce) folds its running total over the rows that
// left the strip so far into them: bias = (sum b^2 + 2 K E) << LT.  V and the bias wrap around in 32 bits by themselves,
// their sum -- the key -- is exact.  One difference, one accumulate and one chain per step instead of two each:
// 7.0 instead of 9.0 instructions per hypothesis in the steady state (+ X + WW - 1 v_not per thread and step).
template <int X, int ND, int WW, bool CENTRED>
__device__ __forceinline__ void march_fused_ssd(int32_t (&V)[X][ND], int32_t (&best)[X], const uint32_t (&pa)[X + WW - 1],
                                                const uint32_t (&pb)[X + WW + ND - 2], const uint32_t (&qa)[X + WW - 1],
                                                const uint32_t (&qb)[X + WW + ND - 2], const uint32_t (&bi)[X + ND - 1], int shift)
{
    constexpr int NA = X + WW - 1;
#pragma unroll
    for (int j = 0; j < ND; j += 2) {
        uint32_t S0[NA], S1[NA];
        uint32_t s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            s0 = pix_dot<CENTRED>(pa[i], pb[i - j + ND - 1], s0);
            s1 = pix_dot<CENTRED>(pa[i], pb[i - j + ND - 2], s1);
            s0 = pix_dot<CENTRED>(qa[i], qb[i - j + ND - 1], s0);
            s1 = pix_dot<CENTRED>(qa[i], qb[i - j + ND - 2], s1);
            S0[i] = s0;
            S1[i] = s1;
        }
#pragma unroll
        for (int x = 0; x < X; ++x) {
            const uint32_t w0 = (x ? S0[x - 1] : 0u) - S0[x + WW - 1]; // (V accumulates -2 * cross)
            const uint32_t w1 = (x ? S1[x - 1] : 0u) - S1[x + WW - 1];
            V[x][j] = (int32_t)((w0 << shift) + (uint32_t)V[x][j]);
            V[x][j + 1] = (int32_t)((w1 << shift) + (uint32_t)V[x][j + 1]);
            const int32_t k0 = (int32_t)bi[x - j + ND - 1] + V[x][j];
            const int32_t k1 = (int32_t)bi[x - j + ND - 2] + V[x][j + 1];
            // (spelled out: the compiler splits min(best, min(k0, k1)) into two v_min_i32 for a third of the columns; the
            // first pair of a step has nothing to compare with yet -- best[] enters as INT_MAX)
            if (j == 0) best[x] = min(k0, k1);
            else asm("v_min3_i32 %0, %1, %2, %3" : "=v"(best[x]) : "v"(best[x]), "v"(k0), "v"(k1));
        }
    }
}

// lane + 1's value (DPP; march_pk_halo says why lane + 1 is the next run of the tile)
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v)
{
    // row_shl:1: lane i reads lane i + 1 of its row of 16; the row's last lane reads 0 (bound_ctrl) -- it is a tile's
    // last run.  (mov_dpp, not update_dpp(0, ...): that one costs a v_mov of the 0 in front of every move)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x101, 0xf, 0xf, true);
}

// The fused chain with the window's right-hand part taken from the neighbouring thread (see march_pk_halo below for the
// idea and the tile geometry): chains over the thread's OWN X columns, window x = S[X-1] - S[x-1] of its own block plus
// the next run's prefix N[x + WW - 1 - X] -- fetched by DPP inside the subtraction itself (v_subrev_u32_dpp).  Per
// disparity at 7 x 7: 16 instead of 28 v_dot4, 14 instead of 8 subtractions.
template <int X, int ND, int WW, bool CENTRED>
__device__ __forceinline__ void march_fused_ssd_halo(int32_t (&V)[X][ND], int32_t (&best)[X], const uint32_t (&pa)[X],
                                                     const uint32_t (&pb)[X + ND - 1], const uint32_t (&qa)[X],
                                                     const uint32_t (&qb)[X + ND - 1], const uint32_t (&bi)[X + ND - 1], int shift)
{
    static_assert(WW - 1 <= X && WW >= 2, "the next run covers the whole overhang of a window");
#pragma unroll
    for (int j = 0; j < ND; j += 2) {
        uint32_t S0[X], S1[X];
        uint32_t s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < X; ++i) {
            s0 = pix_dot<CENTRED>(pa[i], pb[i - j + ND - 1], s0);
            s1 = pix_dot<CENTRED>(pa[i], pb[i - j + ND - 2], s1);
            s0 = pix_dot<CENTRED>(qa[i], qb[i - j + ND - 1], s0);
            s1 = pix_dot<CENTRED>(qa[i], qb[i - j + ND - 2], s1);
            S0[i] = s0;
            S1[i] = s1;
        }
#pragma unroll
        for (int x = 0; x < X; ++x) {
            constexpr int kLastInside = X - WW; // the last column whose window ends inside the thread's own columns
            uint32_t w0, w1;                    // (V accumulates -2 * cross: minus the window's sum)
            if (x <= kLastInside) {
                const int hi = x <= kLastInside ? x + WW - 1 : 0;
                w0 = (x ? S0[x - 1] : 0u) - S0[hi];
                w1 = (x ? S1[x - 1] : 0u) - S1[hi];
            } else {
                const int m = x + WW - 1 - X; // the next run's prefix that completes this window
                w0 = ((x ? S0[x - 1] : 0u) - S0[X - 1]) - from_next_lane(S0[m]);
                w1 = ((x ? S1[x - 1] : 0u) - S1[X - 1]) - from_next_lane(S1[m]);
            }
            V[x][j] = (int32_t)((w0 << shift) + (uint32_t)V[x][j]);
            V[x][j + 1] = (int32_t)((w1 << shift) + (uint32_t)V[x][j + 1]);
            const int32_t k0 = (int32_t)bi[x - j + ND - 1] + V[x][j];
            const int32_t k1 = (int32_t)bi[x - j + ND - 2] + V[x][j + 1];
            // (spelled out: the compiler splits min(best, min(k0, k1)) into two v_min_i32 for a third of the columns; the
            // first pair of a step has nothing to compare with yet -- best[] enters as INT_MAX)
            if (j == 0) best[x] = min(k0, k1);
            else asm("v_min3_i32 %0, %1, %2, %3" : "=v"(best[x]) : "v"(best[x]), "v"(k0), "v"(k1));
        }
    }
}

// ---- SAD in packed 16-bit halves (windows up to 9 x 9) -----------------------------------------------------------
// A window sum of absolute differences is at most ww * wh * 3 * 255: up to 9 x 9 it fits 16 bits, and then two
// disparities share every register and every instruction behind the pixel differences themselves:
//   * the prefix chain of disparity d runs in the low half (v_sad_u8), the chain of d + 1 in the high half of the
//     same accumulator (v_sad_hi_u8: D = (sum |a - b| << 16) + acc); a chain's total stays below 2^16, so the low
//     half never carries into the high one;
//   * entering minus leaving row, the window's difference of two prefix values and the accumulation into the running
//     sums are one v_pk_sub_u16 / v_pk_add_u16 each per PAIR of disparities (modulo 2^16 per half: the running sum
//     itself is exact because it fits);
//   * only the running minimum needs (cost, tie tag) keys: (V << 16) | tag for the low half is one v_lshl_or_b32,
//     (V & 0xffff0000) | tag for the high half one v_and_or_b32, the minimum of both and the best so far one
//     v_min3_u32.  A tag register with its upper half set poisons a disparity beyond d_hi for free; candidates whose
//     target centre leaves [b_lo, b_hi] get their cost field forced to 0xffff by one more v_or -- only in the waves
//     that hold such candidates next to valid ones (tiles at the image's edge: a wave-uniform branch around the loops).
// Per hypothesis in the steady state (X = 8, 9 x 9): 4.0 (chains) + 1.0 + 0.5 + 0.5 (packed T, difference, accumulate)
// + 1.5 (keys, min) = 7.5 instructions, against 8.5 for the 32-bit form; half the running-sum registers.
__host__ __device__ constexpr bool march_pk_window(int ww, int wh) { return ww * wh * 3 * 255 <= 65535; }
constexpr uint32_t kPkNone = 0xffff0000u; // cost field of "no valid candidate" (a valid cost is at most 61 965)

typedef unsigned short ws_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (ws_u16x2)(__builtin_bit_cast(ws_u16x2, a) - __builtin_bit_cast(ws_u16x2, b))); // v_pk_sub_u16
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (ws_u16x2)(__builtin_bit_cast(ws_u16x2, a) + __builtin_bit_cast(ws_u16x2, b))); // v_pk_add_u16
}

// PHASE 0: a row enters, no keys.  1: a row enters, keys.  2: a row enters, a row leaves (qa / qb), keys.
// Which of a thread's candidates have their target centre inside [b_lo, b_hi]: centre of (x, j) = xb0 + m with
// m = x - j + ND - 1, valid iff mlo <= m <= mlo + span.  Two registers and a compare per key in the few waves that need it
// (a table of X + ND - 1 masks cost the 16-disparity kernels 50 spilled registers).
struct PkMask {
    int mlo;
    uint32_t span;
    __device__ __forceinline__ uint32_t operator()(int m) const { return (uint32_t)(m - mlo) > span ? kPkNone : 0u; }
    // called once per row step: the masks are loop invariant, and hoisted out of the row loop they are X + ND - 1
    // registers again -- the empty asm makes them the step's own
    __device__ __forceinline__ PkMask fresh() const
    {
        PkMask f = *this;
        asm volatile("" : "+v"(f.mlo));
        return f;
    }
};

template <int X, int ND, int WW, int PHASE, bool MASKED>
__device__ __forceinline__ void march_pk(uint32_t (&Vp)[X][ND / 2], uint32_t (&best)[X], const uint32_t (&pa)[X + WW - 1],
                                         const uint32_t (&pb)[X + WW + ND - 2], const uint32_t (&qa)[X + WW - 1],
                                         const uint32_t (&qb)[X + WW + ND - 2], const uint32_t (&tagr)[ND],
                                         const PkMask mk)
{
    constexpr int NA = X + WW - 1;
    auto finish = [&](const uint32_t (&T)[NA], int j) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < X; ++x) {
            const uint32_t w = x ? pk_sub(T[x + WW - 1], T[x - 1]) : T[WW - 1];
            const uint32_t v = pk_add(Vp[x][j / 2], w);
            Vp[x][j / 2] = v;
            if constexpr (PHASE >= 1) {
                uint32_t k0 = (v << 16) | tagr[j];              // disparity j      (v_lshl_or_b32)
                uint32_t k1 = (v & 0xffff0000u) | tagr[j + 1];  // disparity j + 1  (v_and_or_b32)
                if constexpr (MASKED) {
                    k0 |= mk(x - j + ND - 1);
                    k1 |= mk(x - j + ND - 2);
                }
                // (spelled out: the compiler splits min(best, min(k0, k1)) into two v_min_u32 for a third of the columns)
                if (j == 0) best[x] = min(k0, k1); // (the first pair of a step: nothing to compare with yet)
                else asm("v_min3_u32 %0, %1, %2, %3" : "=v"(best[x]) : "v"(best[x]), "v"(k0), "v"(k1));
            }
        }
    };
    if constexpr (PHASE == 2) {
        // the entering and the leaving row's chains of one disparity pair interleave in the issue stream
#pragma unroll
        for (int j = 0; j < ND; j += 2) {
            uint32_t T[NA];
            uint32_t sn = 0, so = 0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                sn = __builtin_amdgcn_sad_u8(pa[i], pb[i - j + ND - 1], sn);
                so = __builtin_amdgcn_sad_u8(qa[i], qb[i - j + ND - 1], so);
                sn = __builtin_amdgcn_sad_hi_u8(pa[i], pb[i - j + ND - 2], sn);
                so = __builtin_amdgcn_sad_hi_u8(qa[i], qb[i - j + ND - 2], so);
                T[i] = pk_sub(sn, so);
            }
            finish(T, j);
        }
    } else {
        // one row only: the chains of two disparity pairs interleave
#pragma unroll
        for (int j = 0; j < ND; j += 4) {
            uint32_t T0[NA], T1[NA];
            uint32_t s0 = 0, s1 = 0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                s0 = __builtin_amdgcn_sad_u8(pa[i], pb[i - j + ND - 1], s0);
                s1 = __builtin_amdgcn_sad_u8(pa[i], pb[i - j + ND - 3], s1);
                s0 = __builtin_amdgcn_sad_hi_u8(pa[i], pb[i - j + ND - 2], s0);
                s1 = __builtin_amdgcn_sad_hi_u8(pa[i], pb[i - j + ND - 4], s1);
                T0[i] = s0;
                T1[i] = s1;
            }
            finish(T0, j);
            finish(T1, j + 2);
        }
    }
}

// ---- packed SAD with the window's right-hand part taken from the neighbouring thread ------------------------------
// march_pk runs every prefix chain over X + WW - 1 positions to get X windows: at 9 x 9 half of its v_sad are spent on
// columns the thread to the right covers as well.  Here a thread's chains stop after its OWN X columns (T[0 .. X-1]),
// and the part of window x that lies in the next thread's columns, positions X .. x + WW - 1, is that thread's prefix
// N[x + WW - 1 - X], fetched with one DPP move per value (row_shl:1: lane + 1 is run r + 1 of the same d-chunk, the
// planner keeps nxr at 8 or 16 so that a row of 16 lanes holds whole tiles rows):
//     w[x] = T[X-1] - T[x-1] + N[x + WW - 1 - X]          (w[x] = T[x + WW - 1] - T[x-1] while the window is inside)
// 32 v_sad + 8 v_pk_sub less and 8 v_mov_dpp + 7 v_pk_add more per pair of disparities at 9 x 9 (X = 8).  The LAST run
// of a tile has no right-hand neighbour: its windows are wrong and never leave the workgroup -- tiles advance by
// (nxr - 1) * X columns and the last run only feeds the one before it (it covers exactly that run's window overhang:
// WW - 1 <= X).  A thread also reads only X pixels of A and X + ND - 1 of B per row instead of X + WW - 1 / X + WW + ND - 2.

template <int X, int ND, int WW, int PHASE, bool MASKED>
__device__ __forceinline__ void march_pk_halo(uint32_t (&Vp)[X][ND / 2], uint32_t (&best)[X], const uint32_t (&pa)[X],
                                              const uint32_t (&pb)[X + ND - 1], const uint32_t (&qa)[X],
                                              const uint32_t (&qb)[X + ND - 1], const uint32_t (&tagr)[ND],
                                              const PkMask mk)
{
    static_assert(WW - 1 <= X && WW >= 2, "the next run covers the whole overhang of a window");
    constexpr int NN = WW - 1;
    auto finish = [&](const uint32_t (&T)[X], int j) __attribute__((always_inline)) {
        uint32_t N[NN];
#pragma unroll
        for (int i = 0; i < NN; ++i) N[i] = from_next_lane(T[i]);
#pragma unroll
        for (int x = 0; x < X; ++x) {
            uint32_t w;
            constexpr int kLastInside = X - WW; // the last column whose window ends inside the thread's own columns (< 0: none)
            if (x <= kLastInside) {
                const uint32_t t_hi = T[x <= kLastInside ? x + WW - 1 : 0];
                w = x ? pk_sub(t_hi, T[x - 1]) : t_hi;
            } else {
                const uint32_t own = x ? pk_sub(T[X - 1], T[x - 1]) : T[X - 1];
                w = pk_add(own, N[x + WW - 1 - X]);
            }
            const uint32_t v = pk_add(Vp[x][j / 2], w);
            Vp[x][j / 2] = v;
            if constexpr (PHASE >= 1) {
                uint32_t k0 = (v << 16) | tagr[j];
                uint32_t k1 = (v & 0xffff0000u) | tagr[j + 1];
                if constexpr (MASKED) {
                    k0 |= mk(x - j + ND - 1);
                    k1 |= mk(x - j + ND - 2);
                }
                if (j == 0) best[x] = min(k0, k1);
                else asm("v_min3_u32 %0, %1, %2, %3" : "=v"(best[x]) : "v"(best[x]), "v"(k0), "v"(k1));
            }
        }
    };
    if constexpr (PHASE == 2) {
#pragma unroll
        for (int j = 0; j < ND; j += 2) {
            uint32_t T[X];
            uint32_t sn = 0, so = 0;
#pragma unroll
            for (int i = 0; i < X; ++i) {
                sn = __builtin_amdgcn_sad_u8(pa[i], pb[i - j + ND - 1], sn);
                so = __builtin_amdgcn_sad_u8(qa[i], qb[i - j + ND - 1], so);
                sn = __builtin_amdgcn_sad_hi_u8(pa[i], pb[i - j + ND - 2], sn);
                so = __builtin_amdgcn_sad_hi_u8(qa[i], qb[i - j + ND - 2], so);
                T[i] = pk_sub(sn, so);
            }
            finish(T, j);
        }
    } else {
#pragma unroll
        for (int j = 0; j < ND; j += 4) {
            uint32_t T0[X], T1[X];
            uint32_t s0 = 0, s1 = 0;
#pragma unroll
            for (int i = 0; i < X; ++i) {
                s0 = __builtin_amdgcn_sad_u8(pa[i], pb[i - j + ND - 1], s0);
                s1 = __builtin_amdgcn_sad_u8(pa[i], pb[i - j + ND - 3], s1);
                s0 = __builtin_amdgcn_sad_hi_u8(pa[i], pb[i - j + ND - 2], s0);
                s1 = __builtin_amdgcn_sad_hi_u8(pa[i], pb[i - j + ND - 4], s1);
                T0[i] = s0;
                T1[i] = s1;
            }
            finish(T0, j);
            finish(T1, j + 2);
        }
    }
}

template <int X, int ND, int WW, int WH, bool SSD, int MAXT, bool COST = false, bool HALO = false>
__global__ void __launch_bounds__(MAXT) ws_march_kernel(const MarchArgs g)
{
    static_assert(X % 4 == 0 && ND % 4 == 0, "runs start on 16-byte quads");
    constexpr int NREG = X / 4, NREGB = march_nreg_b(X, ND);
    constexpr int LT = ilog2c(ND);
    constexpr bool CENTRED = SSD && ssd_needs_centring(WW, WH, ND);
    constexpr int NR = WH + 2; // ring rows: WH+1 in use by a step, 1 being unpacked for the next
    // merge slots: SSD (cost << LT | 7) : global tie tag as one signed 64-bit key, SAD the 32-bit key itself;
    // a key at or above kValidKeyBound (in its cost word) is "no valid candidate"
    // (packed SAD: (cost << 16) | global tie tag as an UNSIGNED 32-bit key, cost field 0xffff = no valid candidate)
    constexpr bool PK = !SSD && march_pk_window(WW, WH);
    static_assert(!HALO || PK || (SSD && kFuseSsd), "the halo exchange: packed SAD, or the fused SSD chain");
    typedef typename std::conditional<SSD, long long, typename std::conditional<PK, uint32_t, int32_t>::type>::type slot_t;
    const slot_t kEmpty = SSD ? (slot_t)LLONG_MAX : PK ? (slot_t)0xffffffffu : (slot_t)INT_MAX;

    extern __shared__ uint4 ws_smem4[];
    uint32_t *smem = reinterpret_cast<uint32_t *>(ws_smem4);

    const int NT = blockDim.x, tid = threadIdx.x;
    const int tx = g.st.nxr * X, dt = g.st.nch * ND;
    // pixels of a row the tile's threads read (HALO: nobody reads past the last run's own columns)
    // (the SSD halo kernel keeps the long chains while a strip's window fills: it reads what the plain kernel reads)
    const MarchLds L = march_lds_layout(X, ND, WW, WH, SSD, HALO && PK, g.st.nxr, g.st.nch);
    const int n_a = L.n_a, n_b = L.n_b, n_bi = L.n_bi;
    const int ro_a = march_region_dwords(n_a, NREG), ro_b = march_region_dwords(n_b, NREGB);
    const int ro_bi = SSD ? march_region_dwords(n_bi, NREGB) : 0;
    const int a_w = L.a_w, b_w = L.b_w, bi_w = L.bi_w;
    // the stage area first, at compile-time offsets (ws_device.h): two raw rows of each image -- one landing, one being
    // unpacked -- and, SSD, the column sums G after the last two unpacked rows
    constexpr int kRA = march_raw_bytes(kStageMaxNA), kRB = march_raw_bytes(march_max_nb(ND));
    constexpr int kGB = SSD ? 4 * (march_max_nb(ND) + 16) : 0;
    constexpr int oRawA = 0, oRawB = 2 * kRA, oG = 2 * kRA + 2 * kRB, kStage = march_stage_bytes(ND, SSD);
    static_assert(kStage == oG + 2 * kGB && kStage % 16 == 0, "stage area");
    static_assert(MAXT <= kStageThreads, "the stage area is sized for workgroups of up to kStageThreads");
    uint32_t *ringA = smem + (kStage + L.desc_bytes) / 4; // (behind the stage area: the stages' per-thread descriptors)
    // SSD: a twin of ring A with every dword complemented -- what the fused chain multiplies a LEAVING row's target pixels
    // by ((255 - a) . b, march_fused_ssd).  The unpacking lane writes both; round 3 complemented in the chains' threads: 14
    // v_not per thread and step (X + WW - 1 of ~500 instructions, in every wave) against one more ds_write_b128 and 4
    // v_not per quad of image A in ONE wave.
    constexpr bool TWIN = SSD && kFuseSsd;
    constexpr int NRA = TWIN ? 2 * NR : NR; // rows of ring A's space
    uint32_t *ringAc = ringA + NR * a_w;
    uint32_t *ringB = ringA + NRA * a_w;
    int32_t *biasr = reinterpret_cast<int32_t *>(ringB + NR * b_w);
    slot_t *slots = reinterpret_cast<slot_t *>(biasr + 2 * bi_w);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32 *)smem; // LDS byte address of the stage area

    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (each with its own
    // L2), so ids b and b+8 share one.  Give every XCD a contiguous range of (strip, tile) pairs:
    // the tiles of a strip overlap in the target-image columns they read and then hit the same L2.
    const int nblk = gridDim.x; // padded to a multiple of 8 by the launcher
    const int logical = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    if (logical >= g.tiles * g.strips) return; // uniform per workgroup
    const int tile_i = logical % g.tiles, strip_i = logical / g.tiles;
    const int tile_x0 = g.ox0 + tile_i * g.tile_stride;
    const int tx_out = HALO ? tx - X : tx; // the columns this tile hands out (HALO: the last run only feeds its neighbour)
    const int ys = g.oy0 + strip_i * g.strip_rows;
    const int ye = min(ys + g.strip_rows, g.oy1);
    if (ys >= ye) return; // uniform per workgroup

    const int dhi_t = g.st.d_first + dt - 1;
    // left view: the map's pixels outside the marching interior are zeros (BlockSearch.cpp:33,36,38: the loops never
    // reach them); the tiles along the interior's edge write them -- their stores drain while the strip's window fills
    if (g.border && (g.pass_mode == 0 || g.pass_mode == 3)) {
        const bool first_t = tile_i == 0, last_t = tile_i == g.tiles - 1;
        const int cx0 = first_t ? 0 : tile_x0, cx1 = last_t ? g.out_w : tile_x0 + tx_out;
        auto zero_rect = [&](int x0, int x1, int y0, int y1) __attribute__((always_inline)) {
            const int w = x1 - x0, n = w * (y1 - y0);
            for (int i = tid; i < n; i += NT) {
                const int yy = y0 + i / w, xx = x0 + i % w;
                if (g.out16) g.out16[(size_t)yy * g.out_pitch + xx] = 0;
                else g.out[(size_t)yy * g.out_pitch + xx] = 0.0f;
            }
        };
        if (strip_i == 0 && g.oy0 > 0) zero_rect(cx0, cx1, 0, g.oy0);
        if (strip_i == g.strips - 1 && g.out_h > g.oy1) zero_rect(cx0, cx1, g.oy1, g.out_h);
        if (first_t && g.ox0 > 0) zero_rect(0, g.ox0, ys, ye);
        if (last_t && g.out_w > g.ox1) zero_rect(g.ox1, g.out_w, ys, ye);
    }

    for (int k = tid; k < 2 * tx; k += NT) slots[k] = kEmpty;
    if constexpr (SSD) { // (G before the strip's first row, both parities)
        for (int k = tid; k < L.n_b4 + 16; k += NT) smem[oG / 4 + k] = smem[(oG + kGB) / 4 + k] = 0u;
    }

    const int r = tid % g.st.nxr, c = tid / g.st.nxr;
    const bool worker = c < g.st.nch;
    // run starts (dword offset inside region 0): A at column X*r, B / bias at column X*r + ND*(nch-1-c)
    const int ia = 4 * r;
    const int ib = 4 * (((X / 4) * r + (ND / 4) * (g.st.nch - 1 - (worker ? c : 0))) / NREGB);
    const int d0 = g.st.d_first + c * ND; // first disparity of this thread's chunk
    const int shift = SSD ? LT + 1 : g.tag_bits;
    // SSD merge: global tie tag = chunk tag | local tag (a multiple of ND, so one v_and_or builds it):
    // the chunk's distance from the preferred end of the padded range [d_lo, d_top]
    const int ctag = g.prefer_large ? g.d_top - d0 - (ND - 1) : d0 - g.d_lo;

    int32_t V[PK ? 1 : X][PK ? 1 : ND];
    uint32_t Vp[PK ? X : 1][PK ? ND / 2 : 1], tagr[PK ? ND : 1];
    PkMask mk{0, 0u};
    bool masked = false; // packed SAD: does this WAVE hold candidates whose target centre is out of range next to valid ones?
    if constexpr (PK) {
#pragma unroll
        for (int x = 0; x < X; ++x)
#pragma unroll
            for (int j = 0; j < ND / 2; ++j) Vp[x][j] = 0u;
#pragma unroll
        for (int j = 0; j < ND; ++j) { // global tag; a disparity beyond d_hi is poisoned through its tag register
            const int d = d0 + j;
            tagr[j] = (uint32_t)(g.prefer_large ? g.d_hi - d : d - g.d_lo) & 0xffffu;
            if (d > g.d_hi) tagr[j] |= kPkNone;
        }
        // target centre of (x, j): tile_x0 + r X + x - (d0 + j) + boff, i.e. xb0 + m with m = x - j + ND - 1
        const int xb0 = tile_x0 + r * X - d0 + g.st.boff - (ND - 1);
        mk.mlo = g.st.b_lo - xb0;
        mk.span = (uint32_t)(g.st.b_hi - g.st.b_lo);
        if (g.st.b_hi < g.st.b_lo) { mk.mlo = 1 << 30; mk.span = 0u; } // (no valid centre at all)
        // Per WAVE: a thread all of whose target centres are out of range poisons its tags instead (free), so only the
        // waves that hold a thread with SOME centres out of range -- the few on the diagonal x - d = b_lo of a tile at
        // the image's edge -- pay the extra v_or per key.  (Wave-uniform: different waves of a workgroup then run
        // different copies of the loops, every copy with the same barriers.)
        const int xb_min = xb0, xb_max = xb0 + X + ND - 2;
        const bool none_ok = worker && (xb_max < g.st.b_lo || xb_min > g.st.b_hi);
        const bool some_bad = worker && !none_ok && (xb_min < g.st.b_lo || xb_max > g.st.b_hi);
        if (none_ok) {
#pragma unroll
            for (int j = 0; j < ND; ++j) tagr[j] |= kPkNone;
        }
        masked = __builtin_amdgcn_ballot_w64(some_bad) != 0;
    } else {
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
                    const int xb = tile_x0 + r * X + x - d + g.st.boff;
                    V[x][j] = (d <= g.d_hi && xb >= g.st.b_lo && xb <= g.st.b_hi) ? tag : kPoison;
                }
            }
        }
    }

    const int ra0 = ys + g.wy0; // first window row of the first output row
    const int nsteps = (ye - ys) + WH - 1;

    // ---- the stages ahead of the chains (overview above raw_dma): what step a does for the rows to come ----------
    // Set up once: a 16-byte DESCRIPTOR per thread and role in LDS -- where the lane's 12 raw bytes sit, where its quad
    // goes, which of its pixels lie inside the image, whether it owns what it computes.  A step then costs a producing
    // wave one ds_read_b128, a handful of scalar instructions and the arithmetic itself; nothing of the geometry below
    // stays in registers across the chains.
    //   x: raw buffer byte offset of the quad's 12 bytes (+ the row's byte phase & ~3, + the parity's buffer)
    //   y: byte offset of the quad inside a ring row | the same inside a bias row << 16
    //   z: byte offset of the quad's column sums (+ the parity's buffer)
    //   w: kDescRing: owns the ring quad (and its column sums), kDescBias: owns the bias quad,
    //      bits 8..11: pixel e lies inside the image, bits 12..15: bias value e belongs to an invalid target centre
    constexpr uint32_t kDescRing = 1u, kDescBias = 2u;
    constexpr int NQN = march_nqn(WW, SSD); // quads to the right whose column sums a quad's bias values need
    static_assert(NQN < 8, "a row of 16 lanes holds a quad and its neighbours");
    constexpr int QROW = 16 - NQN, QW = SSD ? 4 * QROW : 64; // owner lanes per row of 16, quads per role-wave
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = NT >> 6, lane = tid & 63;
    // Roles go to the waves from `pw0` on, and so does the flush: behind the two waves that issue the copies in a
    // workgroup of 8 waves, from the first wave on in smaller ones.  Measured, not derived (profiles/r04/stage_waves.txt):
    // the same work dealt otherwise costs config 2 up to 16 % (0.128 .. 0.151 ms) -- the waves with work at the top of a
    // step start their chains late, out of step with the other waves' bursts of LDS reads.
    const int pw0 = g.tune_prod_wave >= 0 ? g.tune_prod_wave % nwaves : (nwaves > 4 ? 2 : 0);
    const int role0 = wave >= pw0 ? wave - pw0 : wave - pw0 + nwaves;
    const int agap = g.tune_a_gap > 0 && L.nslots == 1 && L.roles_b + L.roles_a + g.tune_a_gap <= nwaves ? g.tune_a_gap : 0;
    uint32_t wkinds = 0; // two bits per slot of this wave: 1 = image B, 2 = image A
    uint32_t sflags = 0; // 1: image B hangs over the image's edge, 2: image A does, 4: some target centre of the tile is invalid
    RawDma dmaMine;      // COPIES: the first wave issues image A's, the second image B's (a lone wave both)
    uint32_t ph;         // byte phases of raw index 0 in the row to unpack next: A | B << 8 | A's step << 16 | B's step << 24
    {
        // the tile's columns of both images: logical column 0 of ring A is canonical column tile_x0 + wx0, of ring B (and
        // of the column sums) tile_x0 + wx0 + boff - dhi_t; bias column k is the target centre tile_x0 + boff - dhi_t + k
        const RawSide sideA = raw_side(g.st.img_a, g.st.stride_a, g.st.wa, tile_x0 + g.st.wx0, L.n_a4, g.st.mirror);
        const RawSide sideB = raw_side(g.st.img_b, g.st.stride_b, g.st.wb, tile_x0 + g.st.wx0 + g.st.boff - dhi_t, L.n_b4, g.st.mirror);
        const int nqa = L.n_a4 >> 2, nqb = L.n_b4 >> 2, nqh = (n_bi + 3) >> 2;
        dmaMine = raw_dma_setup(wave == 0 ? sideA : sideB, ra0);
        const uint32_t phA0 = (uint32_t)(reinterpret_cast<uintptr_t>(sideA.base) + (uintptr_t)((long long)ra0 * sideA.stride + 3LL * sideA.c0)) & 15u;
        const uint32_t phB0 = (uint32_t)(reinterpret_cast<uintptr_t>(sideB.base) + (uintptr_t)((long long)ra0 * sideB.stride + 3LL * sideB.c0)) & 15u;
        ph = phA0 | phB0 << 8 | ((uint32_t)sideA.stride & 15u) << 16 | ((uint32_t)sideB.stride & 15u) << 24;
        const int k_lo = g.st.b_lo - (tile_x0 + g.st.boff - dhi_t);
        const uint32_t k_span = g.st.b_hi >= g.st.b_lo ? (uint32_t)(g.st.b_hi - g.st.b_lo) : 0u;
        const bool k_none = g.st.b_hi < g.st.b_lo;
        if (sideB.v_lo > 0 || sideB.v_hi < L.n_b4) sflags |= 1u;
        if (sideA.v_lo > 0 || sideA.v_hi < L.n_a4) sflags |= 2u;
        if (k_none || k_lo > 0 || (uint32_t)(4 * nqh - 1 - k_lo) > k_span) sflags |= 4u;
        for (int slot = 0; slot < L.nslots; ++slot) {
            // (image A's roles `gap` places behind image B's: development knob WS_STAGE_AGAP, to put them on chosen waves)
            const int role_raw = slot * nwaves + role0;
            const int role = role_raw < L.roles_b ? role_raw : role_raw - agap >= L.roles_b ? role_raw - agap : L.roles_b + L.roles_a;
            uint4 d = make_uint4(0u, 0u, 0u, 0u);
            if (role < L.roles_b) {
                wkinds |= 1u << (2 * slot);
                const int li = SSD ? (lane & 15) : lane, row = SSD ? (lane >> 4) : 0;
                const int Qr = role * QW + row * QROW + li;
                const int Q = min(Qr, nqb - 1); // (lanes past the tile row compute on its last quad and store nothing)
                const int qraw = g.st.mirror ? nqb - 1 - Q : Q;
                d.x = (uint32_t)oRawB + 12u * (uint32_t)qraw;
                d.y = (4u * (uint32_t)((Q % NREGB) * ro_b) + 16u * (uint32_t)(Q / NREGB)) |
                      (4u * (uint32_t)((Q % NREGB) * ro_bi) + 16u * (uint32_t)(Q / NREGB)) << 16;
                d.z = (uint32_t)oG + 16u * (uint32_t)Q;
                const bool owner = (!SSD || li < QROW) && Qr < nqb;
                d.w = (owner ? kDescRing : 0u) | (owner && Q < nqh ? kDescBias : 0u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if ((uint32_t)(4 * qraw + e - sideB.v_lo) < (uint32_t)(sideB.v_hi - sideB.v_lo)) d.w |= 0x100u << e;
                    if (k_none || (uint32_t)(4 * Q + e - k_lo) > k_span) d.w |= 0x1000u << e;
                }
            } else if (role < L.roles_b + L.roles_a) {
                wkinds |= 2u << (2 * slot);
                const int Qr = (role - L.roles_b) * 64 + lane;
                const int Q = min(Qr, nqa - 1);
                const int qraw = g.st.mirror ? nqa - 1 - Q : Q;
                d.x = (uint32_t)oRawA + 12u * (uint32_t)qraw;
                d.y = 4u * (uint32_t)((Q % NREG) * ro_a) + 16u * (uint32_t)(Q / NREG);
                d.w = Qr < nqa ? kDescRing : 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((uint32_t)(4 * qraw + e - sideA.v_lo) < (uint32_t)(sideA.v_hi - sideA.v_lo)) d.w |= 0x100u << e;
            }
            lds_store128(lds0 + (uint32_t)kStage + 16u * (uint32_t)(slot * NT + tid), d);
        }
    }
    constexpr int kMul = kFuseSsd ? (CENTRED ? -2 : 510) : 0; // 2 K of the fused chain's correction term (march_fused_ssd)
    const bool lone = nwaves == 1;

    // what this wave does in a step: 1 = issues image A's copies, 2 = image B's, 4 = image B's by the lone wave, 8 = unpacks
    const uint32_t wdo = (wave == 0 ? 1u : 0u) | (wave == 1 ? 2u : 0u) | (lone ? 4u : 0u) | (wkinds != 0 ? 8u : 0u);

    // slot_u: the ring slot of row a + 1
    auto produce = [&](int a_in, int slot_in) __attribute__((always_inline)) {
        if (wdo == 0) return; // (most waves: one scalar compare per step)
        // (what follows is computed HERE, every step: hoisted to the top of the loop it would run in every wave)
        int a = a_in, slot_u = slot_in;
        asm volatile("" : "+s"(a), "+s"(slot_u));
        if ((wdo & 7u) && a + 2 < nsteps) { // row a + 2 of the strip: its bytes start their way to the raw buffer of its parity
            const uint32_t par2 = (uint32_t)(a & 1);
            if (wdo & 1u) raw_dma(lds0 + (uint32_t)oRawA + par2 * (uint32_t)kRA, dmaMine, a + 2, lane);
            if (wdo & 2u) raw_dma(lds0 + (uint32_t)oRawB + par2 * (uint32_t)kRB, dmaMine, a + 2, lane);
            if (wdo & 4u) { // (a workgroup of one wave, tiny images: image B's geometry again, every step)
                // (read through the argument segment itself: taking the address of the by-value argument copies all of it to scratch)
                const StageArgs *sp = (const StageArgs *)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(sp));
                const int dt_ = sp->nch * ND, n_b4_ = (sp->nxr * X + (HALO && PK ? 0 : WW - 1) + dt_ - 1 + 3) & ~3;
                const RawSide sb = raw_side(sp->img_b, sp->stride_b, sp->wb, tile_x0 + sp->wx0 + sp->boff - (sp->d_first + dt_ - 1), n_b4_, sp->mirror);
                raw_dma(lds0 + (uint32_t)oRawB + par2 * (uint32_t)kRB, raw_dma_setup(sb, ra0), a + 2, lane);
            }
        }
        const int iu = a + 1; // the row the chains add in the next step
        if (iu < 0 || iu >= nsteps || !(wdo & 8u)) return;
        {
            const uint32_t par = (uint32_t)(iu & 1);
            uint32_t daddr = lds0 + (uint32_t)kStage + 16u * (uint32_t)tid;
            for (uint32_t kk = wkinds; kk != 0; kk >>= 2, daddr += 16u * (uint32_t)NT) {
                const uint4 d = lds_load128(daddr);
                if (kk & 1u) {
                    // ---- image B: a quad of the ring, its column sums, its bias values
                    const uint32_t sB = (ph >> 8) & 15u;
                    uint32_t px[4];
                    raw_unpack<CENTRED>(px, lds0 + d.x + par * (uint32_t)kRB + (sB & ~3u), sB & 3u);
                    if (sflags & 1u) { // (uniform: a tile that hangs over the image's edge)
#pragma unroll
                        for (int e = 0; e < 4; ++e) px[e] &= (uint32_t)__builtin_amdgcn_sbfe((int)d.w, 8 + e, 1);
                    }
                    if (g.st.mirror) {
                        uint32_t t = px[0]; px[0] = px[3]; px[3] = t;
                        t = px[1]; px[1] = px[2]; px[2] = t;
                    }
                    const uint32_t phys = d.y & 0xffffu;
                    const uint32_t ringB_u = lds0 + (uint32_t)(kStage + L.desc_bytes) + 4u * (uint32_t)(NRA * a_w + slot_u * b_w);
                    if (d.w & kDescRing) lds_store128(ringB_u + phys, make_uint4(px[0], px[1], px[2], px[3]));
                    if constexpr (SSD) {
                        // column sums: row iu enters, row iu - WH leaves and joins the correction term.  Plain bytes: with
                        // x = 255 - b per channel, -b^2 + 510 b = 3 * 255^2 - x.x, so the sums are kept WITHOUT the constant
                        // (G'' = G - 3 * 255^2 * rows left so far) and a leaving pixel costs an xor, a dot and a subtraction;
                        // the bias values get WW times the constant back below
                        const uint4 gp = lds_load128(lds0 + d.z + (par ^ 1u) * (uint32_t)kGB);
                        uint32_t gn[4] = {gp.x, gp.y, gp.z, gp.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) gn[e] = pix_dot<CENTRED>(px[e], px[e], gn[e]);
                        if (iu >= WH) {
                            int slot_l = slot_u + 2; // row iu - WH
                            if (slot_l >= NR) slot_l -= NR;
                            const uint4 lv = lds_load128(lds0 + (uint32_t)(kStage + L.desc_bytes) + 4u * (uint32_t)(NRA * a_w + slot_l * b_w) + phys);
                            const uint32_t lp[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if constexpr (kMul == 510) {
                                    const uint32_t xc = lp[e] ^ 0x00ffffffu;
                                    gn[e] -= pix_dot<false>(xc, xc, 0u);
                                } else {
                                    gn[e] -= pix_dot<CENTRED>(lp[e], lp[e], 0u);
                                    if constexpr (kMul != 0) gn[e] += (uint32_t)kMul * pix_dot<CENTRED>(lp[e], 0x00010101u, 0u);
                                }
                            }
                        }
                        if (d.w & kDescRing) lds_store128(lds0 + d.z + par * (uint32_t)kGB, make_uint4(gn[0], gn[1], gn[2], gn[3]));
                        if (iu >= WH - 1) {
                            // 4 bias values = the sums of WW consecutive column sums each: a running prefix over 4 + WW - 1 of
                            // them, the ones past this quad from the next lanes (every lane of the wave is here: DPP reads them)
                            uint32_t pre[4 + WW - 1];
                            pre[0] = gn[0];
#pragma unroll
                            for (int m = 1; m < 4 + WW - 1; ++m) {
                                uint32_t v;
                                if (m < 4) v = gn[m];
                                else if (m / 4 == 1) v = from_lane_plus<1>(gn[m & 3]);
                                else if (m / 4 == 2) v = from_lane_plus<2>(gn[m & 3]);
                                else if (m / 4 == 3) v = from_lane_plus<3>(gn[m & 3]);
                                else if (m / 4 == 4) v = from_lane_plus<4>(gn[m & 3]);
                                else v = from_lane_plus<5>(gn[m & 3]);
                                pre[m] = pre[m - 1] + v;
                            }
                            const uint32_t cst = kMul == 510 ? ((uint32_t)(WW * 3 * 255 * 255) * (uint32_t)max(0, iu - WH + 1)) << LT : 0u;
                            uint32_t o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = ((pre[e + WW - 1] - (e ? pre[e - 1] : 0u)) << LT) + cst;
                            if (sflags & 4u) { // (uniform: a tile with target centres outside [b_lo, b_hi]: their keys are poisoned)
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] += __builtin_amdgcn_ubfe(d.w, 12 + e, 1) << 29;
                            }
                            static_assert(kPoison == 1 << 29, "the descriptor's poison bits are shifted into place");
                            if (d.w & kDescBias)
                                lds_store128(lds0 + (uint32_t)(kStage + L.desc_bytes) + 4u * (uint32_t)(NRA * a_w + NR * b_w + ((iu + WH + 1) & 1) * bi_w) + (d.y >> 16),
                                             make_uint4(o[0], o[1], o[2], o[3]));
                        }
                    }
                } else {
                    // ---- image A: a quad of the ring
                    const uint32_t sA = ph & 15u;
                    uint32_t px[4];
                    raw_unpack<CENTRED>(px, lds0 + d.x + par * (uint32_t)kRA + (sA & ~3u), sA & 3u);
                    if (sflags & 2u) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) px[e] &= (uint32_t)__builtin_amdgcn_sbfe((int)d.w, 8 + e, 1);
                    }
                    if (g.st.mirror) {
                        uint32_t t = px[0]; px[0] = px[3]; px[3] = t;
                        t = px[1]; px[1] = px[2]; px[2] = t;
                    }
                    if (d.w & kDescRing) {
                        const uint32_t at = lds0 + (uint32_t)(kStage + L.desc_bytes) + 4u * (uint32_t)(slot_u * a_w) + d.y;
                        lds_store128(at, make_uint4(px[0], px[1], px[2], px[3]));
                        if constexpr (TWIN) lds_store128(at + 4u * (uint32_t)(NR * a_w), make_uint4(~px[0], ~px[1], ~px[2], ~px[3]));
                    }
                }
            }
        }
        // the next row's byte phases: each image's own step, modulo 16
        ph = (ph & 0xffff0000u) | (((ph & 0x0f0fu) + ((ph >> 16) & 0x0f0fu)) & 0x0f0fu);
    };

    // prologue: two steps of the stages alone -- row 0 lands, then is unpacked while row 1 lands
#pragma unroll 1
    for (int a = -2; a < 0; ++a) {
        produce(a, 0);
        if (wdo & 7u) dma_wait();
        __syncthreads();
    }

    int add_slot = 0;      // ring slot of the row entering at this step   (a     mod NR)
    int sub_slot = 2 % NR; // ring slot of the row leaving at this step    (a-WH  mod NR)
    // image row of the output flushed at step a (row ys + a - WH) sits in slot (a - WH - wy0) mod NR
    int out_slot = ((-WH - g.wy0) % NR + NR) % NR;

    // 1. hand the row finished in the previous step to HBM
    auto flush = [&](int oi) __attribute__((always_inline)) {
        const int y = ys + oi - 1;
        slot_t *sl = slots + ((oi - 1) & 1) * tx;
        const uint32_t *rowA = ringA + out_slot * a_w;
        // (tx <= NT; which waves flush: see pw0)
        const int k = tid - 64 * min(g.tune_flush_wave >= 0 ? g.tune_flush_wave : pw0, nwaves - (round_up_dev(tx, 64) >> 6));
        if (k >= 0 && k < tx_out) {
            const int si = (k % X) * g.st.nxr + k / X; // slots are stored [x][r] (round 3: [r][x] -- a thread's slots as neighbours, one
                                                    // address for all eight ds_min -- saved 14 instructions a step and lost more to
                                                    // LDS bank conflicts: config 2 0.120 -> 0.125 ms)
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
                const int xo = g.st.mirror ? g.st.wa - 1 - x : x;
                float val;
                const bool none = SSD  ? (int32_t)((long long)key >> 32) >= kValidKeyBound
                                  : PK ? (uint32_t)key >= kPkNone
                                       : (int32_t)key >= kValidKeyBound;
                if (none) {
                    val = g.fallback_neg ? -(float)xo : (float)xo;
                } else {
                    const int gtag = SSD ? (int)(uint32_t)key : PK ? (int)((uint32_t)key & 0xffffu) : ((int)key & ((1 << g.tag_bits) - 1));
                    val = (float)(g.prefer_large ? (SSD ? g.d_top : g.d_hi) - gtag : g.d_lo + gtag);
                }
                // black pixel (BlockSearch.cpp:41, :105): image row y, column x, from the ring
                if (rowA[lds_phys<NREG>(k - g.st.wx0, ro_a)] == (CENTRED ? kCentre : 0u)) val = 0.0f;
                if (g.out16) g.out16[(size_t)y * g.out_pitch + xo] = (int16_t)(int)val;
                else g.out[(size_t)y * g.out_pitch + xo] = val;
                if (COST && !none) { // (a template flag: the test alone cost the hot kernel 2.7 %)
                    int32_t cst;
                    if constexpr (SSD) cst = (int32_t)((long long)key >> 32) >> LT;
                    else if constexpr (PK) cst = (int32_t)((uint32_t)key >> 16);
                    else cst = (int32_t)key >> g.tag_bits;
                    g.cost_out[(size_t)y * g.cost_pitch + xo] = cst;
                }
            }
        }
    };

    // One step = one image row down.  PHASE 0: warm-up (the window fills: rows enter, nothing comes out), 1: the
    // strip's first output row (nothing leaves yet), 2: steady state (a row enters, a row leaves, a row comes out).
    // Each phase is a loop of its own: with one loop and branches on the step number the compiler joined the three
    // variants' running sums at the bottom of the body -- 64 v_mov per thread and step (13 % of the instructions).
    auto step = [&](int a, auto phase, auto mflag) __attribute__((always_inline)) {
        constexpr int PHASE = decltype(phase)::value;
        constexpr bool MASKED = decltype(mflag)::value;
        const int oi = a - (WH - 1); // output row index inside the strip produced by this step
        if constexpr (PHASE == 2) flush(oi);

        // 2. the stages ahead: copy row a + 2, unpack row a + 1 (and sum its bias row)
        int nxt_slot = add_slot + 1;
        if (nxt_slot == NR) nxt_slot = 0;
        produce(a, nxt_slot);

        // 3. arithmetic
        if (worker) {
            const uint32_t *addA = ringA + add_slot * a_w + ia, *addB = ringB + add_slot * b_w + ib;
            const uint32_t *subA = ringA + sub_slot * a_w + ia, *subB = ringB + sub_slot * b_w + ib;
            const uint32_t *subAc = ringAc + sub_slot * a_w + ia; // (TWIN: the leaving row's reference pixels, complemented)
            slot_t *sl = slots + (oi & 1) * tx + r;
            if constexpr (PK) {
                uint32_t bestp[X];
#pragma unroll
                for (int x = 0; x < X; ++x) bestp[x] = 0xffffffffu;
                constexpr int NPA = HALO ? X : X + WW - 1, NPB = HALO ? X + ND - 1 : X + WW + ND - 2;
                uint32_t pa[NPA], pb[NPB], qa[NPA], qb[NPB];
                lds_run<NPA, NREG>(pa, addA, ro_a);
                lds_run<NPB, NREGB>(pb, addB, ro_b);
                if constexpr (PHASE == 2) {
                    lds_run<NPA, NREG>(qa, subA, ro_a);
                    lds_run<NPB, NREGB>(qb, subB, ro_b);
                }
                const PkMask mks = MASKED ? mk.fresh() : mk;
                if constexpr (HALO) march_pk_halo<X, ND, WW, PHASE, MASKED>(Vp, bestp, pa, pb, qa, qb, tagr, mks);
                else march_pk<X, ND, WW, PHASE, MASKED>(Vp, bestp, pa, pb, qa, qb, tagr, mks);
                if constexpr (PHASE >= 1) {
#pragma unroll
                    for (int x = 0; x < X; ++x) atomicMin(sl + x * g.st.nxr, bestp[x]); // ds_min_u32
                }
            } else {
                int32_t best[X];
#pragma unroll
                for (int x = 0; x < X; ++x) best[x] = INT_MAX;
                const int32_t *brow = biasr + (oi & 1) * bi_w + ib;
                if constexpr (PHASE == 0) {
                    march_row<X, ND, WW, SSD, CENTRED, +1, false>(V, best, addA, ro_a, addB, ro_b, nullptr, 0, shift);
                } else if constexpr (PHASE == 1) {
                    march_row<X, ND, WW, SSD, CENTRED, +1, true>(V, best, addA, ro_a, addB, ro_b, brow, ro_bi, shift);
                } else if constexpr (SSD && kFuseSsd && HALO) { // ... over the thread's own columns, the rest from lane + 1
                    uint32_t pa[X], pb[X + ND - 1], qa[X], qb[X + ND - 1], bi[X + ND - 1];
                    lds_run<X, NREG>(pa, addA, ro_a);
                    lds_run<X + ND - 1, NREGB>(pb, addB, ro_b);
                    lds_run<X + ND - 1, NREGB>(bi, reinterpret_cast<const uint32_t *>(brow), ro_bi);
                    lds_run<X, NREG>(qa, subAc, ro_a);
                    lds_run<X + ND - 1, NREGB>(qb, subB, ro_b);
                    march_fused_ssd_halo<X, ND, WW, CENTRED>(V, best, pa, pb, qa, qb, bi, shift);
                } else if constexpr (SSD && kFuseSsd) { // row a enters and row a - WH leaves in one chain
                    uint32_t pa[X + WW - 1], pb[X + WW + ND - 2], qa[X + WW - 1], qb[X + WW + ND - 2], bi[X + ND - 1];
                    march_load<X, ND, WW, true, true>(pa, pb, bi, addA, ro_a, addB, ro_b, brow, ro_bi);
                    lds_run<X + WW - 1, NREG>(qa, subAc, ro_a);
                    lds_run<X + WW + ND - 2, NREGB>(qb, subB, ro_b);
                    march_fused_ssd<X, ND, WW, CENTRED>(V, best, pa, pb, qa, qb, bi, shift);
                } else {
                    march_row<X, ND, WW, SSD, CENTRED, -1, false>(V, best, subA, ro_a, subB, ro_b, nullptr, 0, shift);
                    // (nothing of the entering row moves up into the leaving row's arithmetic: both rows' pixels at once
                    // cost the wide windows up to 26 spilled registers)
                    __builtin_amdgcn_sched_barrier(0);
                    march_row<X, ND, WW, SSD, CENTRED, +1, true>(V, best, addA, ro_a, addB, ro_b, brow, ro_bi, shift);
                }
                if constexpr (PHASE >= 1) {
#pragma unroll
                    for (int x = 0; x < X; ++x) {
                        const int32_t bk = best[x];
                        // no validity test here: a poisoned key is just a large one, the flush sorts it out
                        if constexpr (SSD) {
                            const uint32_t gtag = (uint32_t)(bk & (ND - 1)) | (uint32_t)ctag; // v_and_or_b32
                            const long long key = (long long)(((unsigned long long)(uint32_t)(bk | (ND - 1)) << 32) | gtag);
                            atomicMin(reinterpret_cast<long long *>(sl) + x * g.st.nxr, key); // ds_min_i64, lanes on consecutive slots
                        } else {
                            atomicMin(reinterpret_cast<int32_t *>(sl) + x * g.st.nxr, bk); // ds_min_i32
                        }
                    }
                }
            }
        }

        if (wdo & 7u) dma_wait(); // the row copies issued at the top of this step have long landed (the other waves' stores need not)
        __syncthreads();
        add_slot = nxt_slot;
        if (++sub_slot == NR) sub_slot = 0;
        if (++out_slot == NR) out_slot = 0;
    };

    // (nsteps >= WH: a strip has at least one row)
    auto run = [&](auto mflag) __attribute__((always_inline)) {
        int a = 0;
        for (; a < WH - 1; ++a) step(a, std::integral_constant<int, 0>(), mflag);
        step(a++, std::integral_constant<int, 1>(), mflag);
        for (; a < nsteps; ++a) step(a, std::integral_constant<int, 2>(), mflag);
    };
    if (PK && masked) run(std::true_type()); // (uniform per wave: every wave meets the same barriers either way)
    else run(std::false_type());
    flush(nsteps - (WH - 1)); // the last row
}

typedef void (*MarchFn)(const MarchArgs);
struct MarchEntry {
    int x, ww, wh, ssd, nd; // columns x disparities per thread, window, cost
    MarchFn fn;
    MarchFn fn_cost; // the same kernel also writing the winners' costs (right-view window sizes only)
    const char *name;
};
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
constexpr int kNDNarrow = 4;                           // the second instantiation of every window (march_shape)
// the flush hands one output column to one thread (k < tx <= NT): that needs nxr * X <= round_up(nxr * nch, 64), i.e.
// at least X d-chunks per tile (march_plan: nch >= max(8, X))
static_assert(kX <= 8, "the flush of ws_march_kernel covers a tile's columns with one pass of the workgroup: nch >= X");

#define WS_MARCH_ENTRY_ND(W, H, N, TAG)                                                                          \
    {kX, W, H, 0, N, ws_march_kernel<kX, N, W, H, false, kMaxT>, nullptr, "ws_march_kernel<sad," #W "x" #H TAG ">"}, \
    {kX, W, H, 1, N, ws_march_kernel<kX, N, W, H, true, kMaxT>, nullptr, "ws_march_kernel<ssd," #W "x" #H TAG ">"}
#define WS_MARCH_ENTRY_COST_ND(W, H, N, TAG)                                                                   \
    {kX, W, H, 0, N, ws_march_kernel<kX, N, W, H, false, kMaxT>, ws_march_kernel<kX, N, W, H, false, kMaxT, true>, \
     "ws_march_kernel<sad," #W "x" #H TAG ">"},                                                                    \
    {kX, W, H, 1, N, ws_march_kernel<kX, N, W, H, true, kMaxT>, ws_march_kernel<kX, N, W, H, true, kMaxT, true>,   \
     "ws_march_kernel<ssd," #W "x" #H TAG ">"}
// every window: left view bs x bs, right view (bs-1) x (bs-1)
#define WS_MARCH_TABLE(N, TAG)                                                                                         \
    WS_MARCH_ENTRY_ND(3, 3, N, TAG), WS_MARCH_ENTRY_ND(5, 5, N, TAG), WS_MARCH_ENTRY_ND(7, 7, N, TAG),                 \
    WS_MARCH_ENTRY_ND(9, 9, N, TAG), WS_MARCH_ENTRY_ND(11, 11, N, TAG), WS_MARCH_ENTRY_ND(13, 13, N, TAG),             \
    WS_MARCH_ENTRY_ND(15, 15, N, TAG), WS_MARCH_ENTRY_ND(17, 17, N, TAG), WS_MARCH_ENTRY_COST_ND(2, 2, N, TAG),        \
    WS_MARCH_ENTRY_COST_ND(4, 4, N, TAG), WS_MARCH_ENTRY_COST_ND(6, 6, N, TAG), WS_MARCH_ENTRY_COST_ND(8, 8, N, TAG),  \
    WS_MARCH_ENTRY_COST_ND(10, 10, N, TAG), WS_MARCH_ENTRY_COST_ND(12, 12, N, TAG),                                    \
    WS_MARCH_ENTRY_COST_ND(14, 14, N, TAG), WS_MARCH_ENTRY_COST_ND(16, 16, N, TAG)

// (Measured and dropped in round 3: packed SAD with 16 columns x 4 disparities per thread -- 6.25 instead of 7.75
// instructions per hypothesis at 9 x 9, since the chains' start-up is spread over twice the columns, and 221 VGPRs --
// took 1.11 ms instead of 0.885 at config 3: twice the LDS merges and 30 % more LDS reads per hypothesis cost more than
// the instructions saved.  16 x 8 does not fit the register file: 145 spilled registers.)
// the 4-disparities-per-thread instantiations (ws_march_nd4.hip)
const MarchEntry *march_table_narrow(int *count);
// the halo-exchange instantiations of the packed SAD kernel (march_pk_halo): 16 disparities per thread (8 packed
// running sums per column: tiles of 16 runs then hold 512 disparities in 512 threads -- with 8 per thread a range
// that wide took several d-group passes, each a round trip of the key plane), windows 7 .. 9 wide (left view) and
// 6 .. 8 (right view, with the cost-writing twin); narrower windows have little overhang to save
constexpr int kNDHalo = 16;
// ... and of the fused SSD chain (march_fused_ssd_halo), 8 disparities per thread like the plain kernel
#define WS_MARCH_HALO_SSD_ENTRY(W, H)                                                                                \
    {kX, W, H, 1, kND, ws_march_kernel<kX, kND, W, H, true, kMaxT, false, true>, nullptr, "ws_march_kernel<ssd," #W "x" #H ",halo>"}
#define WS_MARCH_HALO_SSD_ENTRY_COST(W, H)                                                                           \
    {kX, W, H, 1, kND, ws_march_kernel<kX, kND, W, H, true, kMaxT, false, true>,                                     \
     ws_march_kernel<kX, kND, W, H, true, kMaxT, true, true>, "ws_march_kernel<ssd," #W "x" #H ",halo>"}
#define WS_MARCH_HALO_ENTRY(W, H)                                                                                    \
    {kX, W, H, 0, kNDHalo, ws_march_kernel<kX, kNDHalo, W, H, false, kMaxT, false, true>, nullptr,                   \
     "ws_march_kernel<sad," #W "x" #H ",halo>"}
#define WS_MARCH_HALO_ENTRY_COST(W, H)                                                                               \
    {kX, W, H, 0, kNDHalo, ws_march_kernel<kX, kNDHalo, W, H, false, kMaxT, false, true>,                            \
     ws_march_kernel<kX, kNDHalo, W, H, false, kMaxT, true, true>, "ws_march_kernel<sad," #W "x" #H ",halo>"}

} // namespace wsamd
