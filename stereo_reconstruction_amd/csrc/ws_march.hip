// ws_march.hip -- the hot kernel of the WindowSearch path on gfx950 (CDNA4), plus the overview of
// all device code:  ws_march_kernel.h / ws_march.hip (marching kernel, tiling plan), ws_prepass.hip (dword planes for the side kernels),
// ws_border.hip (brute force, border ring, refine, varBlock), ws_smooth.hip (smoothFactor passes),
// ws_consumers.hip (warp, Reconstruction-side maps), ws_device.h (shared helpers).
//
// What the reference computes (BlockSearch.cpp:24-179): for every pixel, for every candidate
// disparity, the L2 norm of the absolute difference of two bs x bs x 3 windows, and the
// candidate with the strictly smallest value.  It re-sums the window for every (pixel, d).
//
// What runs here instead (same integers, same winner):
//   ws_march_kernel  the hot kernel.  A workgroup owns a tile of X*nxr columns and a strip of
//                    rows; thread (r, c) owns X consecutive columns and ND consecutive
//                    disparities and keeps their X*ND window sums in registers while the
//                    workgroup marches down the strip one row at a time:
//                      - the caller's CV_8UC3 rows travel HBM -> LDS as the bytes they are (LDS-DMA) and are
//                        unpacked there, two stages ahead of their use: one dword per pixel (B | G<<8 | R<<16) in
//                        an LDS ring every disparity chunk of the tile reads, zero outside the image, mirrored in x
//                        for the right view; SSD: the box sum of the squared target pixels (the part of
//                        sum (a-b)^2 that does not need a) is summed by the same lanes (ws_march_kernel.h),
//                      - per row and d a prefix chain of v_sad_u8 / v_dot4_u32_u8 (one
//                        instruction per pixel pair, 3 channels at once) gives all horizontal
//                        window sums by differences; the row leaving the window is removed the
//                        same way (sliding box filter, exact in integers),
//                      - the running minimum is a single signed v_min on (cost << k | tie tag),
//                      - the d-chunks of a pixel meet through one ds_min_u64 per thread and row.
//                    A left-view search with smoothFactor 1 is this ONE launch (it writes the border zeros too).
//   ws_pack_kernel   BGR bytes -> zero padded dword planes, only for the kernels below that read planes.
//   ws_ring_kernel   right-view border ring (clipped windows): a small marching kernel, lanes over d.
//   ws_linear_kernel LinearSearch through LDS.
//   ws_generic_kernel  literal per-pixel brute force: window sizes without a marching instantiation
//                    and LinearSearch ranges beyond 4096.
//   ws_refine_*      sub-pixel parabola (extension).
//
// No MFMA: the hot loop is a stencil + reduction on bytes, bounded by VALU issue and LDS, see
// DESIGN.md.  Wave64 throughout; nothing here assumes 32-wide warps.
#include "ws_march_kernel.h"

#include <string.h>

#include <mutex>
#include <utility>
#include <vector>

namespace wsamd {

// ---- instantiation table -----------------------------------------------------------------
constexpr int kMaxChunks = 64; // at most 512 disparities per tile and pass (tools/time_calls.py)
constexpr int kMinXRuns = 4; // narrowest tile: 4 x-runs = 32 columns (D up to 1536)

// Every window comes with 8 and with 4 disparities per thread.  8 (64 running sums, 200..250 VGPRs, two waves per
// SIMD) is the fastest search of a config-2-like pair on an idle chip; 4 (~125 VGPRs for SAD, four waves per SIMD)
// reads more of plane B per hypothesis but lets the workgroups of two searches -- two contexts taking a queue of
// pairs alternately, INTEGRATION.md -- share a CU.  The table behind march_nd's rule: profiles/r03/nd_grid.txt.
static const MarchEntry kMarchWide[] = {WS_MARCH_TABLE(kND, "")};
static const MarchEntry kMarchHalo[] = {WS_MARCH_HALO_ENTRY(7, 7), WS_MARCH_HALO_ENTRY(9, 9), WS_MARCH_HALO_ENTRY_COST(6, 6),
                                       WS_MARCH_HALO_ENTRY_COST(8, 8),
#if WS_FUSE
                                       WS_MARCH_HALO_SSD_ENTRY(7, 7), WS_MARCH_HALO_SSD_ENTRY(9, 9),
                                       WS_MARCH_HALO_SSD_ENTRY_COST(6, 6), WS_MARCH_HALO_SSD_ENTRY_COST(8, 8),
#endif
};

// The thread shapes a search can run with: X columns x ND disparities per thread.
struct MarchShape { int x, nd; };
constexpr MarchShape kShapeWide{kX, kND}, kShapeNarrow{kX, kNDNarrow};
static bool same_shape(MarchShape a, MarchShape b) { return a.x == b.x && a.nd == b.nd; }
static int min_chunks(MarchShape s) { return s.x > 8 ? s.x : 8; } // (the flush needs nch >= X, ws_march_kernel.h)
// d-chunks a tile holds at most: 64 (512 disparities with 8 per thread; more would leave too few columns per tile);
// a shape with more than 8 columns per thread would keep 4 x-runs with 128 chunks
static int max_chunks(MarchShape s) { return s.x > 8 ? kMaxT / kMinXRuns : kMaxChunks; }

// Row times a search would take with a thread shape, in units of one row step of the 8 x 8 kernel, on a chip of
// g_model_cus CUs with one workgroup per CU at a time -- the strip planner's own model (march_plan): the chip works through
// ceil(workgroups / CUs) rounds of strips, a strip of R rows costs R + (wh - 1) / 2 + 3 row steps.  A row step of the
// 8 x 4 kernel covers half the hypotheses per workgroup and costs 0.62 of an 8 x 8 one (measured: 1.19 .. 1.29 times the
// time per hypothesis over windows 5 .. 17 at D = 512, profiles/r03/nd_grid.txt).
// the chip the thread-shape rule plans for: the first context's device (ws_create), 256 CUs without one (ws_plan)
static int g_model_cus = 256;
void march_set_num_cus(int n) { if (n > 0) g_model_cus = n; }

static double march_model_cost(const Canon &c, MarchShape sh)
{
    const int dcount = c.d_hi - c.d_lo + 1, out_w = c.ox1 - c.ox0, out_h = c.oy1 - c.oy0;
    const int nch_total = ceil_div(dcount, sh.nd), passes = ceil_div(nch_total, max_chunks(sh));
    int nch = ceil_div(nch_total, passes);
    if (nch < min_chunks(sh)) nch = min_chunks(sh);
    int nxr = kMaxT / nch;
    const int need = ceil_div(out_w, sh.x);
    if (nxr > need) nxr = need;
    if (nxr < kMinXRuns) nxr = kMinXRuns;
    const int tiles = ceil_div(out_w, nxr * sh.x);
    double best = 0.0;
    for (int sc = 1; sc <= out_h; ++sc) {
        const int rows = ceil_div(out_h, sc), st = ceil_div(out_h, rows);
        if (st != sc) continue;
        const double cost = ceil_div(tiles * st, g_model_cus) * (rows + 0.5 * (c.wh - 1) + 3.0);
        if (sc == 1 || cost < best) best = cost;
    }
    return best * passes * (same_shape(sh, kShapeNarrow) ? 0.62 : 1.0);
}

// Columns x disparities per thread for this search: a function of the canonical problem alone (not of the device), so
// that everything that reads the marching kernel's planes afterwards (smoothFactor, sub-pixel refine) derives the same
// key layout.  Round 2 chose from a hand-made table of thresholds taken at one image size.  Now: whichever
// instantiation the planner's own cost model gives fewer row times.  The 8 x 8 kernel does a hypothesis with ~20 %
// fewer instructions than the 8 x 4 one (shorter target-image runs per hypothesis) and wins wherever both fill the chip
// alike; 8 x 4 makes tiles half as wide for the same disparity range, which pays when the range is narrow (D <= ~128 at
// 1500 columns: the 8 x 8 tiles are then so wide that strips get short and the window's warm-up rows dear) or the image
// small.  Checked against profiles/r03/nd_grid.txt (1500 x 1000, windows 5 .. 17,
// D = 128 / 256 / 512, both costs, either instantiation forced) and at 900 x 750 and 2964 x 1988
// (profiles/r03/nd_rule_check.txt).
static MarchShape march_shape(const Canon &c)
{
    static const MarchShape forced = [] {
        const char *e = getenv("WS_MARCH_ND"); // development knob: 8 or 4
        MarchShape f{0, 0};
        if (e && atoi(e) == kND) f = kShapeWide;
        else if (e && atoi(e) == kNDNarrow) f = kShapeNarrow;
        return f;
    }();
    if (forced.x) return forced;
    if (kND == kNDNarrow) return kShapeWide;
    if (c.ox1 <= c.ox0 || c.oy1 <= c.oy0 || c.d_hi < c.d_lo) return kShapeWide;
    // (asked a dozen times per search -- by the planner, the launchers, everything that needs the key layout -- and the
    // model walks every strip count: remembered per thread for the last few problems, or a 30 us search would spend
    // 50 us of host time on it)
    struct Memo { int key[9]; MarchShape shape; };
    thread_local Memo memo[4] = {};
    thread_local int next = 0;
    const int key[9] = {c.ox1 - c.ox0, c.oy1 - c.oy0, c.d_hi - c.d_lo + 1, c.ww, c.wh, c.ssd, 1, g_model_cus, 0};
    for (const Memo &m : memo)
        if (!memcmp(m.key, key, sizeof key)) return m.shape;
    const MarchShape sh = march_model_cost(c, kShapeNarrow) < march_model_cost(c, kShapeWide) ? kShapeNarrow : kShapeWide;
    memcpy(memo[next].key, key, sizeof key);
    memo[next].shape = sh;
    next = (next + 1) & 3;
    return sh;
}
static int march_nd(const Canon &c) { return march_shape(c).nd; }

static const MarchEntry *find_march(const Canon &c)
{
    const MarchShape sh = march_shape(c);
    int n = (int)(sizeof kMarchWide / sizeof kMarchWide[0]);
    const MarchEntry *t = kMarchWide;
    if (same_shape(sh, kShapeNarrow)) t = march_table_narrow(&n);
    for (int i = 0; i < n; ++i)
        if (t[i].x == sh.x && t[i].ww == c.ww && t[i].wh == c.wh && t[i].ssd == c.ssd && t[i].nd == sh.nd) return &t[i];
    return nullptr;
}

// the halo-exchange twin of the packed SAD kernel for this window, if there is one (and the wide shape was chosen)
static const MarchEntry *find_march_halo(const Canon &c)
{
    static const bool off = [] {
        const char *e = getenv("WS_MARCH_HALO"); // development knob: 0 = never
        return e && atoi(e) == 0;
    }();
    static const bool ssd_off = [] {
        const char *e = getenv("WS_MARCH_HALO_SSD"); // development knob: 0 = not for SSD
        return e && atoi(e) == 0;
    }();
    if (off || (c.ssd && ssd_off) || !same_shape(march_shape(c), kShapeWide)) return nullptr;
    for (const MarchEntry &e : kMarchHalo)
        if (e.ww == c.ww && e.wh == c.wh && e.ssd == c.ssd) return &e;
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
    if (!c.ssd && march_pk_window(c.ww, c.wh)) return dcount <= 65536; // packed SAD: 16-bit cost, 16-bit global tie tag
    const long long worst = c.ssd ? 2LL * c.ww * c.wh * 3 * (march_centred(c) ? 128 * 128 : 255 * 255) * march_nd(c)
                                  : ((long long)c.ww * c.wh * 3 * 255) << tag_bits_for(c);
    return worst < (long long)kValidKeyBound;
}

static int march_slots_per_cu(const Canon &c, int nd, int threads, bool halo)
{
    static const int forced = [] {
        const char *e = getenv("WS_PLAN_SLOTS"); // development knob
        return e ? atoi(e) : 0;
    }();
    if (forced > 0) return forced;
    const MarchEntry *e = halo ? find_march_halo(c) : find_march(c);
    // (without a device: the 8 x 4 kernels of packed SAD up to 6 x 6 and of SSD up to 3 x 3 stay within 128 VGPRs)
    // every instantiation runs two waves per SIMD at least: two workgroups of 256 threads share a CU
    const int guess = threads <= 256 || (same_shape(march_shape(c), kShapeNarrow) && threads <= 512 && (c.ssd ? c.ww <= 3 : c.ww <= 6)) ? 2 : 1;
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

static bool march_plan_threads(const Canon &c, int num_cus, int tune_nxr, int tune_strip_rows, int tune_threads, MarchLaunch *out);

// Workgroups of 512 threads (two waves per SIMD, one workgroup per CU) or of 256 (one wave per SIMD each, two
// workgroups per CU with barriers of their own: while one waits for its slowest wave the other one's wave has the
// SIMD).  Measured (profiles/r03/threads256.txt, sweep_tiles_config3.csv): 256 is 6 % faster alone and 11 % with two
// searches in flight at config 3, 2 .. 9 % faster in flight everywhere else, but 4 .. 6 % slower alone when the whole
// search is one round of workgroups (config 2: its narrower tiles copy 1.7 x the target-row bytes per column and
// there is no second round to hide that behind).  So: 256 when the search is more than one and a half rounds of
// 512-thread workgroups, else 512; a caller that keeps a queue of pairs in flight on two contexts may ask for 256 throughout
// (ws_set_tuning, threads = 256).
bool march_plan(const Canon &c, int num_cus, int tune_nxr, int tune_strip_rows, int tune_threads,
                MarchLaunch *out)
{
    static const int env_threads = [] {
        const char *e = getenv("WS_PLAN_THREADS"); // development knob
        return e ? atoi(e) : 0;
    }();
    if (tune_threads <= 0 && env_threads >= 64) tune_threads = env_threads;
    if (!march_plan_threads(c, num_cus, tune_nxr, tune_strip_rows, tune_threads, out)) return false;
    if (tune_threads > 0 || tune_nxr > 0 || tune_strip_rows > 0) return true;
    MarchLaunch half{};
    // (halo-exchange plans trade d-group passes for runs per tile: their passes run back to back and count as rounds,
    // and the smaller workgroup has more of them by construction.  Config 3, 512 -> 256 threads: 3.94 -> 4.13 * 10^6
    // Mdisp/s with two searches in flight, 3.85 -> 3.52 alone -- gpurun_out/r3_halo.txt)
    const int launches = out->halo ? out->passes : 1;
    if (out->threads == kMaxT && 2 * out->tiles * out->strips * launches >= 3 * num_cus &&
        march_plan_threads(c, num_cus, 0, 0, kMaxT / 2, &half) && (half.passes == out->passes || (half.halo && out->halo)))
        *out = half;
    return true;
}

static bool march_plan_threads(const Canon &c, int num_cus, int tune_nxr, int tune_strip_rows, int tune_threads, MarchLaunch *out)
{
    if (!march_supported(c)) return false;
    MarchLaunch m{};
    const MarchShape sh = march_shape(c);
    const int nd = sh.nd, X = sh.x;
    m.x_per_thread = X;
    m.nd_per_thread = nd;
    m.max_threads = kMaxT;
    const int dcount = c.d_hi - c.d_lo + 1;
    const int out_w = c.ox1 - c.ox0, out_h = c.oy1 - c.oy0;
    // d-chunks per tile.  One tile holds at most kMaxChunks chunks (wider disparity ranges would
    // leave too few columns per tile); beyond that the range is cut into equal d-group passes that
    // meet in a plane of keys.
    const int nch_total = ceil_div(dcount, nd);
    static const int forced_chunks = [] {
        const char *e = getenv("WS_MAX_CHUNKS"); // development knob
        const int v = e ? atoi(e) : 0;
        return v >= 8 && v <= kMaxT / kMinXRuns ? v : 0;
    }();
    int maxt = kMaxT;
    if (tune_threads >= 64 && tune_threads < kMaxT) maxt = tune_threads / 64 * 64;
    m.passes = ceil_div(nch_total, forced_chunks ? forced_chunks : max_chunks(sh));
    m.nch = ceil_div(nch_total, m.passes);
    if (m.nch < min_chunks(sh)) m.nch = min_chunks(sh);
    int nxr = maxt / m.nch;
    if (tune_nxr > 0 && tune_nxr < nxr) nxr = tune_nxr;
    const int need = ceil_div(out_w, X); // no point in tiles wider than the image
    if (nxr > need) nxr = need;
    if (nxr < kMinXRuns) nxr = kMinXRuns;
    // the rest of a plan once (halo, passes, nch, nxr) are set: tiles, strips, LDS; returns its modelled cost in row
    // steps of the plain kernel (0 = does not fit)
    auto complete = [&](MarchLaunch &p, int runs) -> double {
        if (runs * p.nch > kMaxT) return 0.0;
        p.nxr = runs;
        p.threads = round_up(runs * p.nch, 64);
        const int tx = runs * X;
        p.tile_cols = p.halo ? tx - X : tx;
        p.tiles = ceil_div(out_w, p.tile_cols);
        // One workgroup per CU at a time (its registers fill the CU).  A strip of R rows costs about
        // R + (wh - 1) / 2 + 3 row times (the wh - 1 warm-up rows only add, the prologue is worth ~3 rows) and the
        // chip works through ceil(workgroups / CUs) rounds of them: take the strip count with the cheapest total
        // (config 3's own sweep, profiles/r01/sweep_tiles_config3.csv, has its minimum where this puts it).
        // (With 4 disparities per thread and a window up to 9 x 9 the kernel stays within 128 VGPRs: two
        // workgroups share a CU, four waves per SIMD, and a round is twice as many workgroups -- the runtime's
        // occupancy figure where a device is there to ask, that rule of thumb for ws_plan without one.)
        const int pnd = p.nd_per_thread;
        p.lds_bytes = (size_t)march_lds_layout(X, pnd, c.ww, c.wh, c.ssd != 0, p.halo && !c.ssd && march_pk_window(c.ww, c.wh), runs, p.nch).bytes;
        if (p.lds_bytes == 0 || p.lds_bytes > 160 * 1024) return 0.0; // (0: a tile row wider than the kernel's stage area)
        // (the registers' answer, capped by what the CU's 160 KB of LDS hold: the kernel's LDS is dynamic, the runtime is
        // asked without it -- and since round 4 a workgroup's LDS carries the stage area and the descriptors as well)
        const int slots = std::max(1, std::min(march_slots_per_cu(c, pnd, p.threads, p.halo != 0), (int)(160 * 1024 / p.lds_bytes)));
        auto strip_cost = [&](int st, int rows) { return ceil_div(p.tiles * st, num_cus * slots) * (rows + 0.5 * (c.wh - 1) + 3.0); };
        int strips = 1;
        if (tune_strip_rows > 0) {
            strips = ceil_div(out_h, tune_strip_rows);
        } else {
            double best_cost = 0.0;
            for (int sc = 1; sc <= out_h; ++sc) {
                const int rows = ceil_div(out_h, sc), st = ceil_div(out_h, rows);
                if (st != sc) continue; // (the same strips as a smaller count already seen)
                const double cost = strip_cost(st, rows);
                if (sc == 1 || cost < best_cost) { best_cost = cost; strips = sc; }
            }
            if (strips > out_h) strips = out_h;
        }
        p.strip_rows = ceil_div(out_h, strips);
        p.strips = ceil_div(out_h, p.strip_rows);
        // a row step of the halo-exchange kernel against the plain one's, from the instruction counts (march_pk_halo);
        // every d-group pass beyond the first ~2 % for the key plane's round trip (gpurun_out/r3_chunks.txt)
        // (6.19 against 7.52 instructions per hypothesis at 9 x 9, profiles/r03/isa_op_histogram.txt; a thread of the halo
        // kernels carries kNDHalo / nd times the hypotheses)
        const double step = !p.halo ? 1.0
                            : c.ssd ? (c.ww >= 8 ? 0.87 : 0.93) * pnd / nd // (measured: config 5, 9 x 9: -6.7 %; config 2, 7 x 7: 50 instead of 56
                                                                           // instructions per disparity and step, 0.1205 vs 0.1208 ms with 59- instead of 54-row strips)
                                    : (c.ww >= 9 ? 0.82 : c.ww == 8 ? 0.83 : c.ww == 7 ? 0.85 : 0.87) * pnd / nd;
        return p.passes * (1.0 + 0.02 * (p.passes - 1)) * step * strip_cost(p.strips, p.strip_rows);
    };
    m.halo = 0;
    MarchLaunch best = m;
    const double plain_cost = complete(best, nxr);
    // Packed SAD with the halo exchange: the last run of a tile only feeds its neighbour, so tiles want MANY runs --
    // 16 (a row of DPP lanes; 1/16 of the threads lost) or 8, rather than the 4 .. 8 a wide disparity range leaves
    // above.  The range is cut into more d-group passes instead: runs x (threads / runs) chunks per pass.  Taken when
    // the model says it is cheaper than the plain plan.
    // (a caller's tile width of 8 or 16 runs -- tools/sweep_tiles.py -- means the halo kernel with that many runs; any
    // other width the plain one)
    double best_cost = plain_cost;
    const bool tuned_halo = tune_nxr == 16 || tune_nxr == 8;
    if (find_march_halo(c) && (tune_nxr <= 0 || tuned_halo) && !forced_chunks) {
        if (tuned_halo) best_cost = 0.0;
        for (int hx : {16, 8}) { // (powers of two: lane + 1 is the next run inside a DPP row)
            if (tuned_halo && hx != tune_nxr) continue;
            if (need + 1 < hx && hx > 8) continue; // (tiles wider than the image)
            const int hch = maxt / hx;             // chunks per pass that fill the workgroup
            if (hch < min_chunks(sh)) continue;
            MarchLaunch h = m;
            h.halo = 1;
            const int hnd = find_march_halo(c)->nd; // 16 for the packed SAD kernels, 8 for the SSD ones
            h.nd_per_thread = hnd;
            const int hch_total = ceil_div(dcount, hnd);
            h.passes = ceil_div(hch_total, hch);
            h.nch = ceil_div(hch_total, h.passes);
            if (h.nch < min_chunks(sh)) h.nch = min_chunks(sh);
            const double cost = complete(h, hx);
            if (cost > 0.0 && (best_cost <= 0.0 || cost < best_cost)) {
                best = h;
                best_cost = cost;
            }
        }
    }
    if (best_cost <= 0.0 && plain_cost > 0.0) { // (the tuned halo plan does not fit: the plain one)
        best_cost = plain_cost;
    }
    if (best_cost <= 0.0) return false;
    *out = best;
    return true;
}

static int aligned_pad(int base)
{
    // smallest pad >= max(0, -base) that puts column `base` of the image on a 16-byte boundary
    int pad = base < 0 ? -base : 0;
    while (((base + pad) & 3) != 0) ++pad;
    return pad;
}

void march_plane_geometry(const Canon &c, const MarchLaunch &m, Plane *a, Plane *b)
{
    const int tx = m.nxr * m.x_per_thread, dt = m.nch * m.nd_per_thread;
    const int dhi_t = c.d_lo + m.passes * dt - 1; // the last pass reaches furthest to the left
    const int n_a = tx + c.ww - 1, n_b = tx + c.ww + m.passes * dt - 2;
    // first column each tile row copy starts at (tile 0); tiles advance by tile_cols (a multiple of 8)
    const int base_a = c.ox0 + c.wx0;
    const int base_b = c.ox0 + c.wx0 + c.boff - dhi_t;
    const int last = (m.tiles - 1) * m.tile_cols;
    a->pad = aligned_pad(base_a);
    a->pitch = round_up(std::max(base_a + last + round_up(n_a, 4), c.wa) + a->pad + 4, 64);
    b->pad = aligned_pad(base_b);
    b->pitch = round_up(std::max(base_b + last + round_up(n_b, 4), c.wb) + b->pad + 4, 64);
}


const char *march_kernel_name(const Canon &c, const MarchLaunch &m)
{
    const MarchEntry *e = m.halo ? find_march_halo(c) : find_march(c);
    return e ? e->name : "";
}

hipError_t launch_march(const Canon &c, const MarchLaunch &m, const uint8_t *img_a, int stride_a, const uint8_t *img_b, int stride_b,
                        float *out, int16_t *out16, int out_pitch, int border, int out_w, int out_h, void *keys, int keys_pitch,
                        int32_t *cost_out, int cost_pitch, hipStream_t s)
{
    const MarchEntry *e = m.halo ? find_march_halo(c) : find_march(c);
    if (!e) return hipErrorInvalidValue;
    MarchArgs g{};
    g.st.img_a = img_a;
    g.st.stride_a = stride_a;
    g.st.img_b = img_b;
    g.st.stride_b = stride_b;
    g.st.wb = c.wb;
    g.out = out;
    g.out16 = out16;
    g.out_pitch = out_pitch;
    g.border = border;
    g.out_w = out_w;
    g.out_h = out_h;
    g.st.wa = c.wa;
    g.st.nxr = m.nxr;
    g.st.nch = m.nch;
    g.st.wx0 = c.wx0;
    g.wy0 = c.wy0;
    g.st.boff = c.boff;
    g.d_lo = c.d_lo;
    g.d_hi = c.d_hi;
    g.d_top = c.d_lo + m.passes * m.nch * m.nd_per_thread - 1;
    g.st.b_lo = c.b_lo;
    g.st.b_hi = c.b_hi;
    g.tag_bits = tag_bits_for(c);
    g.ox0 = c.ox0;
    g.ox1 = c.ox1;
    g.oy0 = c.oy0;
    g.oy1 = c.oy1;
    g.strip_rows = m.strip_rows;
    g.tiles = m.tiles;
    g.tile_stride = m.tile_cols;
    g.strips = m.strips;
    g.prefer_large = c.prefer_large;
    g.st.mirror = c.mirror;
    g.fallback_neg = c.fallback_neg;
    static const int env_prod = [] { const char *v = getenv("WS_STAGE_WAVE"); return v ? atoi(v) : -1; }();  // development knobs
    static const int env_flush = [] { const char *v = getenv("WS_FLUSH_WAVE"); return v ? atoi(v) : -1; }();
    g.tune_prod_wave = env_prod;
    g.tune_flush_wave = env_flush;
    static const int env_agap = [] { const char *v = getenv("WS_STAGE_AGAP"); return v ? atoi(v) : 0; }();
    g.tune_a_gap = env_agap;
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
        g.st.d_first = c.d_lo + pass * m.nch * m.nd_per_thread;
        g.pass_mode = m.passes == 1 ? 0 : pass == 0 ? 1 : pass == m.passes - 1 ? 3 : 2;
        hipLaunchKernelGGL(fn, grid, dim3(m.threads), m.lds_bytes, s, g);
    }
    return hipGetLastError();
}

} // namespace wsamd
