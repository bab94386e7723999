// ws_capi.cpp -- the C-ABI of include/ws_stereo.h: argument checks that stand in for the
// reference's cv::Exception paths, reduction of the three reference methods to the canonical
// search (ws_kernels.h), scratch / staging memory owned by the context, and the Middlebury
// plumbing (PFM, calib.txt, evaldisp).  Compiled with hipcc; no compute happens on the host.
#include "../../include/ws_stereo.h"
#include "ws_kernels.h"

#include <math.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace wsamd;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct HostBuf { // pinned host memory of the library's own (hipHostMalloc)
    uint8_t *p = nullptr;
    size_t cap = 0;
};

struct Job { // one pair in flight on the batched host path
    uint8_t *d_in = nullptr; // left image, then right image (rows with the caller's stride)
    float *d_out = nullptr;
    int16_t *d_out16 = nullptr;
    size_t in_cap = 0, out_cap = 0; // bytes of d_in; elements of d_out / d_out16
    int wire = 0;            // the wire format this pair's map comes down in
    hipEvent_t ev_h2d = nullptr, ev_done = nullptr;
    void *user_out = nullptr;
    int w = 0, h = 0, out_stride = 0, dtype = 0;
    bool pending = false; // searched (or being searched), result not yet on its way to user_out
    HostBuf h_left, h_right; // gathered rows of images that do not cross as one span (gather_rows), or their stage
    HostBuf h_out;           // stage of a pageable map (HostSpan)
    int out_span = -1;       // index of this pair's output span in ws_context::batch_spans
};

thread_local std::string g_create_error; // ws_last_error(NULL): why the last ws_create on this thread failed

// development knob WS_HOST_TRACE=1: where the host's time goes inside a boundary call (stderr, microseconds since the call began)
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    std::string line;
    HostTrace() : on([] { static const bool v = [] { const char *e = getenv("WS_HOST_TRACE"); return e && atoi(e) == 1; }(); return v; }()), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what, int k = -1)
    {
        if (!on) return;
        char buf[64];
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (k >= 0) snprintf(buf, sizeof buf, " %s%d=%.0f", what, k, us); else snprintf(buf, sizeof buf, " %s=%.0f", what, us);
        line += buf;
    }
    ~HostTrace() { if (on && !line.empty()) fprintf(stderr, "[ws host trace, us]%s\n", line.c_str()); }
};

// rows of `width_bytes` between buffers with row pitches: one linear copy when both sides are dense
// (the runtime's 2-D path is slow, very slow for row lengths that are not a multiple of 4 bytes)
hipError_t copy_rows(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows,
                     hipMemcpyKind kind, hipStream_t s)
{
    if (dpitch == width_bytes && spitch == width_bytes) return hipMemcpyAsync(dst, src, width_bytes * rows, kind, s);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, kind, s);
}

hipError_t host_ensure(HostBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return hipSuccess;
    if (b.p) (void)hipHostFree(b.p);
    b.p = nullptr; b.cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&b.p), want, hipHostMallocDefault);
    if (e == hipSuccess) b.cap = want;
    return e;
}


// ---- caller host buffers ----------------------------------------------------------------------------------------
// This library registers NO caller memory (no hipHostRegister / hipHostUnregister anywhere in it).  Round 2 registered
// the caller's buffers for the duration of a call; round 3 found what that costs inside somebody else's process -- the
// runtime abort()s on an unregister of a pointer that lies inside another live registration (rocclr device.cpp:373,
// tools/ubench/hostreg_probe.hip, profiles/r03/hostreg_probe.txt), and two full test runs ended in a GPU memory fault on
// a host heap page whose cause was never proven -- and made it opt-in; round 4 removed it: the 16-bit
// wire format below wins back more than the registration saved.  How bytes cross now:
//   * pageable memory (a cv::Mat, a numpy array) crosses through pinned staging memory of the library's own
//     (hipHostMalloc): one host copy each way, on a small pool of threads, band by band beside the transfers;
//   * memory the runtime already knows at both ends -- the caller's own hipHostMalloc / hipHostRegister, a framework's
//     pinned allocator -- is used as it is, never registered or released here;
//   * a range the runtime knows only in part goes through the stage (a direct copy across its edge would be refused).
// Never through the runtime's pageable copy path: it blocks the calling thread for the whole transfer
// (profiles/r02/pcie_probe.txt).
//
// WIRE FORMAT of a disparity map: every value a search stores is an integer in [-w, max(maxDisparity, w)]
// (BlockSearch.cpp:33,82,174; LinearSearch.cpp:53) unless the sub-pixel extension is on.  So the map crosses PCIe as
// 16-bit integers (the search kernels store them: GenericArgs::out16) whenever the bounds fit, and is widened to the
// caller's CV_32F / CV_64F inside the stage -> caller copy the pool already performs: 2 instead of 4 / 8 bytes per
// pixel on the bus (config 2, CV_64F: 3 MB instead of 12), exact.  Maps that are not integers (sub-pixel) or that
// other kernels read back (smoothFactor, varBlock) cross as float32; doubles never cross.
bool runtime_knows(uintptr_t q)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, reinterpret_cast<const void *>(q)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type != hipMemoryTypeUnregistered;
}

enum Wire { kWireSame = 0, kWireI16 = 1, kWireF32 = 2 }; // what sits in the stage: the caller's own bytes, int16, float32

// One caller buffer for the duration of a call (or of a batch): how its bytes cross.
struct HostSpan {
    enum How { kUnused, kOurs /* (rounds 2-3: registered by this library; never set any more) */, kCallerPinned, kStaged };
    uint8_t *p = nullptr;
    size_t n = 0;
    How how = kUnused;
    HostBuf *stage = nullptr; // where its bytes cross if they cannot cross directly (set by the call site, always)
    bool loaded = false;      // uploads: the stage holds the caller's bytes
    // a download that went to the stage: `rows` rows of `row_elems` elements, dense in the stage from byte stage_off on
    // in wire format, to the caller's buffer from byte host_off on, rows host_pitch bytes apart, in elements of esz bytes
    struct Seg { size_t stage_off, host_off, row_elems, rows, host_pitch; int wire; int esz; };
    std::vector<Seg> down;    // handed to the caller by spans_finish / span_scatter_seg
    const char *why = "";     // kStaged: the reason (tests, ws_last_host_paths)
};

// Classify the buffers of one call.  Spans with p == nullptr or n == 0 stay kUnused.
void spans_attach(HostSpan *sp, int count)
{
    for (int i = 0; i < count; ++i) {
        if (!sp[i].p || !sp[i].n) continue;
        const uintptr_t a = reinterpret_cast<uintptr_t>(sp[i].p);
        const bool k0 = runtime_knows(a), k1 = runtime_knows(a + sp[i].n - 1);
        if (k0 && k1) {
            sp[i].how = HostSpan::kCallerPinned;
        } else {
            sp[i].how = HostSpan::kStaged;
            sp[i].why = (k0 || k1) ? "the runtime knows a part of the range (registered or allocated by the caller)"
                                   : "pageable memory: this library registers no caller memory";
        }
    }
}

// The stages' host copies: several threads for big buffers (one core moves ~12 GB/s, PCIe 50: a 9 MB image pair would
// spend longer in memcpy than on the bus).  A small pool of helper threads, started at the first big copy and shared by
// all contexts (one copy at a time uses it) -- a banded call copies a megabyte at a time, too little to start threads for
// (tools/pool_stress.cpp runs this class under ThreadSanitizer).  A job is a run of ELEMENTS moved as they are or
// widened on the way (the wire formats above): int16 -> float / double, float -> double.
enum CopyKind { kCopyBytes, kCopyI16F32, kCopyI16F64, kCopyF32F64 };

// Streaming forms (AVX2, non-temporal stores) for the copies that WIDEN TO DOUBLES -- the one host copy that writes far
// more than it reads (config 2, CV_64F: 3 MB of int16 in, 12 MB of doubles out): an ordinary store first reads the line
// it overwrites.  A/B on one GPU box's host (EPYC 9575F, 8 copy threads, 3 x 30 calls each, profiles/r04/host_trace.txt):
// the CV_64F call 0.455 -> 0.437 ms with every copy streaming, but the CV_32F call 0.400 -> 0.420 -- the stage copies
// and the float map are better off in the cache, where the copy engine and the caller find them.  WS_COPY_STREAM=0
// turns the streaming forms off, =2 applies them to every copy.
#if defined(__x86_64__)
#define WS_AVX2 __attribute__((target("avx2")))
WS_AVX2 static void stream_bytes(uint8_t *dst, const uint8_t *src, size_t n)
{
    const size_t head = (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31;
    if (n < 256 + head) { memcpy(dst, src, n); return; }
    memcpy(dst, src, head);
    size_t i = head;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}
WS_AVX2 static void stream_i16_f64(double *d, const int16_t *s, size_t n)
{
    size_t i = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(d + i) & 31); ++i) d[i] = (double)s[i];
    for (; i + 8 <= n; i += 8) {
        const __m256i v = _mm256_cvtepi16_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i)));
        _mm256_stream_pd(d + i, _mm256_cvtepi32_pd(_mm256_castsi256_si128(v)));
        _mm256_stream_pd(d + i + 4, _mm256_cvtepi32_pd(_mm256_extracti128_si256(v, 1)));
    }
    for (; i < n; ++i) d[i] = (double)s[i];
    _mm_sfence();
}
WS_AVX2 static void stream_i16_f32(float *d, const int16_t *s, size_t n)
{
    size_t i = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(d + i) & 31); ++i) d[i] = (float)s[i];
    for (; i + 8 <= n; i += 8)
        _mm256_stream_ps(d + i, _mm256_cvtepi32_ps(_mm256_cvtepi16_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i)))));
    for (; i < n; ++i) d[i] = (float)s[i];
    _mm_sfence();
}
WS_AVX2 static void stream_f32_f64(double *d, const float *s, size_t n)
{
    size_t i = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(d + i) & 31); ++i) d[i] = (double)s[i];
    for (; i + 4 <= n; i += 4) _mm256_stream_pd(d + i, _mm256_cvtps_pd(_mm_loadu_ps(s + i)));
    for (; i < n; ++i) d[i] = (double)s[i];
    _mm_sfence();
}
static int stream_mode() // 0 = never, 1 = the copies that widen to doubles, 2 = every copy
{
    static const int v = [] {
        if (!__builtin_cpu_supports("avx2")) return 0;
        const char *e = getenv("WS_COPY_STREAM");
        return e ? std::max(0, std::min(2, atoi(e))) : 1;
    }();
    return v;
}
#else
static int stream_mode() { return 0; }
static void stream_bytes(uint8_t *, const uint8_t *, size_t) {}
static void stream_i16_f64(double *, const int16_t *, size_t) {}
static void stream_i16_f32(float *, const int16_t *, size_t) {}
static void stream_f32_f64(double *, const float *, size_t) {}
#endif

static void copy_piece(uint8_t *dst, const uint8_t *src, size_t first, size_t count, CopyKind kind)
{
    const int sm = stream_mode();
    const bool fast = count >= 4096 && (sm == 2 || (sm == 1 && (kind == kCopyI16F64 || kind == kCopyF32F64)));
    switch (kind) {
    case kCopyBytes:
        if (fast) stream_bytes(dst + first, src + first, count);
        else memcpy(dst + first, src + first, count);
        break;
    case kCopyI16F32: {
        const int16_t *s = reinterpret_cast<const int16_t *>(src) + first;
        float *d = reinterpret_cast<float *>(dst) + first;
        if (fast) { stream_i16_f32(d, s, count); break; }
        for (size_t i = 0; i < count; ++i) d[i] = (float)s[i];
        break;
    }
    case kCopyI16F64: {
        const int16_t *s = reinterpret_cast<const int16_t *>(src) + first;
        double *d = reinterpret_cast<double *>(dst) + first;
        if (fast) { stream_i16_f64(d, s, count); break; }
        for (size_t i = 0; i < count; ++i) d[i] = (double)s[i];
        break;
    }
    case kCopyF32F64: {
        const float *s = reinterpret_cast<const float *>(src) + first;
        double *d = reinterpret_cast<double *>(dst) + first;
        if (fast) { stream_f32_f64(d, s, count); break; }
        for (size_t i = 0; i < count; ++i) d[i] = (double)s[i];
        break;
    }
    }
}

class CopyPool {
public:
    static CopyPool &get()
    {
        // (never destroyed: its threads wait on members of it, and a process that exits must not join them)
        static CopyPool *pool = [] {
            CopyPool *p = new CopyPool;
            // a forked child has the object but none of its threads (and whatever state a helper was in): it copies alone
            pthread_atfork(nullptr, nullptr, [] { if (instance_) instance_->orphaned(); });
            instance_ = p;
            return p;
        }();
        return *pool;
    }
    // n elements (bytes for kCopyBytes)
    void copy(uint8_t *dst, const uint8_t *src, size_t n, CopyKind kind = kCopyBytes)
    {
        if (n < 2 * kPiece || workers_.empty()) { copy_piece(dst, src, 0, n, kind); return; }
        std::lock_guard<std::mutex> one_at_a_time(submit_);
        {
            // (a helper that woke up late for the copy before is still inside work(): the job's fields are its to read)
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [&] { return active_ == 0; });
            dst_ = dst; src_ = src; n_ = n; kind_ = kind;
            pieces_ = (n + kPiece - 1) / kPiece;
            next_.store(0);
            done_ = 0;
            ++generation_;
        }
        cv_.notify_all();
        const size_t mine = work();
        std::unique_lock<std::mutex> lk(m_);
        done_ += mine;
        cv_done_.wait(lk, [&] { return done_ == pieces_ && active_ == 0; });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }

private:
    static constexpr size_t kPiece = (size_t)128 << 10; // elements per piece
    static CopyPool *instance_;
    CopyPool()
    {
        unsigned cores = std::thread::hardware_concurrency();
        // one process per GPU on a node (torchrun / mpirun export the local world size): the ranks share the host's cores
        for (const char *name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS"})
            if (const char *e = getenv(name)) {
                const int ranks = atoi(e);
                if (ranks > 1) cores /= (unsigned)ranks;
                break;
            }
        unsigned n = cores >= 16 ? 7 : cores >= 8 ? 3 : cores >= 4 ? 1 : 0; // helpers beside the calling thread
        if (const char *e = getenv("WS_COPY_THREADS")) n = (unsigned)std::max(0, std::min(31, atoi(e) - 1));
        for (unsigned i = 0; i < n; ++i) {
            try { workers_.emplace_back([this] { loop(); }); }
            catch (...) { break; }
        }
    }
    void orphaned() // in the child of a fork: no helper exists here, whatever the parent's were doing
    {
        // (the std::thread objects are the parent's: dropped without a join, their destructors never run -- `new`ed state)
        new (&workers_) std::vector<std::thread>();
        new (&submit_) std::mutex();
        new (&m_) std::mutex();
        new (&cv_) std::condition_variable();
        new (&cv_done_) std::condition_variable();
        active_ = 0;
        done_ = pieces_ = 0;
        generation_ = 0;
    }
    size_t work()
    {
        size_t count = 0;
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= pieces_) break;
            const size_t off = i * kPiece;
            copy_piece(dst_, src_, off, std::min(kPiece, n_ - off), kind_);
            ++count;
        }
        return count;
    }
    void loop()
    {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                ++active_;
            }
            const size_t count = work();
            std::lock_guard<std::mutex> lk(m_);
            done_ += count;
            --active_;
            cv_done_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex submit_, m_;
    std::condition_variable cv_, cv_done_;
    uint8_t *dst_ = nullptr;
    const uint8_t *src_ = nullptr;
    size_t n_ = 0, pieces_ = 0, done_ = 0;
    CopyKind kind_ = kCopyBytes;
    int active_ = 0; // helpers inside work()
    std::atomic<size_t> next_{0};
    unsigned long long generation_ = 0;
    bool stop_ = false;
};
CopyPool *CopyPool::instance_ = nullptr;

void stage_copy(uint8_t *dst, const uint8_t *src, size_t n) { CopyPool::get().copy(dst, src, n); }

hipError_t stage_for(HostSpan &sp)
{
    if (!sp.stage) return hipErrorInvalidValue;
    return host_ensure(*sp.stage, sp.n);
}

// bytes [off, off + bytes) of the caller's buffer -> device
hipError_t span_upload(HostSpan &sp, size_t off, void *dev, size_t bytes, hipStream_t s)
{
    if (off + bytes > sp.n) return hipErrorInvalidValue;
    if (sp.how == HostSpan::kCallerPinned) {
        const hipError_t e = hipMemcpyAsync(dev, sp.p + off, bytes, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) return e;
        (void)hipGetLastError(); // caller-pinned memory the runtime will not copy from as one range: through the stage
        sp.how = HostSpan::kStaged;
        sp.why = "the runtime refused a direct copy from caller-pinned memory";
    }
    if (sp.how != HostSpan::kStaged) return hipErrorInvalidValue;
    // just the bytes asked for, at their own offset in the stage (which is as long as the buffer): a call that uploads
    // its images band by band copies the next band into the stage while the last one is on the bus
    const hipError_t e = stage_for(sp);
    if (e != hipSuccess) return e;
    stage_copy(sp.stage->p + off, sp.p + off, bytes);
    sp.loaded = true;
    return hipMemcpyAsync(dev, sp.stage->p + off, bytes, hipMemcpyHostToDevice, s);
}

// rows of `row_bytes`, `pitch` bytes apart in the caller's buffer from byte `off` on -> dense rows on the device
hipError_t span_upload_rows(HostSpan &sp, size_t off, size_t pitch, void *dev, size_t row_bytes, size_t rows, hipStream_t s)
{
    if (!rows || !row_bytes) return hipSuccess;
    if (off + pitch * (rows - 1) + row_bytes > sp.n) return hipErrorInvalidValue;
    if (pitch == row_bytes) return span_upload(sp, off, dev, row_bytes * rows, s);
    if (sp.how == HostSpan::kCallerPinned) {
        const hipError_t e = hipMemcpy2DAsync(dev, row_bytes, sp.p + off, pitch, row_bytes, rows, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
        sp.how = HostSpan::kStaged;
        sp.why = "the runtime refused a direct copy from caller-pinned memory";
    }
    if (sp.how != HostSpan::kStaged) return hipErrorInvalidValue;
    const hipError_t e = stage_for(sp);
    if (e != hipSuccess) return e;
    for (size_t r = 0; r < rows; ++r) memcpy(sp.stage->p + r * row_bytes, sp.p + off + r * pitch, row_bytes); // dense in the stage
    return hipMemcpyAsync(dev, sp.stage->p, row_bytes * rows, hipMemcpyHostToDevice, s);
}

size_t wire_bytes(int wire, int esz) { return wire == kWireI16 ? 2 : wire == kWireF32 ? 4 : (size_t)esz; }

// `rows` dense rows of `row_elems` elements on the device, in wire format -> the caller's buffer from ELEMENT `off` on,
// rows `pitch` elements apart, elements of esz bytes.  A wire format other than the caller's own always goes through
// the stage (the widening is the stage -> caller copy), whatever kind of memory the caller's buffer is.
hipError_t span_download(HostSpan &sp, size_t off, size_t pitch, const void *dev, size_t row_elems, size_t rows, int wire, int esz, hipStream_t s)
{
    if (!rows || !row_elems) return hipSuccess;
    if ((off + pitch * (rows - 1) + row_elems) * (size_t)esz > sp.n) return hipErrorInvalidValue;
    const size_t wb = wire_bytes(wire, esz);
    const bool same = wb == (size_t)esz; // (float32 on the wire for a float32 map)
    if (sp.how == HostSpan::kCallerPinned && same) {
        const hipError_t e = copy_rows(sp.p + off * esz, pitch * esz, dev, row_elems * esz, row_elems * esz, rows, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
        sp.how = HostSpan::kStaged;
        sp.why = "the runtime refused a direct copy to caller-pinned memory";
    }
    if (sp.how != HostSpan::kStaged && sp.how != HostSpan::kCallerPinned) return hipErrorInvalidValue;
    const hipError_t e = stage_for(sp);
    if (e != hipSuccess) return e;
    // dense in the stage, at the wire-format offset of its first element (the stage is as long as the buffer)
    sp.down.push_back({off * wb, off * (size_t)esz, row_elems, rows, pitch * (size_t)esz, same ? (int)kWireSame : wire, esz});
    return hipMemcpyAsync(sp.stage->p + off * wb, dev, row_elems * wb * rows, hipMemcpyDeviceToHost, s);
}

// the same for plain bytes (the consumers' buffers): offsets, pitch and row length in bytes
hipError_t span_download_bytes(HostSpan &sp, size_t off, size_t pitch, const void *dev, size_t row_bytes, size_t rows, hipStream_t s)
{
    return span_download(sp, off, pitch, dev, row_bytes, rows, kWireSame, 1, s);
}

// Hand a staged download to the caller (the copy of this segment into the stage is through).
void span_scatter_seg(HostSpan &sp, HostSpan::Seg &g)
{
    if (!g.rows) return; // handed over already
    const CopyKind kind = g.wire == kWireI16 ? (g.esz == 8 ? kCopyI16F64 : kCopyI16F32) : g.wire == kWireF32 && g.esz == 8 ? kCopyF32F64 : kCopyBytes;
    const size_t unit = kind == kCopyBytes ? (size_t)g.esz : 1; // kCopyBytes counts bytes, the widening kinds elements
    const size_t wb = wire_bytes(g.wire, g.esz);
    if (g.host_pitch == g.row_elems * (size_t)g.esz || g.rows == 1) { // dense: one copy
        CopyPool::get().copy(sp.p + g.host_off, sp.stage->p + g.stage_off, g.row_elems * g.rows * unit, kind);
    } else {
        for (size_t r = 0; r < g.rows; ++r)
            copy_piece(sp.p + g.host_off + r * g.host_pitch, sp.stage->p + g.stage_off + r * g.row_elems * wb, 0, g.row_elems * unit, kind);
    }
    g.rows = 0;
}

void span_scatter(HostSpan &sp) // (the copies into the stage are through: the caller of this has synchronised)
{
    for (HostSpan::Seg &g : sp.down) span_scatter_seg(sp, g);
    sp.down.clear();
}

// Hand staged downloads to the caller.  ONLY after every stream that carried a copy of these spans is idle.
void spans_finish(HostSpan *sp, int count)
{
    for (int i = 0; i < count; ++i) {
        span_scatter(sp[i]);
        sp[i].how = HostSpan::kUnused;
    }
}

} // namespace

struct ws_context {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr;
    hipEvent_t ev_scratch = nullptr;      // end of the last search: the scratch planes are free again
    hipStream_t scratch_stream = nullptr; // ... the stream it ran on
    bool scratch_busy = false;
    bool profiling = false, kernel_timed = false;
    DevBuf plane_a, plane_b, keys, cost, bs_plane, max_block, sel, sel_planes, top3, d_left, d_right, d_out, d_out64 /* the consumers' scratch */, d_out16;
    Job jobs[2];             // ws_enqueue_host alternates between two slots
    int job_next = 0;
    hipStream_t copy_stream = nullptr; // host <-> device copies of the batched path, beside the searches
    hipStream_t down_stream = nullptr; // ws_search_host in bands: maps go down here while images still come up on copy_stream
    static constexpr int kMaxBands = 8;
    hipEvent_t ev_band_up[kMaxBands] = {}, ev_band_done[kMaxBands] = {}, ev_band_down[kMaxBands] = {};
    unsigned int *status_host = nullptr, *status_dev = nullptr; // mapped pinned words the kernels flag trouble in (word 0: ws_smooth_left_bands_kernel gave up; word 1: the integer box filter met a value it cannot carry)
    DevBuf d_flag;                     // 256 bytes: word 0 = the integer box filter met a value it cannot carry
    bool plan_valid = false, plan_ok = false; // run_search: the last problem's plan
    Canon plan_canon{};
    int plan_tune[3] = {0, 0, 0};
    MarchLaunch plan_launch{};
    int last_outliers_path = 0;        // ws_last_outliers_path
    int last_how[3] = {0, 0, 0};       // ws_last_host_paths: how the last host call's left / right / out bytes crossed
    int last_wire = 0;                 // ... and the wire format of its map (ws_last_wire_format)
    std::vector<HostSpan> batch_spans; // caller buffers of the pairs enqueued since the last ws_wait (released there)
    HostBuf h_left, h_right, h_out;    // ws_search_host: gathered rows of cut-out images (gather_rows), stages (HostSpan)
    HostBuf h_aux[2];                  // stages of the consumers' further buffers
    int host_bands = -1;               // ws_set_host_bands: 0 = never split, -1 = automatic
    std::string err;
    std::string last_kernel;
    int last_threads = 0, last_wgs = 0, last_lds = 0;
    bool var_block_ran = false;
    // what the last run_search left behind, for the passes that follow it (smoothFactor)
    bool last_march = false;
    Canon last_canon{};
    Plane last_pa{}, last_pb{};
    int last_skip[4] = {0, 0, 0, 0};
    bool want_cost = false;       // run_search: also leave the winners' costs (right view, smoothFactor)
    bool want_planes = false;     // run_search: also pack the dword planes (the left view's smoothFactor pass reads them)
    bool last_planes = false;     // ... and whether the last search did
    int32_t *last_cost = nullptr; // where it left them (pitch = plane width), or null
    int16_t *direct_i16 = nullptr; // run_search: the kernels store the map as 16-bit integers here (the wire format of a host call)
    int tune_nxr = 0, tune_rows = 0, tune_threads = 0;
};

namespace {

int fail(ws_context *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define WS_HIP(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, WS_ERR_HIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_),        \
                        __FILE__, __LINE__);                                                    \
    } while (0)

int ensure(ws_context *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return WS_OK;
    if (b.p) WS_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    WS_HIP(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return WS_OK;
}

// A narrow cut-out of a much wider image goes through a pinned buffer of the library's own: the rows are gathered
// on the host and cross as one dense linear copy.  (The runtime's 2-D copy from pageable memory takes a per-row
// path, ~15 us a row; and no copy of this library reads or writes pageable memory through the runtime any more,
// see DESIGN.md 5.)
hipError_t gather_rows(HostBuf &b, const ws_image *im)
{
    const size_t rb = (size_t)im->width * 3;
    hipError_t e = host_ensure(b, rb * im->height);
    if (e != hipSuccess) return e;
    for (int y = 0; y < im->height; ++y) memcpy(b.p + (size_t)y * rb, im->data + (size_t)y * im->stride, rb);
    return hipSuccess;
}

// Is an image worth copying as one linear span, row padding included?  Yes unless it is a narrow
// cut out of a much wider image AND big (the per-row path costs ~15 us per row).
bool linear_span(const ws_image *im)
{
    const size_t dense = (size_t)im->width * 3 * im->height;
    const size_t span = (size_t)im->stride * (im->height - 1) + (size_t)im->width * 3;
    return span <= 2 * dense || span <= ((size_t)32 << 20);
}

bool image_ok(const ws_image *im)
{
    return im && im->data && im->width > 0 && im->height > 0 && im->stride >= 3 * im->width;
}

int check_params(ws_context *ctx, const ws_params *p, const ws_image *L, const ws_image *R)
{
    if (!p || !image_ok(L) || !image_ok(R)) return fail(ctx, WS_ERR_ARG, "null or malformed image / params");
    if (p->view != WS_VIEW_LEFT && p->view != WS_VIEW_RIGHT && p->view != WS_VIEW_LINEAR)
        return fail(ctx, WS_ERR_ARG, "unknown view %d", p->view);
    if (p->cost != WS_COST_SSD && p->cost != WS_COST_SAD) return fail(ctx, WS_ERR_ARG, "unknown cost %d", p->cost);
    if (p->view != WS_VIEW_LINEAR && (p->block_size < 1 || p->block_size > 63))
        return fail(ctx, WS_ERR_ARG, "blockSize %d outside [1,63]", p->block_size);
    if (p->view == WS_VIEW_LINEAR && p->linear_range < 1) return fail(ctx, WS_ERR_ARG, "linear_range < 1");
    if (!(p->smooth_factor == p->smooth_factor)) return fail(ctx, WS_ERR_ARG, "smoothFactor is NaN");
    if (p->var_block && p->view == WS_VIEW_RIGHT && p->subpixel)
        return fail(ctx, WS_ERR_UNSUPPORTED, "sub-pixel refinement together with varBlock");
    if (p->var_block && p->view == WS_VIEW_RIGHT && !(p->thres == p->thres))
        return fail(ctx, WS_ERR_ARG, "thres is NaN");
    if (p->subpixel && p->view == WS_VIEW_LINEAR) return fail(ctx, WS_ERR_UNSUPPORTED, "sub-pixel on LinearSearch");
    if (p->subpixel && p->smooth_factor != 1.0) return fail(ctx, WS_ERR_UNSUPPORTED, "sub-pixel refinement together with smoothFactor != 1");
    const int h1 = L->height, w1 = L->width, h2 = R->height, w2 = R->width;
    const int height = std::min(h1, h2);
    const int half = (p->block_size - 1) / 2;
    if (p->view == WS_VIEW_LEFT) {
        // Rect(x-half, y-half, bs, bs) leaves the image for even bs (BlockSearch.cpp:46-49)
        if ((p->block_size & 1) == 0 && height - 2 * half > 0 && w1 - 2 * half > 0)
            return fail(ctx, WS_ERR_GEOMETRY, "even blockSize %d: the reference throws cv::Exception", p->block_size);
    } else if (p->view == WS_VIEW_RIGHT && p->max_disparity > p->min_disparity) {
        if (p->min_disparity < 0)
            return fail(ctx, WS_ERR_GEOMETRY, "minDisparity < 0: left ROI starts before column 0 (BlockSearch.cpp:151)");
        // leftImage_(Rect(.., y-up, .., up+down)) needs y + down <= h1 (BlockSearch.cpp:151-154)
        for (int y = std::max(0, height - half - 1); y < height; ++y) { // (only the last rows can overrun)
            const int down = std::min(h2 - y - 1, half);
            if (y + down > h1)
                return fail(ctx, WS_ERR_GEOMETRY, "left image too short for the right view window at row %d", y);
        }
        // varBlock grows windows by data: with a right image taller than the left one a grown window near
        // row h1 needs left-image rows >= h1 and the reference throws (BlockSearch.cpp:151-154) -- but only
        // if such a pixel happens to grow.  Defined here: rejected up front, whatever the data.
        if (p->var_block && h2 > h1)
            return fail(ctx, WS_ERR_GEOMETRY, "varBlock with a right image taller than the left one: a grown window "
                                              "would leave the left image (BlockSearch.cpp:151-154)");
    }
    (void)w2;
    return WS_OK;
}

// Reduce LEFT / RIGHT to the canonical search.  Returns false when no marching region exists.
bool make_canon(const ws_params *p, const ws_image *L, const ws_image *R, Canon *c)
{
    const int h1 = L->height, w1 = L->width, h2 = R->height, w2 = R->width;
    const int height = std::min(h1, h2);
    const int half = (p->block_size - 1) / 2;
    Canon k{};
    k.ssd = p->cost == WS_COST_SSD;
    if (p->view == WS_VIEW_LEFT) {
        k.wa = w1; k.ha = h1; k.wb = w2; k.hb = h2;
        k.ww = k.wh = p->block_size;
        k.wx0 = k.wy0 = -half;
        k.boff = 0;
        // no candidate beyond what the geometry allows (x - d >= half with x <= w1 - 1 - half): a range far
        // wider than the image costs neither d-group passes nor tie-tag bits; the tags keep their order
        k.d_lo = 1; k.d_hi = k.d_hi_clipped = std::min(p->max_disparity, w1 - 1 - 2 * half);
        k.b_lo = half; k.b_hi = w2 - half - 1;
        k.ox0 = half; k.ox1 = w1 - half;
        k.oy0 = half; k.oy1 = height - half;
        k.prefer_large = 1; k.mirror = 0; k.fallback_neg = 0;
    } else if (p->view == WS_VIEW_RIGHT) {
        if (half < 1) return false;
        k.wa = w2; k.ha = h2; k.wb = w1; k.hb = h1;
        k.ww = k.wh = 2 * half;
        k.wx0 = 1 - half; k.wy0 = -half;
        k.boff = w1 - w2;
        // (x + d + half < w1 with x >= half: the same clamp)
        k.d_lo = p->min_disparity; k.d_hi = std::min(p->max_disparity - 1, w1 - 1 - 2 * half);
        k.d_hi_clipped = std::min(p->max_disparity - 1, w1 - 1); // border ring: x >= 0 and right >= 0 only
        k.b_lo = half; k.b_hi = w1 - 1 - half;
        k.ox0 = half; k.ox1 = w2 - half;
        k.oy0 = half; k.oy1 = std::min(h2 - half, height);
        k.prefer_large = 0; k.mirror = 1; k.fallback_neg = 1;
    } else {
        return false;
    }
    *c = k;
    return k.ox1 > k.ox0 && k.oy1 > k.oy0 && k.d_hi >= k.d_lo;
}

int run_search(ws_context *ctx, const ws_params *p, const ws_image *L, const ws_image *R,
               float *out, int out_stride, hipStream_t s);

int run_device_on(ws_context *ctx, const ws_params *p, const ws_image *L, const ws_image *R,
                  float *out, int out_stride, hipStream_t s);

// The context's scratch planes are shared by every call: a call on another stream than the previous
// one first waits (on the device) for that previous call to be done with them.
int run_device(ws_context *ctx, const ws_params *p, const ws_image *L, const ws_image *R,
               float *out, int out_stride, hipStream_t s)
{
    if (ctx->scratch_busy && s != ctx->scratch_stream) WS_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_scratch, 0));
    const int rc = run_device_on(ctx, p, L, R, out, out_stride, s);
    ctx->scratch_busy = false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap == hipStreamCaptureStatusNone) { // (an event recorded inside a capture cannot be waited for outside it)
        WS_HIP(ctx, hipEventRecord(ctx->ev_scratch, s));
        ctx->scratch_busy = true;
        ctx->scratch_stream = s;
    }
    return rc;
}

// smoothFactor: for the right view and LinearSearch the factor can only reach d = 0 beside a
// zero-valued neighbour (see ws_smooth.hip), and only when d = 0 is a candidate at all.
int run_device_on(ws_context *ctx, const ws_params *p, const ws_image *L, const ws_image *R,
                  float *out, int out_stride, hipStream_t s)
{
    ws_params q = *p;
    if (q.view == WS_VIEW_LINEAR) q.min_disparity = 0;
    if (q.view == WS_VIEW_LEFT && q.smooth_factor != 1.0) {
        // the data-parallel search (smoothFactor 1) gives d1; the raster-order pass does the rest
        q.smooth_factor = 1.0;
        ctx->want_planes = true;
        int rc = run_search(ctx, &q, L, R, out, out_stride, s);
        ctx->want_planes = false;
        if (rc != WS_OK) return rc;
        GenericArgs ga{};
        ga.L = L->data; ga.R = R->data;
        ga.w1 = L->width; ga.h1 = L->height; ga.s1 = L->stride;
        ga.w2 = R->width; ga.h2 = R->height; ga.s2 = R->stride;
        ga.view = p->view; ga.ssd = p->cost == WS_COST_SSD;
        ga.block_size = p->block_size; ga.min_d = 0; ga.max_d = p->max_disparity;
        ga.out = out; ga.out_pitch = out_stride;
        // per pixel the best candidate's cost (0 <= s <= 1) or the three best candidates
        if ((rc = ensure(ctx, ctx->top3, smooth_left_top_bytes(L->width, L->height, p->smooth_factor))) != WS_OK) return rc;
        uint32_t *top3 = static_cast<uint32_t *>(ctx->top3.p);
        WS_HIP(ctx, launch_smooth_left(ga, p->smooth_factor, top3, ctx->last_march && ctx->last_planes ? &ctx->last_canon : nullptr,
                                       ctx->last_pa, ctx->last_pb, ctx->status_dev, s));
        return WS_OK;
    }
    const bool smooth = q.smooth_factor != 1.0 && q.view != WS_VIEW_LEFT && q.min_disparity == 0;
    if (!smooth) return run_search(ctx, &q, L, R, out, out_stride, s);
    q.min_disparity = 1; // the data-parallel part: best candidate among d >= 1
    q.subpixel = 0;
    ctx->want_cost = p->view == WS_VIEW_RIGHT && !p->var_block;
    int rc = run_search(ctx, &q, L, R, out, out_stride, s);
    ctx->want_cost = false;
    if (rc != WS_OK) return rc;
    const int sel_pitch = (R->width + 63) & ~63;
    if ((rc = ensure(ctx, ctx->sel, (size_t)sel_pitch * (smooth_sel_rows(R->height) + 64))) != WS_OK) return rc;
    GenericArgs ga{};
    ga.L = L->data; ga.R = R->data;
    ga.w1 = L->width; ga.h1 = L->height; ga.s1 = L->stride;
    ga.w2 = R->width; ga.h2 = R->height; ga.s2 = R->stride;
    ga.view = p->view; ga.ssd = p->cost == WS_COST_SSD;
    ga.block_size = p->block_size; ga.min_d = 0; ga.max_d = p->max_disparity;
    ga.linear_range = p->linear_range;
    ga.out = out; ga.out_pitch = out_stride;
    if (p->view == WS_VIEW_RIGHT && p->var_block) { // the windows ws_varblock_kernel chose
        ga.bs_plane = static_cast<const int16_t *>(ctx->bs_plane.p);
        ga.bs_pitch = (R->width + 63) & ~63;
    }
    if ((rc = ensure(ctx, ctx->sel_planes, smooth_planes_bytes(R->width, R->height))) != WS_OK) return rc;
    const bool on_planes = p->view == WS_VIEW_RIGHT && !p->var_block && ctx->last_march && ctx->last_planes && ctx->last_cost;
    if (on_planes) {
        ga.skip_x0 = ctx->last_skip[0]; ga.skip_x1 = ctx->last_skip[1];
        ga.skip_y0 = ctx->last_skip[2]; ga.skip_y1 = ctx->last_skip[3];
    }
    WS_HIP(ctx, launch_smooth(ga, p->smooth_factor, static_cast<uint8_t *>(ctx->sel.p), sel_pitch,
                              static_cast<unsigned long long *>(ctx->sel_planes.p),
                              on_planes ? &ctx->last_canon : nullptr, ctx->last_pa, ctx->last_pb,
                              on_planes ? ctx->last_cost : nullptr, on_planes ? ctx->last_canon.wa : 0, s));
    if (p->subpixel) return fail(ctx, WS_ERR_UNSUPPORTED, "sub-pixel refinement together with smoothFactor != 1");
    return WS_OK;
}

int run_search(ws_context *ctx, const ws_params *p, const ws_image *L, const ws_image *R,
               float *out, int out_stride, hipStream_t s)
{
    const int ow = p->view == WS_VIEW_LEFT ? L->width : R->width;
    if (out_stride < ow) return fail(ctx, WS_ERR_ARG, "out_stride %d < width %d", out_stride, ow);

    GenericArgs ga{};
    ga.L = L->data; ga.R = R->data;
    ga.w1 = L->width; ga.h1 = L->height; ga.s1 = L->stride;
    ga.w2 = R->width; ga.h2 = R->height; ga.s2 = R->stride;
    ga.view = p->view; ga.ssd = p->cost == WS_COST_SSD;
    ga.block_size = p->block_size; ga.min_d = p->min_disparity; ga.max_d = p->max_disparity;
    ga.linear_range = p->linear_range;
    ga.out = out; ga.out_pitch = out_stride;
    ga.out16 = ctx->direct_i16;

    ctx->last_cost = nullptr;
    if (p->view == WS_VIEW_RIGHT) ctx->var_block_ran = false;
    if (p->view == WS_VIEW_RIGHT && p->var_block) {
        // the ordinary search first; then one wave per pixel decides the window (ws_varblock_kernel)
        // and searches again only where it grew
        ws_params q = *p;
        q.var_block = 0;
        q.subpixel = 0;
        int rc = run_search(ctx, &q, L, R, out, out_stride, s);
        if (rc != WS_OK) return rc;
        const int bs_pitch = (R->width + 63) & ~63;
        if ((rc = ensure(ctx, ctx->bs_plane, (size_t)bs_pitch * R->height * 2)) != WS_OK) return rc;
        if ((rc = ensure(ctx, ctx->max_block, 64)) != WS_OK) return rc;
        WS_HIP(ctx, launch_varblock(ga, p->thres, static_cast<int16_t *>(ctx->bs_plane.p), bs_pitch,
                                    static_cast<int *>(ctx->max_block.p), s));
        ctx->last_kernel = "ws_varblock_kernel";
        ctx->last_threads = 256;
        ctx->last_wgs = (int)(((long long)R->width * R->height + 3) / 4);
        ctx->last_lds = 0;
        ctx->var_block_ran = true;
        return WS_OK;
    }
    Canon c{};
    MarchLaunch m{};
    Plane ring_a{}, ring_b{};
    // (the plan of the last problem is kept: a queue of equal pairs asks for the same one every call, and the planner
    // walks every strip count for up to three candidate tilings and two workgroup sizes -- 5 us of a 15 us enqueue)
    bool march = make_canon(p, L, R, &c);
    if (march) {
        const int tune[3] = {ctx->tune_nxr, ctx->tune_rows, ctx->tune_threads};
        if (ctx->plan_valid && !memcmp(&ctx->plan_canon, &c, sizeof c) && !memcmp(ctx->plan_tune, tune, sizeof tune)) {
            m = ctx->plan_launch;
            march = ctx->plan_ok;
        } else {
            march = march_plan(c, ctx->num_cus, tune[0], tune[1], tune[2], &m);
            ctx->plan_canon = c;
            memcpy(ctx->plan_tune, tune, sizeof tune);
            ctx->plan_launch = m;
            ctx->plan_ok = march;
            ctx->plan_valid = true;
        }
    }
    // Dword planes of both images: only for the kernels BESIDE the marching kernel that still read them -- the right
    // view's border ring, the sub-pixel refine, the smoothFactor passes.  The marching kernel reads the caller's bytes.
    const bool planes = march && (p->view == WS_VIEW_RIGHT || p->subpixel || ctx->want_planes);
    if (march) {
        Plane pa{}, pb{};
        int rc;
        const ws_image *ia = p->view == WS_VIEW_LEFT ? L : R;
        const ws_image *ib = p->view == WS_VIEW_LEFT ? R : L;
        if (c.mirror) {
            ga.skip_x0 = c.wa - c.ox1; ga.skip_x1 = c.wa - c.ox0;
        } else {
            ga.skip_x0 = c.ox0; ga.skip_x1 = c.ox1;
        }
        ga.skip_y0 = c.oy0; ga.skip_y1 = c.oy1;
        if (planes) {
            march_plane_geometry(c, m, &pa, &pb);
            if ((rc = ensure(ctx, ctx->plane_a, (size_t)pa.pitch * c.ha * 4)) != WS_OK) return rc;
            if ((rc = ensure(ctx, ctx->plane_b, (size_t)pb.pitch * c.hb * 4)) != WS_OK) return rc;
            pa.data = static_cast<uint32_t *>(ctx->plane_a.p);
            pb.data = static_cast<uint32_t *>(ctx->plane_b.p);
            ring_a = pa;
            ring_b = pb;
            WS_HIP(ctx, launch_pack(c, ia->data, ia->stride, pa, ib->data, ib->stride, pb, s));
        }
        if (ctx->profiling) WS_HIP(ctx, hipEventRecord(ctx->evk0, s));
        const int keys_pitch = (c.wa + 15) & ~15;
        if (m.passes > 1 && (rc = ensure(ctx, ctx->keys, (size_t)keys_pitch * c.ha * 8)) != WS_OK) return rc;
        int32_t *cost_out = nullptr; // the smoothFactor passes of the right view want the winners' costs
        if (ctx->want_cost && march_has_cost(c)) {
            if ((rc = ensure(ctx, ctx->cost, (size_t)c.wa * c.ha * 4)) != WS_OK) return rc;
            cost_out = static_cast<int32_t *>(ctx->cost.p);
        }
        ctx->last_cost = cost_out;
        // left view: the marching kernel also writes the zeros outside its interior (BlockSearch.cpp:33,36,38); the
        // right view's ring runs on the packed planes after it
        WS_HIP(ctx, launch_march(c, m, ia->data, ia->stride, ib->data, ib->stride, out, ctx->direct_i16, out_stride,
                                 p->view == WS_VIEW_LEFT, L->width, L->height, ctx->keys.p, keys_pitch, cost_out, c.wa, s));
        if (ctx->profiling) {
            WS_HIP(ctx, hipEventRecord(ctx->evk1, s));
            ctx->kernel_timed = true;
        }
        ctx->last_kernel = march_kernel_name(c, m);
        ctx->last_threads = m.threads;
        ctx->last_wgs = m.tiles * m.strips;
        ctx->last_lds = (int)m.lds_bytes;
    } else {
        ctx->last_kernel = p->view == WS_VIEW_LINEAR ? "ws_linear_kernel" : "ws_generic_kernel";
        ctx->last_threads = 256;
        ctx->last_wgs = ((ow + 255) / 256) * (p->view == WS_VIEW_LEFT ? L->height : R->height);
        ctx->last_lds = 0;
    }
    // everything the marching kernel does not own: border ring, rows past min(h1,h2), or all of it
    if (march && p->view == WS_VIEW_RIGHT)
        WS_HIP(ctx, launch_ring(c, ring_a, ring_b, ga, out, out_stride, ctx->last_cost, c.wa, s));
    else if (!march && p->view == WS_VIEW_LINEAR)
        WS_HIP(ctx, launch_linear(ga, s));
    else if (!march)
        WS_HIP(ctx, launch_generic(ga, s));
    ctx->last_march = march;
    ctx->last_planes = planes;
    if (march) {
        ctx->last_canon = c; ctx->last_pa = ring_a; ctx->last_pb = ring_b;
        ctx->last_skip[0] = ga.skip_x0; ctx->last_skip[1] = ga.skip_x1;
        ctx->last_skip[2] = ga.skip_y0; ctx->last_skip[3] = ga.skip_y1;
    }
    if (p->subpixel) {
        if (march) WS_HIP(ctx, launch_refine_planes(c, m, ring_a, ring_b, out, out_stride, s));
        WS_HIP(ctx, launch_refine(ga, s)); // the pixels outside the marching interior (all of them without it)
    }
    return WS_OK;
}

// A search whose kernels only ever WRITE the map (smoothFactor 1, no sub-pixel refine, no varBlock: the marching
// kernel's flush, the border ring, LinearSearch, the brute force) can store it in the wire format itself.
bool writes_only(const ws_params *p) { return p->smooth_factor == 1.0 && !p->subpixel && !(p->var_block && p->view == WS_VIEW_RIGHT); }

// The wire format of a host call's map (see "WIRE FORMAT" above): 16-bit integers when the search kernels can store
// them themselves and every value fits.  Whatever the disparity range, a stored value is a difference of two columns
// of one image row or a +-x fallback (BlockSearch.cpp:82: x - cx with 0 <= cx < x; :174: cx - x with x <= cx < w1, or
// -x; LinearSearch.cpp:53: col - j), so |value| < max(w1, w2): images up to 32767 pixels wide fit.  Else float32.
int wire_for(const ws_params *p, const ws_image *L, const ws_image *R)
{
    const bool fits = L->width <= 32767 && R->width <= 32767;
    return writes_only(p) && fits ? kWireI16 : kWireF32;
}

// one search whose map ends up in wire format: in out16 (kWireI16: stored by the search kernels, scratch32 stays unused)
// or in scratch32 (kWireF32)
int run_device_wire(ws_context *ctx, const ws_params *p, const ws_image *L, const ws_image *R, float *scratch32, int16_t *out16,
                    int wire, int ow, hipStream_t s)
{
    ctx->direct_i16 = wire == kWireI16 ? out16 : nullptr;
    const int rc = run_device(ctx, p, L, R, scratch32, ow, s);
    ctx->direct_i16 = nullptr;
    return rc;
}

int out_dims(const ws_params *p, const ws_image *L, const ws_image *R, int *w, int *h)
{
    *w = p->view == WS_VIEW_LEFT ? L->width : R->width;
    *h = p->view == WS_VIEW_LEFT ? L->height : R->height;
    return 0;
}

// What the kernels flagged since the last check (the streams that carried them are idle: the caller synchronised).
int check_device_status(ws_context *ctx)
{
    if (!ctx->status_host || !ctx->status_host[0]) return WS_OK;
    ctx->status_host[0] = 0;
    return fail(ctx, WS_ERR_HIP, "the left view's smoothFactor raster pass gave up waiting for the band above it "
                                 "(ws_smooth_left_bands_kernel): the map is not valid");
}

} // namespace

extern "C" {

int ws_version(void) { return WS_VERSION; }

void ws_params_default(ws_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->view = WS_VIEW_LEFT;
    p->cost = WS_COST_SSD;
    p->block_size = 7;
    p->min_disparity = 0;
    p->max_disparity = 64;
    p->smooth_factor = 1.0;
    p->var_block = 0;
    p->thres = 19.0;
    p->subpixel = 0;
    p->linear_range = 200;
}

int ws_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ws_create(int device, ws_context **out)
{
    if (!out) return WS_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, WS_ERR_HIP, "no HIP device available (%s): this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, WS_ERR_ARG, "device %d out of range [0,%d)", device, n);
    ws_context *ctx = new (std::nothrow) ws_context();
    if (!ctx) return WS_ERR_NOMEM;
    ctx->device = device;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
        (e = hipEventCreate(&ctx->evk0)) != hipSuccess || (e = hipEventCreate(&ctx->evk1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_scratch, hipEventDisableTiming)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->jobs[0].ev_h2d, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->jobs[0].ev_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->jobs[1].ev_h2d, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->jobs[1].ev_done, hipEventDisableTiming)) != hipSuccess) {
        fail(nullptr, WS_ERR_HIP, "ws_create: %s", hipGetErrorString(e));
        delete ctx;
        return WS_ERR_HIP;
    }
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    march_set_num_cus(ctx->num_cus);
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&ctx->status_host), 64, hipHostMallocMapped)) != hipSuccess ||
        (e = hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->status_dev), ctx->status_host, 0)) != hipSuccess) {
        fail(nullptr, WS_ERR_HIP, "ws_create: %s", hipGetErrorString(e));
        ws_destroy(ctx);
        return WS_ERR_HIP;
    }
    memset(ctx->status_host, 0, 64);
    *out = ctx;
    return WS_OK;
}

void ws_destroy(ws_context *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    // a batch that was never waited for: its maps are NOT handed over -- only ws_wait delivers, and a caller who
    // abandoned the batch may have freed the buffers the staged maps would be written to
    for (HostSpan &b : ctx->batch_spans) b.down.clear();
    ctx->batch_spans.clear();
    for (DevBuf *b : {&ctx->plane_a, &ctx->plane_b, &ctx->keys, &ctx->cost, &ctx->bs_plane, &ctx->max_block, &ctx->sel, &ctx->sel_planes, &ctx->top3, &ctx->d_left, &ctx->d_right, &ctx->d_out, &ctx->d_out64, &ctx->d_out16, &ctx->d_flag})
        if (b->p) (void)hipFree(b->p);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (HostBuf *b : {&ctx->h_left, &ctx->h_right, &ctx->h_out, &ctx->h_aux[0], &ctx->h_aux[1], &ctx->jobs[0].h_left, &ctx->jobs[0].h_right,
                       &ctx->jobs[0].h_out, &ctx->jobs[1].h_left, &ctx->jobs[1].h_right, &ctx->jobs[1].h_out})
        if (b->p) (void)hipHostFree(b->p);
    for (Job &j : ctx->jobs) {
        if (j.d_in) (void)hipFree(j.d_in);
        if (j.d_out) (void)hipFree(j.d_out);
        if (j.d_out16) (void)hipFree(j.d_out16);
        if (j.ev_h2d) (void)hipEventDestroy(j.ev_h2d);
        if (j.ev_done) (void)hipEventDestroy(j.ev_done);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->down_stream) {
        (void)hipStreamSynchronize(ctx->down_stream);
        (void)hipStreamDestroy(ctx->down_stream);
    }
    for (int i = 0; i < ws_context::kMaxBands; ++i) {
        if (ctx->ev_band_up[i]) (void)hipEventDestroy(ctx->ev_band_up[i]);
        if (ctx->ev_band_done[i]) (void)hipEventDestroy(ctx->ev_band_done[i]);
        if (ctx->ev_band_down[i]) (void)hipEventDestroy(ctx->ev_band_down[i]);
    }
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->evk0) (void)hipEventDestroy(ctx->evk0);
    if (ctx->evk1) (void)hipEventDestroy(ctx->evk1);
    if (ctx->ev_scratch) (void)hipEventDestroy(ctx->ev_scratch);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *ws_last_error(const ws_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int ws_validate(const ws_params *p, const ws_image *left, const ws_image *right)
{
    // same checks as the search calls, without touching a device (message via ws_last_error(NULL))
    return check_params(nullptr, p, left, right);
}

int ws_plan(const ws_params *p, const ws_image *left, const ws_image *right, int num_cus, ws_plan_info *out)
{
    if (!out) return WS_ERR_ARG;
    memset(out, 0, sizeof *out);
    int rc = check_params(nullptr, p, left, right);
    if (rc != WS_OK) return rc;
    Canon c{};
    MarchLaunch m{};
    if (make_canon(p, left, right, &c) && march_plan(c, num_cus > 0 ? num_cus : 256, 0, 0, 0, &m)) {
        out->marching = 1;
        out->x_per_thread = m.x_per_thread; out->d_per_thread = m.nd_per_thread;
        out->x_runs = m.nxr; out->d_chunks = m.nch; out->threads = m.threads;
        out->tiles = m.tiles; out->strips = m.strips; out->strip_rows = m.strip_rows;
        out->lds_bytes = (int)m.lds_bytes;
        out->interior_x0 = c.mirror ? c.wa - c.ox1 : c.ox0;
        out->interior_x1 = c.mirror ? c.wa - c.ox0 : c.ox1;
        out->interior_y0 = c.oy0; out->interior_y1 = c.oy1;
        out->passes = m.passes;
        out->tile_cols = m.tile_cols;
    }
    return WS_OK;
}

int ws_search_device(ws_context *ctx, const ws_params *p, const ws_image *left_dev,
                     const ws_image *right_dev, float *out_dev, int out_stride, void *stream)
{
    if (!ctx) return WS_ERR_ARG;
    int rc = check_params(ctx, p, left_dev, right_dev);
    if (rc != WS_OK) return rc;
    if (!out_dev) return fail(ctx, WS_ERR_ARG, "null output");
    WS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    return run_device(ctx, p, left_dev, right_dev, out_dev, out_stride, s);
}

// One boundary call in row bands.  A pixel's result depends on the image rows its window covers and on nothing
// else when smoothFactor == 1 (no raster dependency, no varBlock growth): the map's rows [y0, y1) are the interior
// rows of a search on the sub-images [y0 - half, y1 + half).  So the call is cut into K bands and three queues run
// beside each other: band k+1's image rows go up (copy_stream) while band k is searched (the context's stream) and
// band k-1's map rows come down (down_stream) -- PCIe is full duplex and the copy engines are idle during a search.
// The bytes that cross are the same as in the plain path; only their timing changes.  Results are identical
// (tests/test_gpu_parity.py::test_host_call_in_bands_equals_the_plain_call).
static int search_host_banded(ws_context *ctx, const ws_params *p, const ws_image *left, const ws_image *right,
                              void *out, int out_dtype, int ow, int oh, int nb)
{
    const int half = (p->block_size - 1) / 2;
    const size_t lb = (size_t)left->width * 3, rb = (size_t)right->width * 3;
    const int H = oh; // (equal heights: the caller checked)
    int rc;
    if (!ctx->down_stream) WS_HIP(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
    for (int i = 0; i < nb; ++i) {
        if (!ctx->ev_band_up[i]) WS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_band_up[i], hipEventDisableTiming));
        if (!ctx->ev_band_done[i]) WS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_band_done[i], hipEventDisableTiming));
        if (!ctx->ev_band_down[i]) WS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_band_down[i], hipEventDisableTiming));
    }
    const size_t span_l = (size_t)left->stride * (H - 1) + lb, span_r = (size_t)right->stride * (H - 1) + rb;
    if ((rc = ensure(ctx, ctx->d_left, span_l)) != WS_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_right, span_r)) != WS_OK) return rc;
    const int wire = wire_for(p, left, right);
    if ((rc = ensure(ctx, ctx->d_out, (size_t)ow * (H + 2 * half * nb) * 4)) != WS_OK) return rc;
    if (wire == kWireI16 && (rc = ensure(ctx, ctx->d_out16, (size_t)ow * (H + 2 * half * nb) * 2)) != WS_OK) return rc;
    const int esz = out_dtype == WS_OUT_F32 ? 4 : 8;
    uint8_t *dl = static_cast<uint8_t *>(ctx->d_left.p), *dr = static_cast<uint8_t *>(ctx->d_right.p);
    float *scratch = static_cast<float *>(ctx->d_out.p);
    // the caller's three buffers for the duration of the call (HostSpan: caller-pinned or staged)
    HostSpan sp[3];
    sp[0].p = const_cast<uint8_t *>(left->data); sp[0].n = span_l; sp[0].stage = &ctx->h_left;
    sp[1].p = const_cast<uint8_t *>(right->data); sp[1].n = span_r; sp[1].stage = &ctx->h_right;
    sp[2].p = static_cast<uint8_t *>(out); sp[2].n = (size_t)ow * H * esz; sp[2].stage = &ctx->h_out;
    HostTrace tr;
    spans_attach(sp, 3);
    tr.mark("attached");
    rc = [&]() -> int {
    int up_to = 0; // image rows [0, up_to) are on their way up
    for (int k = 0; k < nb; ++k) {
        // the first band's upload and the last band's download are what nothing can hide: those two bands are
        // half as tall as the others (nb >= 3)
        auto cut = [&](int i) -> int {
            if (nb < 3) return (int)((long long)H * i / nb);
            const long long units = 2LL * nb - 2; // 1 + 2 (nb - 2) + 1 half-bands
            const long long u = i == 0 ? 0 : i == nb ? units : 2LL * i - 1;
            return (int)(H * u / units);
        };
        const int y0 = cut(k), y1 = cut(k + 1);
        const int a = std::max(0, y0 - half), b = std::min(H, y1 + half);
        if (b > up_to) { // the rows this band adds: one linear copy per image, row padding included
            const size_t ol = (size_t)up_to * left->stride, orr = (size_t)up_to * right->stride;
            const size_t nl = (size_t)(b - 1 - up_to) * left->stride + lb, nr = (size_t)(b - 1 - up_to) * right->stride + rb;
            WS_HIP(ctx, span_upload(sp[0], ol, dl + ol, nl, ctx->copy_stream));
            WS_HIP(ctx, span_upload(sp[1], orr, dr + orr, nr, ctx->copy_stream));
            up_to = b;
        }
        tr.mark("up", k);
        WS_HIP(ctx, hipEventRecord(ctx->ev_band_up[k], ctx->copy_stream));
        WS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_band_up[k], 0));
        ws_image bl{dl + (size_t)a * left->stride, left->width, b - a, left->stride};
        ws_image br{dr + (size_t)a * right->stride, right->width, b - a, right->stride};
        float *bout = scratch + (size_t)ow * (a + 2 * half * k); // the band's own map: its border rows are scrap
        int16_t *bout16 = wire == kWireI16 ? static_cast<int16_t *>(ctx->d_out16.p) + (size_t)ow * (a + 2 * half * k) : nullptr;
        if ((rc = run_device_wire(ctx, p, &bl, &br, bout, bout16, wire, ow, ctx->stream)) != WS_OK) return rc;
        const void *src = wire == kWireI16 ? static_cast<const void *>(bout16 + (size_t)ow * (y0 - a)) : static_cast<const void *>(bout + (size_t)ow * (y0 - a));
        WS_HIP(ctx, hipEventRecord(ctx->ev_band_done[k], ctx->stream));
        WS_HIP(ctx, hipStreamWaitEvent(ctx->down_stream, ctx->ev_band_done[k], 0));
        WS_HIP(ctx, span_download(sp[2], (size_t)ow * y0, (size_t)ow, src, (size_t)ow, (size_t)(y1 - y0), wire, esz, ctx->down_stream));
        WS_HIP(ctx, hipEventRecord(ctx->ev_band_down[k], ctx->down_stream));
        tr.mark("enq", k);
    }
    // a staged map: every band's rows go from the stage to the caller's buffer -- widened on the way -- as soon as they
    // are down, while the bands behind it are still being searched (segment k of the span is band k's download).
    // (Handing a band over earlier, between two later bands' uploads, measured 3 % slower: profiles/r04/host_trace.txt.)
    if (sp[2].down.size() == (size_t)nb) {
        for (int k = 0; k < nb; ++k) {
            WS_HIP(ctx, hipEventSynchronize(ctx->ev_band_down[k]));
            tr.mark("down", k);
            span_scatter_seg(sp[2], sp[2].down[(size_t)k]);
            tr.mark("out", k);
        }
    }
    return WS_OK;
    }();
    // (also after an error: nothing may still be copying when the ranges are released)
    const hipError_t e1 = hipStreamSynchronize(ctx->copy_stream), e2 = hipStreamSynchronize(ctx->stream),
                     e3 = hipStreamSynchronize(ctx->down_stream);
    for (int i = 0; i < 3; ++i) ctx->last_how[i] = (int)sp[i].how;
    ctx->last_wire = wire;
    if (rc != WS_OK || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) sp[2].down.clear(); // (nothing half-done reaches the caller)
    spans_finish(sp, 3);
    if (rc == WS_OK && (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess))
        return fail(ctx, WS_ERR_HIP, "banded host call: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3));
    return rc == WS_OK ? check_device_status(ctx) : rc;
}

int ws_device_status(ws_context *ctx, void *stream)
{
    if (!ctx) return WS_ERR_ARG;
    WS_HIP(ctx, hipSetDevice(ctx->device));
    WS_HIP(ctx, hipStreamSynchronize(stream ? static_cast<hipStream_t>(stream) : ctx->stream));
    return check_device_status(ctx);
}

int ws_last_host_paths(const ws_context *ctx, int how[3])
{
    if (!ctx || !how) return WS_ERR_ARG;
    for (int i = 0; i < 3; ++i) how[i] = ctx->last_how[i];
    return WS_OK;
}

int ws_last_wire_format(const ws_context *ctx, int *wire)
{
    if (!ctx || !wire) return WS_ERR_ARG;
    *wire = ctx->last_wire;
    return WS_OK;
}

int ws_last_outliers_path(const ws_context *ctx, int *path)
{
    if (!ctx || !path) return WS_ERR_ARG;
    *path = ctx->last_outliers_path;
    return WS_OK;
}

int ws_set_host_bands(ws_context *ctx, int bands)
{
    if (!ctx || bands < -1 || bands > ws_context::kMaxBands) return WS_ERR_ARG;
    ctx->host_bands = bands;
    return WS_OK;
}

int ws_search_host(ws_context *ctx, const ws_params *p, const ws_image *left, const ws_image *right,
                   void *out, int out_stride, int out_dtype)
{
    if (!ctx) return WS_ERR_ARG;
    int rc = check_params(ctx, p, left, right);
    if (rc != WS_OK) return rc;
    if (!out || (out_dtype != WS_OUT_F32 && out_dtype != WS_OUT_F64)) return fail(ctx, WS_ERR_ARG, "bad output");
    int ow, oh;
    out_dims(p, left, right, &ow, &oh);
    if (out_stride < ow) return fail(ctx, WS_ERR_ARG, "out_stride %d < width %d", out_stride, ow);
    WS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    {
        // bands pay when the copies are worth hiding and each band still fills the chip
        const bool can = (p->view == WS_VIEW_LEFT || p->view == WS_VIEW_RIGHT) && p->smooth_factor == 1.0 && !p->var_block &&
                         left->height == right->height && out_stride == ow && linear_span(left) && linear_span(right) &&
                         (size_t)left->stride <= 2 * (size_t)left->width * 3 && (size_t)right->stride <= 2 * (size_t)right->width * 3;
        int nb = ctx->host_bands;
        // (measured on one MI355X, tools/host_bands_time.py: 1.5 Mpixel 0.61 -> 0.47 ms with 2..4 bands,
        // 5.9 Mpixel 2.6 -> 1.5 ms with 5..6; below a megapixel the bands' fixed costs eat the overlap)
        const size_t px = (size_t)ow * oh;
        if (nb < 0) nb = px < ((size_t)1 << 20) ? 0 : px < 3000000 ? 2 : px < 5000000 ? 4 : 6;
        if (can && nb >= 2 && oh >= 64 * nb) return search_host_banded(ctx, p, left, right, out, out_dtype, ow, oh, nb);
    }
    // One linear copy per image, row padding included (the kernels take any row stride): a 2-D copy
    // whose row length is not a multiple of 4 bytes -- 3 * width for most widths -- falls to a
    // per-row path in the runtime (measured: 15 ms instead of 0.2 ms for a 1482 x 994 image).
    // Only a big image cut out of a much wider one is copied row by row (linear_span).
    const size_t lb = (size_t)left->width * 3, rb = (size_t)right->width * 3;
    const bool lin_l = linear_span(left), lin_r = linear_span(right);
    const size_t span_l = lin_l ? (size_t)left->stride * (left->height - 1) + lb : lb * left->height;
    const size_t span_r = lin_r ? (size_t)right->stride * (right->height - 1) + rb : rb * right->height;
    if ((rc = ensure(ctx, ctx->d_left, span_l)) != WS_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_right, span_r)) != WS_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_out, (size_t)ow * oh * 4)) != WS_OK) return rc;
    const int esz = out_dtype == WS_OUT_F32 ? 4 : 8;
    const int wire = wire_for(p, left, right);
    if (wire == kWireI16 && (rc = ensure(ctx, ctx->d_out16, (size_t)ow * oh * 2)) != WS_OK) return rc;
    // The caller's buffers for the duration of the call (HostSpan): caller-pinned or staged -- every
    // host copy of this library goes the same way, whatever the band setting of the moment, and none through the
    // runtime's pageable path.
    const size_t span_o = ((size_t)out_stride * (oh - 1) + ow) * esz;
    HostSpan sp[3];
    if (lin_l) { sp[0].p = const_cast<uint8_t *>(left->data); sp[0].n = span_l; sp[0].stage = &ctx->h_left; }
    if (lin_r) { sp[1].p = const_cast<uint8_t *>(right->data); sp[1].n = span_r; sp[1].stage = &ctx->h_right; }
    sp[2].p = static_cast<uint8_t *>(out); sp[2].n = span_o; sp[2].stage = &ctx->h_out;
    spans_attach(sp, 3);
    rc = [&]() -> int {
        // (a cut-out that is not worth its whole span: gathered into pinned memory, dense rows; the call ends with
        // a synchronisation, so the two buffers are free again when the next call gathers)
        if (!lin_l) WS_HIP(ctx, gather_rows(ctx->h_left, left));
        if (!lin_r) WS_HIP(ctx, gather_rows(ctx->h_right, right));
        if (lin_l) WS_HIP(ctx, span_upload(sp[0], 0, ctx->d_left.p, span_l, s));
        else WS_HIP(ctx, hipMemcpyAsync(ctx->d_left.p, ctx->h_left.p, span_l, hipMemcpyHostToDevice, s));
        if (lin_r) WS_HIP(ctx, span_upload(sp[1], 0, ctx->d_right.p, span_r, s));
        else WS_HIP(ctx, hipMemcpyAsync(ctx->d_right.p, ctx->h_right.p, span_r, hipMemcpyHostToDevice, s));
        ws_image dl{static_cast<const uint8_t *>(ctx->d_left.p), left->width, left->height, lin_l ? left->stride : (int)lb};
        ws_image dr{static_cast<const uint8_t *>(ctx->d_right.p), right->width, right->height, lin_r ? right->stride : (int)rb};
        float *dout = static_cast<float *>(ctx->d_out.p);
        int16_t *dout16 = wire == kWireI16 ? static_cast<int16_t *>(ctx->d_out16.p) : nullptr;
        int rc2;
        if ((rc2 = run_device_wire(ctx, p, &dl, &dr, dout, dout16, wire, ow, s)) != WS_OK) return rc2;
        const void *src = wire == kWireI16 ? static_cast<const void *>(dout16) : static_cast<const void *>(dout);
        WS_HIP(ctx, span_download(sp[2], 0, (size_t)out_stride, src, (size_t)ow, (size_t)oh, wire, esz, s));
        return WS_OK;
    }();
    // (also after an error: nothing may still be copying when the ranges are released)
    const hipError_t es = hipStreamSynchronize(s);
    for (int i = 0; i < 3; ++i) ctx->last_how[i] = (int)sp[i].how;
    ctx->last_wire = wire;
    if (rc != WS_OK || es != hipSuccess) sp[2].down.clear(); // (nothing half-done reaches the caller)
    spans_finish(sp, 3);
    if (rc == WS_OK && es != hipSuccess) return fail(ctx, WS_ERR_HIP, "host call: %s", hipGetErrorString(es));
    return rc == WS_OK ? check_device_status(ctx) : rc;
}

// The batched host path keeps two pairs in flight: while one is searched (context stream) the next
// one's images go up and the previous one's map comes down on a second stream, straight from / to
// the caller's buffers as linear copies (see ws_search_host).
static int flush_job(ws_context *ctx, Job &j)
{
    if (!j.pending) return WS_OK;
    j.pending = false;
    WS_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, j.ev_done, 0));
    const int esz = j.dtype == WS_OUT_F32 ? 4 : 8;
    const void *src = j.wire == kWireI16 ? static_cast<const void *>(j.d_out16) : static_cast<const void *>(j.d_out);
    WS_HIP(ctx, span_download(ctx->batch_spans[(size_t)j.out_span], 0, (size_t)j.out_stride, src, (size_t)j.w, (size_t)j.h, j.wire, esz,
                              ctx->copy_stream));
    return WS_OK;
}

int ws_enqueue_host(ws_context *ctx, const ws_params *p, const ws_image *left, const ws_image *right,
                    void *out, int out_stride, int out_dtype)
{
    if (!ctx) return WS_ERR_ARG;
    int rc = check_params(ctx, p, left, right);
    if (rc != WS_OK) return rc;
    if (!out || (out_dtype != WS_OUT_F32 && out_dtype != WS_OUT_F64)) return fail(ctx, WS_ERR_ARG, "bad output");
    int ow, oh;
    out_dims(p, left, right, &ow, &oh);
    if (out_stride < ow) return fail(ctx, WS_ERR_ARG, "out_stride %d < width %d", out_stride, ow);
    WS_HIP(ctx, hipSetDevice(ctx->device));
    Job &job = ctx->jobs[ctx->job_next];
    Job &prev = ctx->jobs[ctx->job_next ^ 1];
    if ((rc = flush_job(ctx, job)) != WS_OK) return rc; // (only after an error left it pending)
    // this slot's previous pair: a map that came down through the slot's stage goes to its caller now, before the
    // stage is used again
    if (job.out_span >= 0 && !ctx->batch_spans[(size_t)job.out_span].down.empty()) {
        WS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        span_scatter(ctx->batch_spans[(size_t)job.out_span]);
    }
    job.out_span = -1;
    const size_t lb = (size_t)left->width * 3, rb = (size_t)right->width * 3;
    const bool lin_l = linear_span(left), lin_r = linear_span(right);
    const size_t span_l = lin_l ? (size_t)left->stride * (left->height - 1) + lb : lb * left->height;
    const size_t span_r = lin_r ? (size_t)right->stride * (right->height - 1) + rb : rb * right->height;
    const size_t off_r = (span_l + 255) & ~(size_t)255;
    const size_t in_bytes = off_r + span_r;
    const size_t out_elems = (size_t)ow * oh;
    if (in_bytes > job.in_cap) { // (hipFree waits for the device: nothing still reads the old buffer)
        if (job.d_in) WS_HIP(ctx, hipFree(job.d_in));
        job.d_in = nullptr; job.in_cap = 0;
        WS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&job.d_in), in_bytes + in_bytes / 4));
        job.in_cap = in_bytes + in_bytes / 4;
    }
    if (out_elems > job.out_cap) {
        if (job.d_out) WS_HIP(ctx, hipFree(job.d_out));
        if (job.d_out16) WS_HIP(ctx, hipFree(job.d_out16));
        job.d_out = nullptr; job.d_out16 = nullptr; job.out_cap = 0;
        const size_t cap = out_elems + out_elems / 4;
        WS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&job.d_out), cap * 4));
        WS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&job.d_out16), cap * 2));
        job.out_cap = cap;
    }
    uint8_t *d_left = job.d_in, *d_right = job.d_in + off_r;
    hipStream_t cs = ctx->copy_stream;
    // The caller's buffers for the life of the batch (HostSpan; ws_wait hands the last maps over): pageable buffers cross
    // through the job slots' pinned stages (a host copy each way), buffers the caller pinned itself cross directly.
    const int esz = out_dtype == WS_OUT_F32 ? 4 : 8;
    HostSpan sp[3];
    if (lin_l) { sp[0].p = const_cast<uint8_t *>(left->data); sp[0].n = span_l; sp[0].stage = &job.h_left; }
    if (lin_r) { sp[1].p = const_cast<uint8_t *>(right->data); sp[1].n = span_r; sp[1].stage = &job.h_right; }
    sp[2].p = static_cast<uint8_t *>(out); sp[2].n = ((size_t)out_stride * (oh - 1) + ow) * esz; sp[2].stage = &job.h_out;
    spans_attach(sp, 3);
    const size_t first = ctx->batch_spans.size();
    for (int i = 0; i < 3; ++i) ctx->batch_spans.push_back(sp[i]);
    HostSpan *bs = ctx->batch_spans.data() + first; // (valid until the next push_back: only used inside this call)
    job.out_span = (int)first + 2;
    const bool direct = (!lin_l || bs[0].how == HostSpan::kCallerPinned) && (!lin_r || bs[1].how == HostSpan::kCallerPinned);
    if (!lin_l || !lin_r || !direct) {
        // bytes that cross through this slot's pinned buffers (gathered cut-outs, stages): the slot's previous upload
        // from them must be through
        WS_HIP(ctx, hipEventSynchronize(job.ev_h2d));
        if (!lin_l) WS_HIP(ctx, gather_rows(job.h_left, left));
        if (!lin_r) WS_HIP(ctx, gather_rows(job.h_right, right));
    }
    if (lin_l) WS_HIP(ctx, span_upload(bs[0], 0, d_left, span_l, cs));
    else WS_HIP(ctx, hipMemcpyAsync(d_left, job.h_left.p, span_l, hipMemcpyHostToDevice, cs));
    if (lin_r) WS_HIP(ctx, span_upload(bs[1], 0, d_right, span_r, cs));
    else WS_HIP(ctx, hipMemcpyAsync(d_right, job.h_right.p, span_r, hipMemcpyHostToDevice, cs));
    WS_HIP(ctx, hipEventRecord(job.ev_h2d, cs));
    WS_HIP(ctx, hipStreamWaitEvent(ctx->stream, job.ev_h2d, 0));
    ws_image dl{d_left, left->width, left->height, lin_l ? left->stride : (int)lb};
    ws_image dr{d_right, right->width, right->height, lin_r ? right->stride : (int)rb};
    job.wire = wire_for(p, left, right);
    if ((rc = run_device_wire(ctx, p, &dl, &dr, job.d_out, job.d_out16, job.wire, ow, ctx->stream)) != WS_OK) return rc;
    WS_HIP(ctx, hipEventRecord(job.ev_done, ctx->stream));
    job.user_out = out; job.w = ow; job.h = oh; job.out_stride = out_stride; job.dtype = out_dtype;
    job.pending = true;
    ctx->job_next ^= 1;
    // the previous pair's map goes down while this one is searched
    return flush_job(ctx, prev);
}

int ws_wait(ws_context *ctx)
{
    if (!ctx) return WS_ERR_ARG;
    WS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = flush_job(ctx, ctx->jobs[ctx->job_next]); // the older one first
    if (rc == WS_OK) rc = flush_job(ctx, ctx->jobs[ctx->job_next ^ 1]);
    for (Job &j : ctx->jobs) j.pending = false; // (after an error nothing stays queued for a later batch)
    const hipError_t e1 = hipStreamSynchronize(ctx->copy_stream), e2 = hipStreamSynchronize(ctx->stream);
    if (ctx->batch_spans.size() >= 3)
        for (int i = 0; i < 3; ++i) ctx->last_how[i] = (int)ctx->batch_spans[ctx->batch_spans.size() - 3 + (size_t)i].how;
    if (e1 != hipSuccess || e2 != hipSuccess)
        for (HostSpan &b : ctx->batch_spans) b.down.clear(); // (nothing half-done reaches the caller)
    spans_finish(ctx->batch_spans.data(), (int)ctx->batch_spans.size());
    ctx->batch_spans.clear();
    ctx->jobs[0].out_span = ctx->jobs[1].out_span = -1;
    if (e1 != hipSuccess || e2 != hipSuccess)
        return fail(ctx, WS_ERR_HIP, "ws_wait: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return rc == WS_OK ? check_device_status(ctx) : rc;
}

static bool invert3x3(const double m[9], double out[9])
{
    // closed form (adjugate / determinant), as cv::invert does for 3x3 matrices
    const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                     m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0.0) return false;
    const double r = 1.0 / d;
    out[0] = (m[4] * m[8] - m[5] * m[7]) * r;
    out[1] = (m[2] * m[7] - m[1] * m[8]) * r;
    out[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    out[3] = (m[5] * m[6] - m[3] * m[8]) * r;
    out[4] = (m[0] * m[8] - m[2] * m[6]) * r;
    out[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    out[6] = (m[3] * m[7] - m[4] * m[6]) * r;
    out[7] = (m[1] * m[6] - m[0] * m[7]) * r;
    out[8] = (m[0] * m[4] - m[1] * m[3]) * r;
    return true;
}

int ws_warp_nearest_device(ws_context *ctx, const float *src_dev, int src_w, int src_h, int src_stride,
                           const double m[9], float *dst_dev, int dst_w, int dst_h, int dst_stride, void *stream)
{
    if (!ctx) return WS_ERR_ARG;
    if (!src_dev || !dst_dev || !m || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0 ||
        src_stride < src_w || dst_stride < dst_w)
        return fail(ctx, WS_ERR_ARG, "bad warp arguments");
    double inv[9];
    if (!invert3x3(m, inv)) return fail(ctx, WS_ERR_ARG, "singular warp matrix");
    WS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    WS_HIP(ctx, launch_warp(src_dev, src_w, src_h, src_stride, dst_dev, dst_w, dst_h, dst_stride, inv, s));
    return WS_OK;
}

int ws_warp_nearest_host(ws_context *ctx, const double *src, int src_w, int src_h, int src_stride,
                         const double m[9], double *dst, int dst_w, int dst_h, int dst_stride)
{
    if (!ctx) return WS_ERR_ARG;
    if (!src || !dst || !m || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0 || src_stride < src_w ||
        dst_stride < dst_w)
        return fail(ctx, WS_ERR_ARG, "bad warp arguments");
    // disparity maps are integer valued (or f32 sub-pixel): f32 on the device, CV_64F at the boundary; the floats
    // cross from / to pinned memory of the library's own
    WS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ns = (size_t)src_w * src_h, nd = (size_t)dst_w * dst_h;
    WS_HIP(ctx, host_ensure(ctx->h_left, ns * 4));
    WS_HIP(ctx, host_ensure(ctx->h_right, nd * 4));
    float *hs = reinterpret_cast<float *>(ctx->h_left.p), *hd = reinterpret_cast<float *>(ctx->h_right.p);
    for (int y = 0; y < src_h; ++y)
        for (int x = 0; x < src_w; ++x) hs[(size_t)y * src_w + x] = (float)src[(size_t)y * src_stride + x];
    int rc;
    if ((rc = ensure(ctx, ctx->d_out, ns * 4)) != WS_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_out64, nd * 4)) != WS_OK) return rc;
    WS_HIP(ctx, hipMemcpyAsync(ctx->d_out.p, hs, ns * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = ws_warp_nearest_device(ctx, static_cast<const float *>(ctx->d_out.p), src_w, src_h, src_w, m,
                                static_cast<float *>(ctx->d_out64.p), dst_w, dst_h, dst_w, ctx->stream);
    if (rc != WS_OK) return rc;
    WS_HIP(ctx, hipMemcpyAsync(hd, ctx->d_out64.p, nd * 4, hipMemcpyDeviceToHost, ctx->stream));
    WS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int y = 0; y < dst_h; ++y)
        for (int x = 0; x < dst_w; ++x) dst[(size_t)y * dst_stride + x] = (double)hd[(size_t)y * dst_w + x];
    return WS_OK;
}

// ---- consumers of the map (src/Reconstruction/reconstruction.cpp) ----------------------

int ws_remove_disparity_outliers(ws_context *ctx, float *map, int width, int height, int stride, int kernel_size,
                                 float thr_front, float thr_back)
{
    if (!ctx) return WS_ERR_ARG;
    if (!map || width <= 0 || height <= 0 || stride < width || kernel_size < 1)
        return fail(ctx, WS_ERR_ARG, "bad removeDisparityOutliers arguments");
    WS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)width * height;
    int rc;
    if ((rc = ensure(ctx, ctx->d_out, n * 4)) != WS_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_out64, std::max(n * 8, outliers_u32_scratch_bytes(width, height, ctx->num_cus)))) != WS_OK) return rc;
    if (!ctx->d_flag.p) {
        if ((rc = ensure(ctx, ctx->d_flag, 256)) != WS_OK) return rc;
        WS_HIP(ctx, hipMemsetAsync(ctx->d_flag.p, 0, 256, s));
    }
    float *dmap = static_cast<float *>(ctx->d_out.p);
    // the caller's map for the duration of the call (HostSpan, like ws_search_host: no pageable copies)
    HostSpan sp[1];
    sp[0].p = reinterpret_cast<uint8_t *>(map); sp[0].n = ((size_t)stride * (height - 1) + width) * 4; sp[0].stage = &ctx->h_out;
    spans_attach(sp, 1);
    // 8-bit maps (the pipeline's PNG: integers in [0, 255]) take the 32-bit integer kernels; a map with any other value
    // raises status word 1, is left as uploaded, and goes through the double kernels after the first synchronisation
    const bool try_u32 = ctx->d_flag.p && outliers_u32_applies(width, height, kernel_size, ctx->num_cus);
    rc = [&]() -> int {
        WS_HIP(ctx, span_upload_rows(sp[0], 0, (size_t)stride * 4, dmap, (size_t)width * 4, (size_t)height, s));
        if (try_u32)
            WS_HIP(ctx, launch_outliers_u32(dmap, width, width, height, kernel_size, thr_front, thr_back, static_cast<uint32_t *>(ctx->d_out64.p),
                                            static_cast<uint32_t *>(ctx->d_flag.p), ctx->status_dev + 1, ctx->num_cus, s));
        else
            WS_HIP(ctx, launch_outliers(dmap, width, width, height, kernel_size, thr_front, thr_back, static_cast<double *>(ctx->d_out64.p), s));
        WS_HIP(ctx, span_download_bytes(sp[0], 0, (size_t)stride * 4, dmap, (size_t)width * 4, (size_t)height, s));
        return WS_OK;
    }();
    hipError_t es = hipStreamSynchronize(s);
    ctx->last_outliers_path = try_u32 ? 1 : 0;
    if (rc == WS_OK && es == hipSuccess && try_u32 && ctx->status_host[1]) {
        ctx->status_host[1] = 0;
        ctx->last_outliers_path = 2;
        rc = [&]() -> int {
            WS_HIP(ctx, hipMemsetAsync(ctx->d_flag.p, 0, 256, s));
            WS_HIP(ctx, launch_outliers(dmap, width, width, height, kernel_size, thr_front, thr_back, static_cast<double *>(ctx->d_out64.p), s));
            WS_HIP(ctx, span_download_bytes(sp[0], 0, (size_t)stride * 4, dmap, (size_t)width * 4, (size_t)height, s));
            return WS_OK;
        }();
        es = hipStreamSynchronize(s);
    }
    spans_finish(sp, 1);
    if (rc == WS_OK && es != hipSuccess) return fail(ctx, WS_ERR_HIP, "removeDisparityOutliers: %s", hipGetErrorString(es));
    return rc;
}

static int depth_vertices_host(ws_context *ctx, const float *in, int width, int height, int stride, int input_is_depth,
                               float focal, float baseline, const float *k, const ws_image *bgr, float *depth,
                               int depth_stride, float *positions, uint8_t *colors)
{
    if (!ctx) return WS_ERR_ARG;
    if (!in || width <= 0 || height <= 0 || stride < width) return fail(ctx, WS_ERR_ARG, "bad map");
    if (positions && (!colors || !k || !bgr || !bgr->data || bgr->width != width || bgr->height != height ||
                      bgr->stride < 3 * width))
        return fail(ctx, WS_ERR_ARG, "back-projection needs K, a colour image of the map's size and both outputs");
    WS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)width * height;
    int rc;
    if ((rc = ensure(ctx, ctx->d_out, n * 4)) != WS_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_out64, n * 4 + n * 16 + n * 4)) != WS_OK) return rc;
    if (positions && (rc = ensure(ctx, ctx->d_left, n * 3)) != WS_OK) return rc;
    float *din = static_cast<float *>(ctx->d_out.p);
    uint8_t *base = static_cast<uint8_t *>(ctx->d_out64.p);
    float *dpos = reinterpret_cast<float *>(base);           // n * 16 bytes, 16-byte aligned
    float *ddepth = reinterpret_cast<float *>(base + n * 16); // n * 4
    uint8_t *dcol = base + n * 20;                            // n * 4
    // the caller's buffers for the duration of the call (HostSpan, like ws_search_host: no pageable copies)
    const bool lin_bgr = positions && linear_span(bgr);
    HostSpan sp[5];
    sp[0].p = reinterpret_cast<uint8_t *>(const_cast<float *>(in)); sp[0].n = ((size_t)stride * (height - 1) + width) * 4; sp[0].stage = &ctx->h_out;
    if (depth) { sp[1].p = reinterpret_cast<uint8_t *>(depth); sp[1].n = ((size_t)depth_stride * (height - 1) + width) * 4; sp[1].stage = &ctx->h_right; }
    if (positions) {
        sp[2].p = reinterpret_cast<uint8_t *>(positions); sp[2].n = n * 16; sp[2].stage = &ctx->h_aux[0];
        sp[3].p = colors; sp[3].n = n * 4; sp[3].stage = &ctx->h_aux[1];
    }
    if (lin_bgr) { sp[4].p = const_cast<uint8_t *>(bgr->data); sp[4].n = (size_t)bgr->stride * (height - 1) + (size_t)width * 3; sp[4].stage = &ctx->h_left; }
    spans_attach(sp, 5);
    rc = [&]() -> int {
        int rc2;
        WS_HIP(ctx, span_upload_rows(sp[0], 0, (size_t)stride * 4, din, (size_t)width * 4, (size_t)height, s));
        int bgr_stride = width * 3;
        if (positions) { // the colour image as one linear copy with its own row stride (see ws_search_host)
            if (lin_bgr) {
                bgr_stride = bgr->stride;
                if ((rc2 = ensure(ctx, ctx->d_left, sp[4].n)) != WS_OK) return rc2;
                WS_HIP(ctx, span_upload(sp[4], 0, ctx->d_left.p, sp[4].n, s));
            } else { // a cut-out of a much wider image: gathered rows from pinned memory of our own
                WS_HIP(ctx, gather_rows(ctx->h_left, bgr));
                WS_HIP(ctx, hipMemcpyAsync(ctx->d_left.p, ctx->h_left.p, (size_t)width * 3 * height, hipMemcpyHostToDevice, s));
            }
        }
        WS_HIP(ctx, launch_depth_vertices(din, width, width, height, focal, baseline, k,
                                          static_cast<const uint8_t *>(ctx->d_left.p), bgr_stride, depth ? ddepth : nullptr, width,
                                          positions ? dpos : nullptr, positions ? dcol : nullptr, input_is_depth, s));
        if (depth) WS_HIP(ctx, span_download_bytes(sp[1], 0, (size_t)depth_stride * 4, ddepth, (size_t)width * 4, (size_t)height, s));
        if (positions) {
            WS_HIP(ctx, span_download_bytes(sp[2], 0, n * 16, dpos, n * 16, 1, s));
            WS_HIP(ctx, span_download_bytes(sp[3], 0, n * 4, dcol, n * 4, 1, s));
        }
        return WS_OK;
    }();
    const hipError_t es = hipStreamSynchronize(s);
    spans_finish(sp, 5);
    if (rc == WS_OK && es != hipSuccess) return fail(ctx, WS_ERR_HIP, "depth / vertices: %s", hipGetErrorString(es));
    return rc;
}

int ws_convert_disparity_to_depth(ws_context *ctx, const float *disp, int width, int height, int stride, float focal_length,
                                  float baseline, float *depth, int depth_stride)
{
    if (!depth || depth_stride < width) return ctx ? fail(ctx, WS_ERR_ARG, "bad depth output") : WS_ERR_ARG;
    return depth_vertices_host(ctx, disp, width, height, stride, 0, focal_length, baseline, nullptr, nullptr, depth,
                               depth_stride, nullptr, nullptr);
}

int ws_back_project(ws_context *ctx, const float *depth, int width, int height, int stride, const float intrinsics[9],
                    const ws_image *bgr, float *positions, uint8_t *colors)
{
    if (!positions || !colors) return ctx ? fail(ctx, WS_ERR_ARG, "null vertex output") : WS_ERR_ARG;
    return depth_vertices_host(ctx, depth, width, height, stride, 1, 0.0f, 0.0f, intrinsics, bgr, nullptr, 0, positions, colors);
}

int ws_timer_begin(ws_context *ctx, void *stream)
{
    if (!ctx) return WS_ERR_ARG;
    WS_HIP(ctx, hipSetDevice(ctx->device));
    WS_HIP(ctx, hipEventRecord(ctx->ev0, stream ? static_cast<hipStream_t>(stream) : ctx->stream));
    return WS_OK;
}

int ws_timer_end(ws_context *ctx, void *stream, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return WS_ERR_ARG;
    WS_HIP(ctx, hipEventRecord(ctx->ev1, stream ? static_cast<hipStream_t>(stream) : ctx->stream));
    WS_HIP(ctx, hipEventSynchronize(ctx->ev1));
    WS_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return WS_OK;
}

int ws_set_profiling(ws_context *ctx, int enable)
{
    if (!ctx) return WS_ERR_ARG;
    ctx->profiling = enable != 0;
    ctx->kernel_timed = false;
    return WS_OK;
}

int ws_last_kernel_ms(ws_context *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return WS_ERR_ARG;
    if (!ctx->kernel_timed) return fail(ctx, WS_ERR_ARG, "no marching-kernel launch was timed (ws_set_profiling off, or the generic path ran)");
    WS_HIP(ctx, hipEventSynchronize(ctx->evk1));
    WS_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->evk0, ctx->evk1));
    return WS_OK;
}

int ws_last_max_block(ws_context *ctx, int block_size, int *max_block)
{
    if (!ctx || !max_block) return WS_ERR_ARG;
    *max_block = block_size;
    if (!ctx->var_block_ran) return WS_OK;
    int v = 0;
    WS_HIP(ctx, hipSetDevice(ctx->device));
    WS_HIP(ctx, hipDeviceSynchronize());
    WS_HIP(ctx, hipMemcpy(&v, ctx->max_block.p, sizeof v, hipMemcpyDeviceToHost));
    if (v > block_size) *max_block = v;
    return WS_OK;
}

int ws_last_launch_info(const ws_context *ctx, char *kernel_name, int name_cap, int *threads,
                        int *workgroups, int *lds_bytes)
{
    if (!ctx) return WS_ERR_ARG;
    if (kernel_name && name_cap > 0) {
        strncpy(kernel_name, ctx->last_kernel.c_str(), (size_t)name_cap - 1);
        kernel_name[name_cap - 1] = 0;
    }
    if (threads) *threads = ctx->last_threads;
    if (workgroups) *workgroups = ctx->last_wgs;
    if (lds_bytes) *lds_bytes = ctx->last_lds;
    return WS_OK;
}

int ws_set_tuning(ws_context *ctx, int x_runs_per_tile, int strip_rows, int threads)
{
    if (!ctx || x_runs_per_tile < 0 || strip_rows < 0 || threads < 0) return WS_ERR_ARG;
    ctx->tune_nxr = x_runs_per_tile;
    ctx->tune_rows = strip_rows;
    ctx->tune_threads = threads;
    return WS_OK;
}

// ---- Middlebury plumbing ---------------------------------------------------------------

void ws_free(void *p) { free(p); }

int ws_pfm_read(const char *path, float **data, int *width, int *height)
{
    if (!path || !data || !width || !height) return WS_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) return WS_ERR_IO;
    char tag[8] = {0};
    int w = 0, h = 0;
    double scale = 0;
    // "Pf" = one channel; header fields are whitespace separated, one whitespace byte before data
    if (fscanf(f, "%7s %d %d %lf", tag, &w, &h, &scale) != 4 || strcmp(tag, "Pf") != 0 || w <= 0 || h <= 0 ||
        scale == 0) {
        fclose(f);
        return WS_ERR_IO;
    }
    fgetc(f);
    float *buf = static_cast<float *>(malloc((size_t)w * h * sizeof(float)));
    if (!buf) { fclose(f); return WS_ERR_NOMEM; }
    const uint16_t probe = 1;
    const bool host_little = *reinterpret_cast<const uint8_t *>(&probe) == 1;
    const bool file_little = scale < 0;
    for (int y = h - 1; y >= 0; --y) { // the file stores the bottom row first
        float *row = buf + (size_t)y * w;
        if (fread(row, sizeof(float), (size_t)w, f) != (size_t)w) { free(buf); fclose(f); return WS_ERR_IO; }
        if (host_little != file_little)
            for (int x = 0; x < w; ++x) {
                uint32_t v;
                memcpy(&v, row + x, 4);
                v = (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
                memcpy(row + x, &v, 4);
            }
    }
    fclose(f);
    *data = buf; *width = w; *height = h;
    return WS_OK;
}

int ws_pfm_write(const char *path, const float *data, int width, int height, int stride)
{
    if (!path || !data || width <= 0 || height <= 0 || stride < width) return WS_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) return WS_ERR_IO;
    const uint16_t probe = 1;
    const bool host_little = *reinterpret_cast<const uint8_t *>(&probe) == 1;
    fprintf(f, "Pf\n%d %d\n%s\n", width, height, host_little ? "-1.0" : "1.0");
    for (int y = height - 1; y >= 0; --y)
        if (fwrite(data + (size_t)y * stride, sizeof(float), (size_t)width, f) != (size_t)width) { fclose(f); return WS_ERR_IO; }
    return fclose(f) == 0 ? WS_OK : WS_ERR_IO;
}

// Binary PPM ("P6", maxval 255) <-> BGR rows, standing in for cv::imread(IMREAD_COLOR) / imwrite
// of the reference's PNGs (data_loader.cpp:71-72): no PNG decoder is linked here.
int ws_ppm_read(const char *path, uint8_t **bgr, int *width, int *height)
{
    if (!path || !bgr || !width || !height) return WS_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) return WS_ERR_IO;
    char tag[3] = {0};
    int vals[3], n = 0;
    if (fread(tag, 1, 2, f) != 2 || tag[0] != 'P' || tag[1] != '6') { fclose(f); return WS_ERR_IO; }
    while (n < 3) { // width, height, maxval with '#' comments allowed between them
        int c = fgetc(f);
        if (c == EOF) { fclose(f); return WS_ERR_IO; }
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') continue;
        ungetc(c, f);
        if (fscanf(f, "%d", &vals[n]) != 1) { fclose(f); return WS_ERR_IO; }
        ++n;
    }
    fgetc(f); // the single whitespace byte before the pixels
    const int w = vals[0], h = vals[1];
    if (w <= 0 || h <= 0 || vals[2] != 255) { fclose(f); return WS_ERR_IO; }
    uint8_t *buf = static_cast<uint8_t *>(malloc((size_t)w * h * 3));
    if (!buf) { fclose(f); return WS_ERR_NOMEM; }
    if (fread(buf, 3, (size_t)w * h, f) != (size_t)w * h) { free(buf); fclose(f); return WS_ERR_IO; }
    fclose(f);
    for (size_t i = 0; i < (size_t)w * h; ++i) std::swap(buf[3 * i], buf[3 * i + 2]); // RGB -> BGR
    *bgr = buf; *width = w; *height = h;
    return WS_OK;
}

int ws_ppm_write(const char *path, const uint8_t *bgr, int width, int height, int stride)
{
    if (!path || !bgr || width <= 0 || height <= 0 || stride < 3 * width) return WS_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) return WS_ERR_IO;
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<uint8_t> row((size_t)width * 3);
    for (int y = 0; y < height; ++y) {
        const uint8_t *p = bgr + (size_t)y * stride;
        for (int x = 0; x < width; ++x) { row[3 * x] = p[3 * x + 2]; row[3 * x + 1] = p[3 * x + 1]; row[3 * x + 2] = p[3 * x]; }
        if (fwrite(row.data(), 1, row.size(), f) != row.size()) { fclose(f); return WS_ERR_IO; }
    }
    return fclose(f) == 0 ? WS_OK : WS_ERR_IO;
}

static bool parse_cam(const char *line, float m[9])
{
    // "cam0=[fx 0 cx; 0 fy cy; 0 0 1]": drop 6 leading characters and the closing bracket,
    // semicolons become blanks (data_loader.cpp:148-154)
    std::string s(line);
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
    if (s.size() < 8) return false;
    s = s.substr(6, s.size() - 7);
    std::replace(s.begin(), s.end(), ';', ' ');
    return sscanf(s.c_str(), "%f %f %f %f %f %f %f %f %f", m, m + 1, m + 2, m + 3, m + 4, m + 5, m + 6, m + 7, m + 8) == 9;
}

// WriteMesh (reconstruction.cpp:72-149) with CheckTriangularValidity (:46-69): COFF text, every
// vertex written (invalid ones as "0 0 0"), two triangles per grid cell when all three corners are
// valid and every edge is at most edge_threshold long.  Host only: file I/O bound.
static bool mesh_triangle_ok(const float *pos, unsigned a, unsigned b, unsigned c, float thr)
{
    const float minf = -INFINITY;
    if (pos[4 * a] == minf || pos[4 * b] == minf || pos[4 * c] == minf) return false;
    auto len = [&](unsigned p, unsigned q) {
        return sqrtf(powf(pos[4 * p] - pos[4 * q], 2) + powf(pos[4 * p + 1] - pos[4 * q + 1], 2) +
                     powf(pos[4 * p + 2] - pos[4 * q + 2], 2));
    };
    return !(len(a, b) > thr || len(a, c) > thr || len(b, c) > thr);
}

int ws_write_mesh_off(const char *path, const float *positions, const uint8_t *colors, int width, int height,
                      float edge_threshold)
{
    if (!path || !positions || !colors || width <= 0 || height <= 0) return WS_ERR_ARG;
    std::vector<unsigned> tri;
    const unsigned w = (unsigned)width, h = (unsigned)height;
    for (unsigned y = 0; y + 1 < h; ++y)
        for (unsigned x = 0; x + 1 < w; ++x) {
            const unsigned i00 = y * w + x, i10 = (y + 1) * w + x, i01 = y * w + x + 1, i11 = (y + 1) * w + x + 1;
            if (mesh_triangle_ok(positions, i00, i10, i01, edge_threshold)) { tri.push_back(i00); tri.push_back(i10); tri.push_back(i01); }
            if (mesh_triangle_ok(positions, i10, i11, i01, edge_threshold)) { tri.push_back(i10); tri.push_back(i11); tri.push_back(i01); }
        }
    std::ofstream out(path);
    if (!out.is_open()) return WS_ERR_IO;
    out << "COFF" << std::endl;
    out << (size_t)w * h << " " << tri.size() / 3 << " 0" << std::endl;
    const float minf = -INFINITY;
    for (size_t n = 0; n < (size_t)w * h; ++n) {
        if (positions[4 * n] == minf) out << "0 0 0 ";
        else out << positions[4 * n] << " " << positions[4 * n + 1] << " " << positions[4 * n + 2] << " ";
        out << (unsigned)colors[4 * n] << " " << (unsigned)colors[4 * n + 1] << " " << (unsigned)colors[4 * n + 2] << " "
            << (unsigned)colors[4 * n + 3] << std::endl;
    }
    for (size_t n = 0; n < tri.size() / 3; ++n)
        out << "3 " << tri[3 * n] << " " << tri[3 * n + 1] << " " << tri[3 * n + 2] << std::endl;
    out.close();
    return out.fail() ? WS_ERR_IO : WS_OK;
}

int ws_calib_read(const char *path, ws_calib *out)
{
    if (!path || !out) return WS_ERR_ARG;
    FILE *f = fopen(path, "r");
    if (!f) return WS_ERR_IO;
    memset(out, 0, sizeof *out);
    out->doffs = out->baseline = -1.0f;
    out->width = out->height = out->ndisp = -1;
    char line[512];
    int n = 0;
    bool ok = true;
    while (fgets(line, sizeof line, f)) {
        if (n == 0) ok = ok && parse_cam(line, out->cam0);
        else if (n == 1) ok = ok && parse_cam(line, out->cam1);
        else {
            float v;
            if (sscanf(line, "doffs=%f", &v) == 1) out->doffs = v;
            else if (sscanf(line, "baseline=%f", &v) == 1) out->baseline = v;
            else if (sscanf(line, "width=%f", &v) == 1) out->width = (int)v;
            else if (sscanf(line, "height=%f", &v) == 1) out->height = (int)v;
            else if (sscanf(line, "ndisp=%f", &v) == 1) out->ndisp = (int)v;
        }
        ++n;
    }
    fclose(f);
    return (ok && n >= 2) ? WS_OK : WS_ERR_IO;
}

int ws_evaldisp(const float *disp, const float *gt, const uint8_t *mask, int width, int height,
                float badthresh, float maxdisp, int rounddisp, double res[6])
{
    if (!disp || !gt || !mask || !res || width <= 0 || height <= 0) return WS_ERR_ARG;
    int n = 0, bad = 0, invalid = 0;
    float serr = 0;
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const size_t o = (size_t)y * width + x;
            const float g = gt[o];
            if (g == INFINITY) continue;                    // unknown (utils.cpp:137)
            float d = disp[o];
            const bool valid = d != 0;                      // utils.cpp:140
            if (valid) d = fmaxf(0.0f, fminf(maxdisp, d));
            if (valid && rounddisp) d = roundf(d);
            const float err = fabsf(d - g);
            if (mask[o] != 255) continue;                   // utils.cpp:146
            ++n;
            if (valid) { serr += err; if (err > badthresh) ++bad; }
            else ++invalid;
        }
    res[0] = n;
    res[1] = (float)(100.0 * bad / n);
    res[2] = (float)(100.0 * invalid / n);
    res[3] = (float)(100.0 * (bad + invalid) / n);
    res[4] = serr / (float)(n - invalid);
    res[5] = 100.0 * n / ((double)width * height);
    return WS_OK;
}

} // extern "C"
