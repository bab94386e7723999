// Probe (GPU box): what hipHostRegister / hipHostUnregister and the runtime's pageable copy path really do on this
// platform, and which combinations of the two end a process.  Round 2 saw two aborts "without a message" inside
// ws_search_host; this program asks the questions the code in hand raises, ONE scenario per child
// process (forked before anything touches HIP), so that a signal or a runtime abort is attributed to its scenario and
// whatever the runtime prints on stderr is kept (under pytest's fd capture such a message is lost with the process).
//
//   usage:  hostreg_probe [scenario ...]      (default: all, in order of increasing risk)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define SAY(...) do { printf(__VA_ARGS__); fflush(stdout); } while (0)
#define TRY(x) ([&]() { hipError_t e_ = (x); SAY("    %-78s -> %s\n", #x, e_ == hipSuccess ? "ok" : hipGetErrorName(e_)); if (e_ != hipSuccess) (void)hipGetLastError(); return e_; }())
#define MUST(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { SAY("    FATAL %s: %s\n", #x, hipGetErrorString(e_)); _exit(3); } } while (0)

static const size_t MB = 1 << 20;
static char *map_at(void *hint, size_t n, bool fixed = false)
{
    void *p = mmap(hint, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | (fixed ? MAP_FIXED_NOREPLACE : 0), -1, 0);
    return p == MAP_FAILED ? nullptr : (char *)p;
}
static void fill(char *p, size_t n, int seed) { for (size_t i = 0; i < n; i += 1) p[i] = (char)((i * 131 + seed) >> 3); }
static size_t mismatches(const char *p, size_t n, int seed) { size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += p[i] != (char)((i * 131 + seed) >> 3); return bad; }
static void attrs(const char *what, const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); SAY("    attributes(%s): %s\n", what, hipGetErrorName(e)); return; }
    SAY("    attributes(%s): type %d device %d devptr %p hostptr %p managed %d\n", what, (int)a.type, a.device, a.devicePointer, a.hostPointer, a.isManaged);
}

// ---------------------------------------------------------------------------------------------------------------------
static void s_timing()
{
    MUST(hipSetDevice(0));
    char *d; MUST(hipMalloc(&d, 16 * MB));
    for (int kind = 0; kind < 3; ++kind) {
        const size_t n = 12 * MB;
        char *h = kind == 1 ? (char *)aligned_alloc(4096, n) : map_at(nullptr, n);
        if (kind != 2) memset(h, 1, n);
        SAY("  %s 12 MB:\n", kind == 0 ? "mmap, touched" : kind == 1 ? "aligned_alloc, touched" : "mmap, never touched");
        for (int i = 0; i < 4; ++i) {
            double t0 = now(); hipError_t e1 = hipHostRegister(h, n, hipHostRegisterDefault); double t1 = now();
            if (i == 0) attrs("registered", h);
            MUST(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
            double t2 = now(); hipError_t e2 = hipHostUnregister(h); double t3 = now();
            SAY("    round %d: register %.1f us (%s), unregister %.1f us (%s)\n", i, (t1 - t0) * 1e6, hipGetErrorName(e1), (t3 - t2) * 1e6, hipGetErrorName(e2));
        }
        attrs("after unregister", h);
        // back to back without a copy in between (what pcie_probe timed in round 2)
        double t0 = now();
        for (int i = 0; i < 20; ++i) { MUST(hipHostRegister(h, n, hipHostRegisterDefault)); MUST(hipHostUnregister(h)); }
        SAY("    20 x (register + unregister), no copy: %.1f us each\n", (now() - t0) * 1e6 / 20);
        // the same range registered page by page larger / smaller
        t0 = now(); MUST(hipHostRegister(h, 64 * 1024, hipHostRegisterDefault)); double t1 = now(); MUST(hipHostUnregister(h));
        SAY("    register 64 KB: %.1f us\n", (t1 - t0) * 1e6);
    }
    // pageable copies: how long does the call itself block
    char *h = map_at(nullptr, 12 * MB); memset(h, 2, 12 * MB);
    hipStream_t s; MUST(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int i = 0; i < 3; ++i) {
        double t0 = now(); MUST(hipMemcpyAsync(d, h, 12 * MB, hipMemcpyHostToDevice, s)); double t1 = now(); MUST(hipStreamSynchronize(s)); double t2 = now();
        SAY("    pageable H2D 12 MB: call %.1f us, + sync %.1f us\n", (t1 - t0) * 1e6, (t2 - t1) * 1e6);
    }
    attrs("pageable after copies", h);
}

static void s_overlap()
{
    MUST(hipSetDevice(0));
    char *h = map_at(nullptr, 16 * MB); memset(h, 1, 16 * MB);
    SAY("  base range [0, 8 MB) registered, then:\n");
    TRY(hipHostRegister(h, 8 * MB, hipHostRegisterDefault));
    SAY("   the same range again\n");
    hipError_t again = TRY(hipHostRegister(h, 8 * MB, hipHostRegisterDefault));
    SAY("   a range inside it [1 MB, 2 MB)\n");
    hipError_t inside = TRY(hipHostRegister(h + 1 * MB, 1 * MB, hipHostRegisterDefault));
    SAY("   a range straddling its end [4 MB, 12 MB)\n");
    hipError_t strad = TRY(hipHostRegister(h + 4 * MB, 8 * MB, hipHostRegisterDefault));
    SAY("   a range that shares only its last page [8 MB - 100, 8 MB + 1 MB)\n");
    hipError_t page = TRY(hipHostRegister(h + 8 * MB - 100, 1 * MB, hipHostRegisterDefault));
    SAY("   unregister in the order inside, straddle, page, again, base\n");
    if (inside == hipSuccess) TRY(hipHostUnregister(h + 1 * MB));
    if (strad == hipSuccess) TRY(hipHostUnregister(h + 4 * MB));
    if (page == hipSuccess) TRY(hipHostUnregister(h + 8 * MB - 100));
    if (again == hipSuccess) TRY(hipHostUnregister(h));
    TRY(hipHostUnregister(h));
    SAY("   unregister once more (nothing left)\n");
    TRY(hipHostUnregister(h));
    SAY("   unregister a pointer in the middle of a registered range\n");
    TRY(hipHostRegister(h, 8 * MB, hipHostRegisterDefault));
    TRY(hipHostUnregister(h + 4096));
    TRY(hipHostUnregister(h));
    SAY("  two small buffers on one page (1000 bytes each, 1200 apart), and an unaligned start\n");
    hipError_t a = TRY(hipHostRegister(h + 100, 1000, hipHostRegisterDefault));
    hipError_t b = TRY(hipHostRegister(h + 1300, 1000, hipHostRegisterDefault));
    attrs("first small", h + 100);
    attrs("second small", h + 1300);
    if (b == hipSuccess) TRY(hipHostUnregister(h + 1300));
    if (a == hipSuccess) TRY(hipHostUnregister(h + 100));
    SAY("  a hipHostMalloc'd buffer registered again\n");
    char *pin; MUST(hipHostMalloc((void **)&pin, 1 * MB, hipHostMallocDefault));
    hipError_t c = TRY(hipHostRegister(pin, 1 * MB, hipHostRegisterDefault));
    if (c == hipSuccess) TRY(hipHostUnregister(pin));
    MUST(hipHostFree(pin));
}

// a copy whose host pointer lies inside a registered range but runs past its end (left/right views of one array:
// the second registration fails, round 2 then copied "unregistered")
static void s_straddle()
{
    MUST(hipSetDevice(0));
    hipStream_t s; MUST(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    char *h = map_at(nullptr, 16 * MB); fill(h, 16 * MB, 7);
    char *d; MUST(hipMalloc(&d, 16 * MB)); char *back = map_at(nullptr, 16 * MB);
    TRY(hipHostRegister(h, 8 * MB, hipHostRegisterDefault));
    SAY("  H2D of [4 MB, 12 MB): starts inside the registered [0, 8 MB), ends 4 MB past it\n");
    hipError_t e = TRY(hipMemcpyAsync(d, h + 4 * MB, 8 * MB, hipMemcpyHostToDevice, s));
    TRY(hipStreamSynchronize(s));
    if (e == hipSuccess) { MUST(hipMemcpy(back, d, 8 * MB, hipMemcpyDeviceToHost)); SAY("    bytes wrong after the round trip: %zu\n", (size_t)(memcmp(back, h + 4 * MB, 8 * MB) != 0)); }
    SAY("  D2H into [4 MB, 12 MB)\n");
    MUST(hipMemset(d, 0x5a, 8 * MB));
    e = TRY(hipMemcpyAsync(h + 4 * MB, d, 8 * MB, hipMemcpyDeviceToHost, s));
    TRY(hipStreamSynchronize(s));
    if (e == hipSuccess) { size_t bad = 0; for (size_t i = 4 * MB; i < 12 * MB; ++i) bad += h[i] != 0x5a; SAY("    bytes wrong: %zu\n", bad); }
    SAY("  2-D H2D, 1899 rows of 2999 bytes with pitch 6001 starting at byte 3001 (a cut-out next to a registered one)\n");
    e = TRY(hipMemcpy2DAsync(d, 2999, h + 3001, 6001, 2999, 1899, hipMemcpyHostToDevice, s));
    TRY(hipStreamSynchronize(s));
    TRY(hipHostUnregister(h));
}

// the runtime's own pinning of pageable copies, and addresses that come back with another size
static void s_recycle(int mode)
{
    MUST(hipSetDevice(0));
    hipStream_t s; MUST(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    char *d; MUST(hipMalloc(&d, 32 * MB)); MUST(hipMemset(d, 0x33, 32 * MB));
    char *hint = (char *)0x7e0000000000ull;
    for (int round = 0; round < 6; ++round) {
        const size_t n = (round % 3 == 0 ? 12 : round % 3 == 1 ? 6 : 20) * MB;
        char *h = map_at(hint, n, true);
        if (!h) { SAY("    mmap at the hint failed\n"); return; }
        fill(h, n, round);
        SAY("  round %d: %zu MB at %p\n", round, n / MB, (void *)h);
        if (mode == 1) TRY(hipHostRegister(h, n, hipHostRegisterDefault));
        if (mode == 2 && round % 2 == 0) TRY(hipHostRegister(h, n, hipHostRegisterDefault)); // ... and never unregistered
        if (mode == 2 && round % 2 == 1) { SAY("   (the previous, larger or smaller, mapping here was registered and never released)\n"); hipError_t e = TRY(hipHostRegister(h, n, hipHostRegisterDefault)); if (e == hipSuccess) TRY(hipHostUnregister(h)); }
        hipError_t e = TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
        TRY(hipStreamSynchronize(s));
        memset(h, 0, n);
        e = TRY(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
        TRY(hipStreamSynchronize(s));
        SAY("    bytes wrong after up + down: %zu\n", mismatches(h, n, round));
        if (mode == 1) TRY(hipHostUnregister(h));
        munmap(h, n);
    }
}

// the second failing shape of round 2: a 2-D copy of a cut-out of a 35 MB image from pageable memory, in a process
// that registers and releases other caller memory
static void s_cutout()
{
    MUST(hipSetDevice(0));
    hipStream_t s; MUST(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int W = 6000, H = 1944, w = 1481; // 35 MB, rows of 4443 bytes (not a multiple of 4)
    const size_t n = (size_t)W * 3 * H;
    char *d; MUST(hipMalloc(&d, n));
    for (int round = 0; round < 3; ++round) {
        char *img = (char *)malloc(n); fill(img, n, round);
        char *other = (char *)malloc(12 * MB); memset(other, 1, 12 * MB);
        TRY(hipHostRegister(other, 12 * MB, hipHostRegisterDefault));
        MUST(hipMemcpyAsync(d, other, 12 * MB, hipMemcpyHostToDevice, s));
        MUST(hipStreamSynchronize(s));
        TRY(hipHostUnregister(other));
        free(other);
        double t0 = now();
        TRY(hipMemcpy2DAsync(d, (size_t)w * 3, img + 999, (size_t)W * 3, (size_t)w * 3, H, hipMemcpyHostToDevice, s));
        TRY(hipStreamSynchronize(s));
        SAY("    2-D pageable H2D of %d rows x %d bytes: %.2f ms\n", H, w * 3, (now() - t0) * 1e3);
        t0 = now();
        TRY(hipMemcpy2DAsync(img + 999, (size_t)W * 3, d, (size_t)w * 3, (size_t)w * 3, H, hipMemcpyDeviceToHost, s));
        TRY(hipStreamSynchronize(s));
        SAY("    2-D pageable D2H: %.2f ms, bytes wrong in the image: %zu\n", (now() - t0) * 1e3, mismatches(img, n, round));
        free(img);
    }
}

// a registered range released while a copy that uses it is still queued on ANOTHER stream
static void s_early_release()
{
    MUST(hipSetDevice(0));
    hipStream_t s1, s2; MUST(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); MUST(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char *d; MUST(hipMalloc(&d, 64 * MB)); MUST(hipMemset(d, 0x44, 64 * MB));
    char *h = map_at(nullptr, 64 * MB); memset(h, 0, 64 * MB);
    TRY(hipHostRegister(h, 64 * MB, hipHostRegisterDefault));
    TRY(hipMemcpyAsync(h, d, 64 * MB, hipMemcpyDeviceToHost, s2));
    double t0 = now();
    TRY(hipHostUnregister(h)); // s2 not synchronised
    SAY("    unregister with the copy in flight took %.1f us\n", (now() - t0) * 1e6);
    TRY(hipStreamSynchronize(s2));
    size_t bad = 0; for (size_t i = 0; i < 64 * MB; ++i) bad += h[i] != 0x44;
    SAY("    bytes wrong: %zu\n", bad);
}


// memory the runtime already knows (the caller's own hipHostMalloc / hipHostRegister, a framework's pinned allocator):
// what does a second registration by a library do to it, and how can a library tell beforehand
static void range_of(const char *what, void *p)
{
    hipDeviceptr_t base = nullptr; size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p);
    if (e != hipSuccess) { (void)hipGetLastError(); SAY("    address range(%s): %s\n", what, hipGetErrorName(e)); return; }
    SAY("    address range(%s): base %+ld bytes from the pointer, size %zu\n", what, (long)((char *)base - (char *)p), size);
}
static void s_foreign()
{
    MUST(hipSetDevice(0));
    hipStream_t s; MUST(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    char *d; MUST(hipMalloc(&d, 8 * MB));
    char *pin; MUST(hipHostMalloc((void **)&pin, 4 * MB, hipHostMallocDefault)); memset(pin, 3, 4 * MB);
    char *plain = map_at(nullptr, 8 * MB); memset(plain, 4, 8 * MB);
    attrs("hipHostMalloc base", pin); attrs("hipHostMalloc + 1 MB", pin + MB); attrs("plain", plain);
    range_of("hipHostMalloc base", pin); range_of("hipHostMalloc + 1 MB", pin + MB); range_of("plain", plain);
    TRY(hipHostRegister(plain, 4 * MB, hipHostRegisterDefault));
    attrs("registered base", plain); attrs("registered + 1 MB", plain + MB); attrs("just past the registered range", plain + 4 * MB);
    range_of("registered base", plain); range_of("registered + 1 MB", plain + MB); range_of("past the range", plain + 4 * MB);
    SAY("  a part of a hipHostMalloc'd block registered again (not at its base), used, unregistered\n");
    hipError_t e = TRY(hipHostRegister(pin + MB, MB, hipHostRegisterDefault));
    range_of("hipHostMalloc + 1 MB after that", pin + MB);
    TRY(hipMemcpyAsync(d, pin + MB, MB, hipMemcpyHostToDevice, s)); TRY(hipStreamSynchronize(s));
    if (e == hipSuccess) TRY(hipHostUnregister(pin + MB));
    SAY("  the whole hipHostMalloc'd block registered again at its base, unregistered, then used and freed by its owner\n");
    e = TRY(hipHostRegister(pin, 4 * MB, hipHostRegisterDefault));
    if (e == hipSuccess) TRY(hipHostUnregister(pin));
    attrs("hipHostMalloc base afterwards", pin); range_of("hipHostMalloc base afterwards", pin);
    TRY(hipMemcpyAsync(d, pin, 4 * MB, hipMemcpyHostToDevice, s)); TRY(hipStreamSynchronize(s));
    SAY("    first byte still readable: %d\n", pin[0]);
    TRY(hipHostFree(pin));
}

// the same base pointer registered under two sizes (one key in the runtime's map, two entries in a (pointer, size)
// registry), released twice -- alone, and with an enclosing registration that starts below it
static void s_twice(int enclosed)
{
    MUST(hipSetDevice(0));
    char *h = map_at(nullptr, 16 * MB); memset(h, 1, 16 * MB);
    if (enclosed) TRY(hipHostRegister(h, 16 * MB, hipHostRegisterDefault));
    TRY(hipHostRegister(h + MB, 4 * MB, hipHostRegisterDefault));
    TRY(hipHostRegister(h + MB, 6 * MB, hipHostRegisterDefault));
    range_of("the twice-registered pointer", h + MB);
    TRY(hipHostUnregister(h + MB));
    range_of("after the first release", h + MB);
    TRY(hipHostUnregister(h + MB));
    if (enclosed) TRY(hipHostUnregister(h));
}
static void twice0() { s_twice(0); }
static void twice1() { s_twice(1); }

struct Scenario { const char *name; void (*fn)(); const char *what; };
static void recycle0() { s_recycle(0); }
static void recycle1() { s_recycle(1); }
static void recycle2() { s_recycle(2); }
static const Scenario kAll[] = {
    {"timing", s_timing, "what register / unregister cost and do"},
    {"overlap", s_overlap, "status codes for overlapping / repeated / page-sharing registrations"},
    {"cutout", s_cutout, "2-D pageable copies of a cut-out beside register / unregister traffic"},
    {"recycle-pageable", recycle0, "pageable copies at an address that comes back with other sizes"},
    {"recycle-registered", recycle1, "the same with register / unregister around every copy"},
    {"early-release", s_early_release, "unregister while a copy on another stream is in flight"},
    {"foreign", s_foreign, "memory the runtime already knows, registered again by a library"},
    {"twice", twice0, "one base pointer registered under two sizes, released twice"},
    {"twice-enclosed", twice1, "the same inside a larger registration that starts below it"},
    {"straddle", s_straddle, "copies that start inside a registered range and run past its end"},
    {"recycle-stale", recycle2, "a registration that outlives its mapping (caller freed early)"},
};

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (const Scenario &sc : kAll) {
        bool wanted = argc < 2;
        for (int i = 1; i < argc; ++i) wanted |= !strcmp(argv[i], sc.name);
        if (!wanted) continue;
        SAY("== %s: %s\n", sc.name, sc.what);
        const pid_t pid = fork(); // (nothing in this process has touched HIP)
        if (pid == 0) {
            dup2(1, 2); // the runtime's own messages belong to the scenario's record
            alarm(120);
            sc.fn();
            SAY("   scenario returned\n");
            _exit(0);
        }
        int st = 0;
        waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) SAY("== %s: KILLED BY SIGNAL %d (%s)\n\n", sc.name, WTERMSIG(st), strsignal(WTERMSIG(st)));
        else SAY("== %s: exit code %d\n\n", sc.name, WEXITSTATUS(st));
    }
    return 0;
}
