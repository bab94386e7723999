// Probe (GPU box): can a device-to-host copy into registered memory run BESIDE a kernel?  The banded host call
// assumes so; a rocprofv3 trace shows the runtime's linear D2H as a blit kernel (__amd_rocclr_copyBuffer) next to
// which every other kernel loses about the copy's duration.  Cases: the linear copy, a pitched (2-D) copy, a
// copy into hipHostMalloc memory -- each alone and beside a memory-bound and a compute-bound kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> static double med(F f, int n = 9) { std::vector<double> t; for (int i = 0; i < n; ++i) { double a = now(); f(); t.push_back(now() - a); } std::sort(t.begin(), t.end()); return t[n / 2] * 1e3; }
__global__ void stream_copy(const uint4 *s, uint4 *d, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i]; }
__global__ void spin(float *p, int iters) { float a = threadIdx.x, b = 1.0001f; for (int i = 0; i < iters; ++i) a = a * b + 0.5f; if (a == 12345.f) p[0] = a; }
int main()
{
    const size_t W = 1500 * 8, H = 500, PITCH = W + 256, bytes = W * H; // one band of config 2's CV_64F map: 6 MB
    char *h_reg = (char *)aligned_alloc(4096, bytes), *h_pin = nullptr;
    memset(h_reg, 0, bytes);
    CK(hipHostRegister(h_reg, bytes, hipHostRegisterDefault));
    CK(hipHostMalloc((void **)&h_pin, bytes, hipHostMallocDefault));
    char *d_lin, *d_pit; uint4 *d_a, *d_b; float *d_f;
    const size_t big = 64u << 20;
    CK(hipMalloc(&d_lin, bytes)); CK(hipMalloc(&d_pit, PITCH * H)); CK(hipMalloc(&d_a, big)); CK(hipMalloc(&d_b, big)); CK(hipMalloc(&d_f, 4096));
    hipStream_t sc, sk; CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    auto mem_kernel = [&] { hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, sk, d_a, d_b, big / 16); };      // ~64 MB read + write
    auto alu_kernel = [&] { hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, sk, d_f, 20000); };
    auto lin = [&](char *h) { CK(hipMemcpyAsync(h, d_lin, bytes, hipMemcpyDeviceToHost, sc)); };
    auto rect = [&](char *h) { CK(hipMemcpy2DAsync(h, W, d_pit, PITCH, W, H, hipMemcpyDeviceToHost, sc)); };
    auto both = [&](auto copy, auto kern) { kern(); copy(); CK(hipStreamSynchronize(sc)); CK(hipStreamSynchronize(sk)); };
    auto only_k = [&](auto kern) { kern(); CK(hipStreamSynchronize(sk)); };
    auto only_c = [&](auto copy) { copy(); CK(hipStreamSynchronize(sc)); };
    for (int i = 0; i < 3; ++i) { only_k(mem_kernel); only_k(alu_kernel); only_c([&] { lin(h_reg); }); only_c([&] { rect(h_reg); }); only_c([&] { lin(h_pin); }); }
    const double km = med([&] { only_k(mem_kernel); }), ka = med([&] { only_k(alu_kernel); });
    printf("memory-bound kernel alone %.3f ms, compute-bound kernel alone %.3f ms\n", km, ka);
    struct Case { const char *name; int kind; char *h; } cases[] = {{"linear D2H 6 MB -> registered", 0, h_reg}, {"pitched (2-D) D2H 6 MB -> registered", 1, h_reg},
                                                                    {"linear D2H 6 MB -> hipHostMalloc", 0, h_pin}, {"pitched (2-D) D2H 6 MB -> hipHostMalloc", 1, h_pin}};
    for (const Case &c : cases) {
        auto copy = [&] { if (c.kind) rect(c.h); else lin(c.h); };
        const double alone = med([&] { only_c(copy); });
        const double bm = med([&] { both(copy, mem_kernel); }), ba = med([&] { both(copy, alu_kernel); });
        printf("%-42s alone %.3f ms | beside the memory-bound kernel %.3f (sum %.3f) | beside the compute-bound kernel %.3f (sum %.3f)\n",
               c.name, alone, bm, alone + km, ba, alone + ka);
    }
    return 0;
}
