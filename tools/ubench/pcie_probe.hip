// Probe (GPU box): what the boundary call's copies can cost.  Pageable vs registered host memory, copy sizes of
// config 2 (2 x 4.5 MB up, 6 MB f32 / 12 MB f64 down), register / unregister prices, device kernels writing
// straight into mapped host memory, and up + down at once from two host threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> static double med(F f, int n = 9) { std::vector<double> t; for (int i = 0; i < n; ++i) { double a = now(); f(); t.push_back(now() - a); } std::sort(t.begin(), t.end()); return t[n / 2] * 1e3; }
__global__ void widen(const float *s, double *d, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (double)s[i]; }
int main()
{
    const size_t up = 4500000, dn32 = 6000000, dn64 = 12000000;
    char *h_up = (char *)aligned_alloc(4096, up * 2), *h_dn = (char *)aligned_alloc(4096, dn64);
    memset(h_up, 1, up * 2); memset(h_dn, 0, dn64);
    char *d_up, *d_dn; float *d_f32;
    CK(hipMalloc(&d_up, up * 2)); CK(hipMalloc(&d_dn, dn64)); CK(hipMalloc(&d_f32, dn32));
    CK(hipMemset(d_f32, 0, dn32));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int i = 0; i < 3; ++i) { CK(hipMemcpy(d_up, h_up, up * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(h_dn, d_dn, dn64, hipMemcpyDeviceToHost)); }
    printf("pageable H2D 9 MB (2 copies)      %.3f ms\n", med([&] { CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(d_up + up, h_up + up, up, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }));
    printf("pageable H2D: time inside the two async calls alone %.3f ms\n", med([&] { CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(d_up + up, h_up + up, up, hipMemcpyHostToDevice, s1)); }));
    CK(hipStreamSynchronize(s1));
    printf("pageable D2H 6 MB                 %.3f ms\n", med([&] { CK(hipMemcpyAsync(h_dn, d_dn, dn32, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); }));
    printf("pageable D2H 12 MB                %.3f ms\n", med([&] { CK(hipMemcpyAsync(h_dn, d_dn, dn64, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); }));
    printf("pageable D2H 12 MB: time inside the async call alone %.3f ms\n", med([&] { CK(hipMemcpyAsync(h_dn, d_dn, dn64, hipMemcpyDeviceToHost, s1)); }));
    CK(hipStreamSynchronize(s1));
    printf("hipHostRegister 12 MB             %.3f ms\n", med([&] { CK(hipHostRegister(h_dn, dn64, hipHostRegisterDefault)); CK(hipHostUnregister(h_dn)); }));
    printf("hipHostRegister 9 MB              %.3f ms\n", med([&] { CK(hipHostRegister(h_up, up * 2, hipHostRegisterDefault)); CK(hipHostUnregister(h_up)); }));
    CK(hipHostRegister(h_dn, dn64, hipHostRegisterMapped)); CK(hipHostRegister(h_up, up * 2, hipHostRegisterDefault));
    printf("registered H2D 9 MB               %.3f ms\n", med([&] { CK(hipMemcpyAsync(d_up, h_up, up * 2, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }));
    printf("registered D2H 12 MB              %.3f ms\n", med([&] { CK(hipMemcpyAsync(h_dn, d_dn, dn64, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); }));
    printf("registered D2H 6 MB               %.3f ms\n", med([&] { CK(hipMemcpyAsync(h_dn, d_dn, dn32, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); }));
    printf("registered up 9 MB + down 12 MB on two streams %.3f ms\n", med([&] { CK(hipMemcpyAsync(d_up, h_up, up * 2, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h_dn, d_dn, dn64, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); }));
    double *map = nullptr; CK(hipHostGetDevicePointer((void **)&map, h_dn, 0));
    printf("kernel widening 1.5 M floats straight into mapped host memory (12 MB) %.3f ms\n", med([&] { hipLaunchKernelGGL(widen, dim3(1024), dim3(256), 0, s1, d_f32, map, dn32 / 4); CK(hipStreamSynchronize(s1)); }));
    CK(hipHostUnregister(h_dn)); CK(hipHostUnregister(h_up));
    printf("pageable up 9 MB + down 12 MB from two host threads %.3f ms\n", med([&] {
        std::thread t([&] { CK(hipMemcpyAsync(h_dn, d_dn, dn64, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
        CK(hipMemcpyAsync(d_up, h_up, up * 2, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); t.join(); }));
    void *pin = nullptr; CK(hipHostMalloc(&pin, dn64, hipHostMallocDefault));
    printf("D2H 12 MB into a hipHostMalloc buffer %.3f ms, then memcpy to the pageable one %.3f ms\n",
           med([&] { CK(hipMemcpyAsync(pin, d_dn, dn64, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); }), med([&] { memcpy(h_dn, pin, dn64); }));
    return 0;
}
