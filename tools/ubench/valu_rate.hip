// Micro-benchmark: sustained issue rate of the VALU instructions the marching kernel is made of.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x5bd1e995u;
    uint32_t acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) acc[c] = __builtin_amdgcn_sad_u8(a, b + c, acc[c]);
                if (OP == 1) acc[c] = __builtin_amdgcn_udot4(a, b + c, acc[c], false);
                if (OP == 2) acc[c] = (acc[c] << 3) + a;          // v_lshl_add_u32
                if (OP == 3) acc[c] = min((int)acc[c], (int)(a + c)) + 1; // v_min + v_add
                if (OP == 4) acc[c] = acc[c] - (b + c);           // v_sub
                if (OP == 5) acc[c] = __builtin_fmaf(__uint_as_float(acc[c]), 1.0001f, 0.5f); // v_fma_f32
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r ^= acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP, int CHAINS>
void run(const char *name, int waves_per_simd)
{
    uint32_t *d;
    hipMalloc(&d, 256 * 1024 * 16 * 4);
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd; // 256-thread blocks = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 16 * CHAINS * waves_per_simd * (OP == 3 ? 2 : 1);
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-12s chains=%d waves/SIMD=%d : %.3f ms  -> %.2f cycles(@2.4GHz)/wave-instr/SIMD, %.2f Tlane-op/s\n", name, CHAINS,
           waves_per_simd, ms, cyc / insts_per_simd, insts_per_simd * 1024 * 64 / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0, 1>("sad_u8", w); run<0, 4>("sad_u8", w);
        run<1, 1>("dot4", w);   run<1, 2>("dot4", w); run<1, 4>("dot4", w);
        run<2, 4>("lshl_add", w);
        run<3, 4>("min+add", w);
        run<4, 4>("sub", w);
        run<5, 4>("fma_f32", w);
    }
    return 0;
}
