// Micro-benchmark: sustained issue rate of the VALU instructions the marching kernel is made of, per SIMD,
// at 1 / 2 / 4 / 8 waves per SIMD, and of the kernel's own instruction mix.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
// (tools/ubench/run.sh keeps the output under profiles/).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x5bd1e995u;
    uint32_t acc[CHAINS];
#pragma unroll
    // (per-lane values: with uniform ones the compiler runs the loops that never touch a or b -- lshl_add, add3, sub -- on the
    // SCALAR unit, and round 3's "12 cycles per v_lshl_add, 8 per v_add3" were 192 / 130 s_lshl + s_add per 64 updates)
    for (int c = 0; c < CHAINS; ++c) acc[c] = c + seed + a * (uint32_t)(2 * c + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) acc[c] = __builtin_amdgcn_sad_u8(a, b + c, acc[c]);
                if (OP == 1) acc[c] = __builtin_amdgcn_udot4(a, b + c, acc[c], false);
                if (OP == 2) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(acc[c]) : "v"(acc[c]), "v"(acc[(c + 1) % CHAINS])); // (spelled out: the compiler splits some into v_lshlrev + v_add)
                if (OP == 3) acc[c] = min((int)acc[c], (int)(a + c)) + 1; // v_min + v_add
                if (OP == 4) acc[c] = acc[c] - acc[(c + 1) % CHAINS];           // v_sub
                if (OP == 5) acc[c] = __builtin_fmaf(__uint_as_float(acc[c]), 1.0001f, 0.5f); // v_fma_f32
                if (OP == 6) acc[c] = (uint32_t)min(min((int)acc[c] + 0, (int)acc[(c + 1) % CHAINS] ^ (int)a), (int)acc[(c + 2) % CHAINS]); // v_xor + v_min3_i32
                if (OP == 7) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(acc[c]) : "v"(acc[c]), "v"(acc[(c + 1) % CHAINS]), "v"(acc[(c + 2) % CHAINS]));
                if (OP == 8) acc[c] = (uint32_t)__builtin_amdgcn_sdot4((int)a, (int)(b + c), (int)acc[c], false); // v_dot4_i32_i8
                if (OP == 9) { // v_pk_add_u16
                    u16x2 x = __builtin_bit_cast(u16x2, acc[c]), y = __builtin_bit_cast(u16x2, acc[(c + 1) % CHAINS]);
                    acc[c] = __builtin_bit_cast(uint32_t, (u16x2)(x + y));
                }
                if (OP == 10) { // v_pk_min_u16
                    u16x2 x = __builtin_bit_cast(u16x2, acc[c] + 0x00010001u), y = __builtin_bit_cast(u16x2, acc[(c + 1) % CHAINS]);
                    acc[c] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x, y));
                }
                if (OP == 11) acc[c] = __builtin_amdgcn_perm(acc[c], a, 0x05040100u + c); // v_perm_b32
                if (OP == 12) acc[c] = (acc[c] & 0xffffffu) * (a & 0xffffffu) + b; // v_mad_u32_u24
                if (OP == 13) acc[c] = __builtin_amdgcn_udot4(a, b + c, acc[c], false) - a;   // dot4 + dependent sub
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r ^= acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// The marching step's own mix for one (thread, disparity pair): two prefix chains of 14 v_dot4 (or v_sad), then per
// output column a difference of prefix sums, the accumulate (v_lshl_add), the key (v_add) and the running minimum.
template <bool SSD>
__global__ void __launch_bounds__(256) mixk(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t pa[14], pb[15];
#pragma unroll
    for (int i = 0; i < 14; ++i) pa[i] = (threadIdx.x + i) * 2654435761u + seed;
#pragma unroll
    for (int i = 0; i < 15; ++i) pb[i] = (threadIdx.x * 3 + i) * 0x9E3779B9u + seed;
    int32_t V[8][2], best[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) { V[x][0] = x; V[x][1] = -x; best[x] = 0x7fffffff; }
    for (int it = 0; it < iters; ++it) {
        uint32_t S0[14], S1[14], s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            s0 = SSD ? __builtin_amdgcn_udot4(pa[i], pb[i + 1], s0, false) : __builtin_amdgcn_sad_u8(pa[i], pb[i + 1], s0);
            s1 = SSD ? __builtin_amdgcn_udot4(pa[i], pb[i], s1, false) : __builtin_amdgcn_sad_u8(pa[i], pb[i], s1);
            S0[i] = s0; S1[i] = s1;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const uint32_t w0 = S0[x + 6] - (x ? S0[x - 1] : 0u), w1 = S1[x + 6] - (x ? S1[x - 1] : 0u);
            V[x][0] = (int32_t)((w0 << 4) + (uint32_t)V[x][0]);
            V[x][1] = (int32_t)((w1 << 4) + (uint32_t)V[x][1]);
            const int32_t k0 = (int32_t)pb[x] + V[x][0], k1 = (int32_t)pb[x + 1] + V[x][1];
            best[x] = min(best[x], min(k0, k1));
        }
        pa[it & 7] += (uint32_t)best[it & 7]; // keep the loop from being hoisted
    }
    uint32_t r = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) r ^= (uint32_t)best[x] ^ (uint32_t)V[x][0] ^ (uint32_t)V[x][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static uint32_t *g_d;

template <typename F>
static float time_ms(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(4000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

template <int OP, int CHAINS>
void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd; // 256-thread blocks = 4 waves = 1 per SIMD per block
    const float ms = time_ms([&](int iters) { hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, g_d, iters, 1u); });
    const double insts_per_simd = 4000.0 * 16 * CHAINS * waves_per_simd * ((OP == 3 || OP == 13 || OP == 6 || OP == 10) ? 2 : 1);
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s chains=%d waves/SIMD=%d : %8.3f ms -> %5.2f cycles(@2.4GHz)/wave-instr/SIMD, %6.2f Tlane-op/s\n", name, CHAINS,
           waves_per_simd, ms, cyc / insts_per_simd, insts_per_simd * 1024 * 64 / (ms * 1e-3) / 1e12);
}

template <bool SSD>
void run_mix(int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    const float ms = time_ms([&](int iters) { hipLaunchKernelGGL((mixk<SSD>), dim3(blocks), dim3(256), 0, 0, g_d, iters, 1u); });
    // per iteration: 28 chain + 15 sub + 16 lshl_add + 16 add + 8 min3 (+1 loop-carried add) = 84 VALU for 16 hypotheses
    const double insts_per_simd = 4000.0 * 84 * waves_per_simd, hyps = 4000.0 * 16 * 64 * waves_per_simd * 1024;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s          waves/SIMD=%d : %8.3f ms -> %5.2f cycles(@2.4GHz)/wave-instr/SIMD, %6.2f Tlane-op/s, %.3g hypotheses/s (5.25 lane-ops each)\n",
           SSD ? "march-mix ssd" : "march-mix sad", waves_per_simd, ms, cyc / insts_per_simd,
           insts_per_simd * 1024 * 64 / (ms * 1e-3) / 1e12, hyps / (ms * 1e-3));
}

int main()
{
    hipMalloc(&g_d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<5, 4>("fma_f32", w);
        run<0, 1>("sad_u8", w); run<0, 2>("sad_u8", w); run<0, 4>("sad_u8", w);
        run<1, 1>("dot4_u8", w); run<1, 2>("dot4_u8", w); run<1, 4>("dot4_u8", w);
        run<8, 4>("dot4_i8", w);
        run<13, 4>("dot4+sub", w);
        run<2, 4>("lshl_add", w);
        run<3, 4>("min+add", w);
        run<4, 4>("sub", w);
        run<6, 4>("xor+min3", w);
        run<7, 4>("add3", w);
        run<9, 4>("pk_add_u16", w);
        run<10, 4>("pk_add+pk_min", w);
        run<11, 4>("perm_b32", w);
        run<12, 4>("mad_u32_u24", w);
        run_mix<true>(w);
        run_mix<false>(w);
    }
    return 0;
}
