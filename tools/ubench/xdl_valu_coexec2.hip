// Second look at XDL (bf16 MFMA) beside the FP32 VALU, with the loops written in inline assembly (the compiler's own
// version of the same-wave loop shuffled accumulators through AGPRs: xdl_valu_coexec.hip's one-wave numbers are void).
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/xdl_valu_coexec2 tools/ubench/xdl_valu_coexec2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA_BF(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a8), "v"(b8))
#define MFMA_F32(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a1), "v"(b1))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))

// one wave per SIMD (WAVES = 4) or two (WAVES = 8), every wave the same stream: NACC independent accumulators, each MFMA
// followed by NV independent v_fma_f32 (NV distinct registers)
template <int KIND, int NACC, int NV, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_mix(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a1 = threadIdx.x * 0.001f, b1 = 1.0001f, c1 = 1.0001f, c2 = 0.5f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == 0) MFMA_F32(acc[m % NACC]); else if (KIND == 1) MFMA_BF(acc[m % NACC]);
#pragma unroll
            for (int i = 0; i < NV; ++i) FMA(v[(m * NV + i) % 16]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// two waves per SIMD, waves 0..3 MFMAs only (each followed by s_nop NOPS-1 when NOPS > 0), waves 4..7 FMAs only
template <int KIND, int NOPS>
__global__ __launch_bounds__(512, 1) void k_roles(float *out, unsigned long long *cyc, int iters, int roles, int prio)
{
    const int wv = threadIdx.x >> 6;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a1 = threadIdx.x * 0.001f, b1 = 1.0001f, c1 = 1.0001f, c2 = 0.5f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    if (prio && wv >= 4) __builtin_amdgcn_s_setprio(3);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wv < 4) {
        if (roles & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (KIND == 0) MFMA_F32(acc[m]); else MFMA_BF(acc[m]);
                    if (NOPS > 0) asm volatile("s_nop %0" :: "n"(NOPS - 1));
                }
            }
    } else {
        if (roles & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int i = 0; i < 16; ++i) FMA(v[i]);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}

static float *out; static unsigned long long *cyc;
template <int KIND, int NACC, int NV, int WAVES> static void mix()
{
    const int iters = 2000;
    hipLaunchKernelGGL((k_mix<KIND, NACC, NV, WAVES>), dim3(1), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d wave(s)/SIMD, %s, %d accumulators, + %2d FMAs each: %6.1f ticks per (MFMA + FMAs) per wave\n", WAVES / 4,
           KIND == 0 ? "f32 16x16x4  " : KIND == 1 ? "bf16 16x16x32" : "no MFMA      ", NACC, NV, (double)c / (iters * 8.0));
}
template <int KIND, int NOPS> static void roles(int r, int prio)
{
    const int iters = 2000;
    (void)hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL((k_roles<KIND, NOPS>), dim3(1), dim3(512), 0, 0, out, cyc, iters, r, prio);
    (void)hipDeviceSynchronize();
    unsigned long long c[8];
    (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("roles=%d %s nops=%d prio=%d: MFMA wave %6.1f ticks per 8 MFMAs, VALU wave %6.1f ticks per 64 FMAs\n", r,
           KIND ? "bf16 16x16x32" : "f32 16x16x4  ", NOPS, prio, (double)c[0] / iters, (double)c[4] / iters);
}

int main()
{
    (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&cyc, 64);
    // issue rates alone
    mix<0, 8, 0, 4>(); mix<0, 1, 0, 4>(); mix<1, 8, 0, 4>(); mix<1, 4, 0, 4>(); mix<1, 2, 0, 4>(); mix<1, 1, 0, 4>();
    mix<2, 8, 8, 4>(); mix<2, 8, 8, 8>();
    // same wave: MFMA + NV FMAs
    mix<0, 8, 2, 4>(); mix<0, 8, 4, 4>(); mix<0, 8, 8, 4>();
    mix<1, 8, 2, 4>(); mix<1, 8, 4, 4>(); mix<1, 8, 6, 4>(); mix<1, 8, 8, 4>(); mix<1, 8, 12, 4>();
    mix<1, 8, 2, 8>(); mix<1, 8, 4, 8>(); mix<1, 8, 8, 8>(); mix<1, 1, 4, 8>(); mix<1, 2, 4, 8>();
    // two waves with separate roles
    roles<1, 0>(1, 0); roles<1, 0>(2, 0); roles<1, 0>(3, 0); roles<1, 0>(3, 1);
    roles<1, 2>(1, 0); roles<1, 2>(3, 0); roles<1, 4>(1, 0); roles<1, 4>(3, 0); roles<1, 8>(1, 0); roles<1, 8>(3, 0);
    roles<0, 0>(3, 0); roles<0, 8>(1, 0); roles<0, 8>(3, 0);
    return 0;
}
