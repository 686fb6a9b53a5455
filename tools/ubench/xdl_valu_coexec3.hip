// Third look: (a) chip-wide wall-clock rate of back-to-back v_mfma_f32_16x16x32_bf16 at one and two waves per SIMD (is a
// tick of s_memtime a shader cycle, and what does the matrix pipe sustain), (b) a PHASED stream like the kernels' - every
// wave alternates a GEMM phase (NM MFMAs, three accumulator chains) and a jet phase (NF v_fmac_f32, VOP2), the two waves
// of a SIMD half a period apart - with the FP32 MFMA and with the bf16 MFMA.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/xdl_valu_coexec3 tools/ubench/xdl_valu_coexec3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a8), "v"(b8))
#define MFMA_F32(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a1), "v"(b1))
#define FMAC(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))

template <int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_rate(float *out, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a1 = threadIdx.x * 0.001f, b1 = 1.0001f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { if (KIND == 0) MFMA_F32(acc[m]); else MFMA_BF(acc[m]); }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// KIND 0: f32 MFMA x NM; 1: bf16 MFMA x NM; 2: no MFMAs.  Waves 4..7 start with the jet phase.
template <int KIND, int NM, int NF>
__global__ __launch_bounds__(512, 1) void k_phase(float *out, unsigned long long *cyc, int iters)
{
    const int wv = threadIdx.x >> 6;
    f32x4 acc[3];
    for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a1 = threadIdx.x * 0.001f, b1 = 1.0001f, c1 = 1.0001f, c2 = 0.5f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wv >= 4) {
#pragma unroll
        for (int i = 0; i < NF; ++i) FMAC(v[i % 8]);
    }
    for (int it = 0; it < iters; ++it) {
        if (KIND != 2) {
#pragma unroll
            for (int m = 0; m < NM; ++m) { if (KIND == 0) MFMA_F32(acc[m % 3]); else MFMA_BF(acc[m % 3]); }
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) FMAC(v[i % 8]);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}

static float *out; static unsigned long long *cyc;
template <int KIND, int WAVES> static void rate()
{
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<KIND, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    const double flop = (KIND == 0 ? 2048.0 : 16384.0) * 8.0 * iters * WAVES * blocks;
    printf("chip, %d wave(s)/SIMD, back-to-back %s: %.3f ms, %.0f TFLOP/s, %.2f ns per MFMA and SIMD\n", WAVES / 4,
           KIND ? "bf16 16x16x32" : "f32 16x16x4  ", best, flop / best * 1e-9, best * 1e6 / (8.0 * iters * (WAVES / 4)));
}
template <int KIND, int NM, int NF> static void phase()
{
    const int iters = 500;
    (void)hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL((k_phase<KIND, NM, NF>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c[8];
    (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("phased, two waves per SIMD, %s x %3d + %3d v_fmac per period: %7.1f ticks per period and wave (wave 4: %7.1f)\n",
           KIND == 0 ? "f32 16x16x4  " : KIND == 1 ? "bf16 16x16x32" : "no MFMA      ", KIND == 2 ? 0 : NM, NF, (double)c[0] / iters, (double)c[4] / iters);
}

int main()
{
    (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&cyc, 64);
    rate<0, 4>(); rate<0, 8>(); rate<1, 4>(); rate<1, 8>();
    phase<2, 36, 300>(); phase<0, 36, 300>(); phase<1, 36, 300>();
    phase<2, 36, 150>(); phase<0, 36, 150>(); phase<1, 36, 150>();
    phase<1, 36, 0>(); phase<0, 36, 0>();
    return 0;
}
