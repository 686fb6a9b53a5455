// Microbenchmark 2: do the FP32 MFMA and the FP32 VALU share an execution pipe?
//  (a) two waves per SIMD, one issuing only v_mfma_f32_16x16x4_f32, the other only v_fma_f32:
//      if together they take max(a, b) the pipes are separate and only in-order issue inside one wave
//      prevents overlap; if they take a + b the pipe is shared.
//  (b) same-wave mix of the bf16 MFMA (XDL) with VALU FMAs, for comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// role bits: 1 = waves 0..3 run the MFMA loop, 2 = waves 4..7 run the VALU loop
__global__ __launch_bounds__(512, 1) void k_two(float *out, unsigned long long *cyc, int iters, int roles, int use_pk)
{
    const int wv = threadIdx.x >> 6;
    float s = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wv < 4) {
        if (roles & 1) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float a = threadIdx.x * 0.001f, b = 1.0001f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            }
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        if (roles & 2) {
            if (!use_pk) {
                float v[16];
                for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);   // 64 FMAs = 256 cycles
                }
                for (int i = 0; i < 16; ++i) s += v[i];
            } else {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 v[16];
                for (int i = 0; i < 16; ++i) v[i] = (f32x2){(float)threadIdx.x + i, 1.0f};
                const f32x2 m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);   // 64 pk FMAs
                }
                for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}

template <int NV>
__global__ __launch_bounds__(256, 1) void k_bf16(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x4 a = {(short)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (short)threadIdx.x};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static void two(float *out, unsigned long long *cyc, int roles, int pk)
{
    const int iters = 2000;
    (void)hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL(k_two, dim3(1), dim3(512), 0, 0, out, cyc, iters, roles, pk);
    (void)hipDeviceSynchronize();
    unsigned long long c[8];
    (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("two-wave roles=%d pk=%d: mfma wave %.1f ticks/iter (8 MFMA), valu wave %.1f ticks/iter (64 FMA)\n", roles, pk,
           (double)c[0] / iters, (double)c[4] / iters);
}

template <int NV> void bf(float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL(k_bf16<NV>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("bf16 16x16x16 NV=%2d: %.1f ticks per (MFMA + %d FMA)\n", NV, (double)c / (iters * 8.0), NV);
}

int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&cyc, 64);
    two(out, cyc, 1, 0); two(out, cyc, 2, 0); two(out, cyc, 3, 0);
    two(out, cyc, 2, 1); two(out, cyc, 3, 1);
    bf<0>(out, cyc); bf<1>(out, cyc); bf<2>(out, cyc); bf<3>(out, cyc); bf<4>(out, cyc); bf<6>(out, cyc); bf<8>(out, cyc);
    return 0;
}
