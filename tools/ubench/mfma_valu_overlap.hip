// Microbenchmark: how many independent VALU FMAs hide under one v_mfma_f32_16x16x4_f32 when ONE wave
// per SIMD issues both?  Prints cycles per (1 MFMA + NV FMA) group for NV = 0..12.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
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

template <int NV> void run(float *out, unsigned long long *cyc, int blocks)
{
    const int iters = 2000;
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("NV=%2d blocks=%3d: %.1f cycles per (MFMA + %d FMA)\n", NV, blocks, (double)c / (iters * 8.0), NV);
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    for (int blocks : {1, 256}) {
        run<0>(out, cyc, blocks); run<2>(out, cyc, blocks); run<4>(out, cyc, blocks); run<6>(out, cyc, blocks);
        run<7>(out, cyc, blocks); run<8>(out, cyc, blocks); run<10>(out, cyc, blocks); run<12>(out, cyc, blocks);
    }
    return 0;
}
