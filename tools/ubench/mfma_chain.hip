// Microbenchmark 4: issue rate of v_mfma_f32_16x16x4_f32 with NA independent accumulators (dependent
// chains through the C operand), one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NA>
__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[NA];
    for (int i = 0; i < NA; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24 / NA; ++rep)
#pragma unroll
            for (int m = 0; m < NA; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NA> void run(float *out, unsigned long long *cyc, int threads)
{
    const int iters = 1000;
    {   // whole-chip wall-clock rate: 1024 workgroups
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<NA>, dim3(1024), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<NA>, dim3(1024), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = 1024.0 * (threads / 64) * iters * 24.0 * 2048.0;
        printf("  chip: %.3f ms, %.1f TFLOP/s  ", ms, flops / (ms * 1e-3) / 1e12);
    }
    hipLaunchKernelGGL(k<NA>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c[8];
    (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("NA=%d waves/SIMD=%d: %.1f ticks per MFMA per wave (%.1f per SIMD)\n", NA, threads / 256, (double)c[0] / (iters * 24.0),
           (double)c[0] / (iters * 24.0) / (threads / 256));
}

int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&cyc, 64);
    for (int threads : {256, 512}) {
        run<1>(out, cyc, threads); run<2>(out, cyc, threads); run<3>(out, cyc, threads);
        run<4>(out, cyc, threads); run<6>(out, cyc, threads); run<8>(out, cyc, threads);
    }
    return 0;
}
