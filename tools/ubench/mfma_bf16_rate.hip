// Issue rate of the bf16 MFMAs a three-way bf16 split of the FP32 GEMMs would use, against v_mfma_f32_16x16x4_f32.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/mfma_bf16_rate tools/ubench/mfma_bf16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void k(int n, float *out)
{
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
            if (KIND == 1) acc[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[u], 0, 0, 0);
            if (KIND == 2) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[u], 0, 0, 0);
        }
    }
    float r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int KIND> float run(int n, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, n, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    return best;
}

int main()
{
    float *out; (void)hipMalloc(&out, 4096);
    const int n = 50000;
    const float t0 = run<0>(n, out), t1 = run<1>(n, out), t2 = run<2>(n, out);
    printf("one wave per SIMD, 4 accumulators, %d x 4 MFMAs: f32 16x16x4 %.3f ms | bf16 16x16x16 %.3f ms (%.2fx) | bf16 16x16x32 %.3f ms (%.2fx)\n",
           n, t0, t1, t1 / t0, t2, t2 / t0);
    printf("per K: f32 %.3f, bf16 x16 %.3f, bf16 x32 %.3f (us per 1e3 MFMA-K-units)\n", t0 / 4, t1 / 16, t2 / 32);
    return 0;
}
