// Microbenchmark 5: VALU issue rate for NC independent dependency chains of v_fma_f32 / v_pk_fma_f32 /
// v_mul+v_add mixes, 1 or 2 waves per SIMD (ticks per instruction per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NC, int PK>
__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    unsigned long long t0, t1;
    float s = 0;
    if (!PK) {
        float v[NC];
        for (int i = 0; i < NC; ++i) v[i] = threadIdx.x + i;
        const float m = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 48 / NC; ++rep)
#pragma unroll
                for (int i = 0; i < NC; ++i) v[i] = fmaf(v[i], m, c);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < NC; ++i) s += v[i];
    } else {
        f32x2 v[NC];
        for (int i = 0; i < NC; ++i) v[i] = (f32x2){(float)threadIdx.x + i, 1.0f};
        const f32x2 m = {1.0001f + threadIdx.x * 1e-9f, 1.0002f}, c = {0.5f, 0.25f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 48 / NC; ++rep)
#pragma unroll
                for (int i = 0; i < NC; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < NC; ++i) s += v[i][0] + v[i][1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NC, int PK> void run(float *out, unsigned long long *cyc, int threads)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NC, PK>), dim3(1024), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NC, PK>), dim3(1024), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 1024.0 / 256.0 * (threads / 256) * iters * 48.0;   // 4 workgroups per CU in sequence
    hipLaunchKernelGGL((k<NC, PK>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c[8];
    (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%s chains=%2d waves/SIMD=%d: %.2f ticks/instr (wave 0, single WG); chip %.3f ms = %.2f ns per instr per SIMD, %.1f TFLOP/s\n",
           PK ? "pk_fma" : "fma   ", NC, threads / 256, (double)c[0] / (iters * 48.0), ms, ms * 1e6 / instr_per_simd,
           1024.0 * threads * iters * 48.0 * (PK ? 4 : 2) / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&cyc, 64);
    for (int threads : {256, 512}) {
        run<1, 0>(out, cyc, threads); run<2, 0>(out, cyc, threads); run<4, 0>(out, cyc, threads); run<8, 0>(out, cyc, threads);
        run<16, 0>(out, cyc, threads);
        run<1, 1>(out, cyc, threads); run<2, 1>(out, cyc, threads); run<4, 1>(out, cyc, threads); run<8, 1>(out, cyc, threads);
    }
    return 0;
}
