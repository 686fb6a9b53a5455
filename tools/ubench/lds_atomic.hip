// Microbenchmark 3: cost of a 64-lane ds_add_f32 for different address patterns (1 and 4 waves per CU),
// against a plain read-add-write of a wave-private image.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// pattern 0: addr = lane (conflict free);  1: addr = (4q)*50 + p (the row-major width-50 weight-gradient
// layout);  2: all waves the same addresses as 0 (contention between waves);  3: private b128 read-add-write
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int iters, int pattern, int nwaves)
{
    __shared__ float lds[16384];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.f;
    __syncthreads();
    if (wv >= nwaves) return;
    float v = lane * 0.5f;
    int base;
    if (pattern == 0) base = wv * 2560 + lane;
    else if (pattern == 1) base = wv * 2560 + 4 * q * 50 + p;
    else if (pattern == 2) base = lane;
    else base = wv * 2560 + lane * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (pattern < 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 9; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&lds[base + (m * 4 + r) * 64], v);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                f32x4 c = *(f32x4 *)&lds[base + m * 256];
                c += v;
                *(f32x4 *)&lds[base + m * 256] = c;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = lds[base];
    if (lane == 0) cyc[wv] = t1 - t0;
}

int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&cyc, 64);
    const int iters = 500;
    for (int nw : {1, 4})
        for (int pat = 0; pat < 4; ++pat) {
            (void)hipMemset(cyc, 0, 64);
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, cyc, iters, pat, nw);
            (void)hipDeviceSynchronize();
            unsigned long long c[8];
            (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
            printf("waves=%d pattern=%d: %.1f ticks per 64-lane %s\n", nw, pat, (double)c[0] / (iters * (pat < 3 ? 36.0 : 9.0)),
                   pat < 3 ? "ds_add_f32" : "b128 read+add+write");
        }
    return 0;
}
