// What bandwidth can the kernels' stash pattern reach at all?  2 048 waves (256 workgroups of 8), each with a private
// region of REGION bytes that it writes and then reads back, tile after tile - the traffic of the reverse sweep's
// stash without any arithmetic.  b128 or dword accesses (the kernels use both), regions of 221 KB (7kdv: 453 MB in all,
// more than the 256 MB Infinity Cache) and 64 KB (131 MB: fits).
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/stash_bw tools/ubench/stash_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WIDE>
__global__ __launch_bounds__(512, 1) void k(float *stash, long long region_floats, long long stride_floats, int rounds, float *out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *st = stash + ((long long)blockIdx.x * 8 + wv) * stride_floats;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (WIDE) {
            for (long long i = lane * 4; i < region_floats; i += 256) *(f32x4 *)&st[i] = (f32x4){acc, 1.f, 2.f, (float)r};
            __builtin_amdgcn_s_waitcnt(0);
            for (long long i = lane * 4; i < region_floats; i += 256) { const f32x4 v = *(const volatile f32x4 *)&st[i]; acc += v[0] + v[3]; }
        } else {
            for (long long i = lane; i < region_floats; i += 64) st[i] = acc + (float)r;
            __builtin_amdgcn_s_waitcnt(0);
            for (long long i = lane; i < region_floats; i += 64) acc += *(const volatile float *)&st[i];
        }
    }
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int WIDE> static void run(float *stash, float *out, long long region_bytes, const char *name, long long stride_bytes = 0)
{
    if (!stride_bytes) stride_bytes = region_bytes;
    const int rounds = 30;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<WIDE>, dim3(256), dim3(512), 0, 0, stash, region_bytes / 4, stride_bytes / 4, rounds, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = 2.0 * region_bytes * 2048.0 * rounds;
    printf("%-44s region %4lld KB per wave (%4.0f MB in all): %.2f ms, %.2f TB/s written + read\n", name, region_bytes / 1024,
           region_bytes * 2048.0 / 1e6, best, bytes / best * 1e-9);
}

int main()
{
    float *stash, *out;
    const long long maxb = 221LL * 1024;
    (void)hipMalloc(&stash, maxb * 2048); (void)hipMalloc(&out, 4096);
    (void)hipMemset(stash, 0, maxb * 2048);
    run<1>(stash, out, 221 * 1024, "b128 stores and loads");
    run<0>(stash, out, 221 * 1024, "dword stores and loads");
    run<1>(stash, out, 64 * 1024, "b128 stores and loads");
    run<0>(stash, out, 64 * 1024, "dword stores and loads");
    run<1>(stash, out, 16 * 1024, "b128 stores and loads");
    // what the 7kdv kernel touches: 138 KB of each wave's 221 KB (283 MB in all), and a set just inside the cache
    run<1>(stash, out, 138 * 1024, "b128, 138 KB touched of a 221 KB stride", 221 * 1024);
    run<1>(stash, out, 110 * 1024, "b128, 110 KB touched of a 221 KB stride", 221 * 1024);
    run<1>(stash, out, 96 * 1024, "b128,  96 KB touched of a 221 KB stride", 221 * 1024);
    return 0;
}
