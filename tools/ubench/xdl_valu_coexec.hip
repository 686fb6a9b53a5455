// Do the bf16 MFMAs (XDL pipe) run BESIDE the FP32 VALU, where the FP32 MFMA does not (mfma_valu_twowave.hip)?
// And is a three-way bf16 split of an FP32 GEMM (six product terms on v_mfma_f32_16x16x32_bf16) as accurate as the
// FP32 MFMA?  Both decide whether the kernels' GEMM phases could move off the FMA lanes.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/xdl_valu_coexec tools/ubench/xdl_valu_coexec.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (1) two waves per SIMD: waves 0..3 only MFMAs (KIND 0: f32 16x16x4, 1: bf16 16x16x32), waves 4..7 only v_fma_f32
template <int KIND>
__global__ __launch_bounds__(512, 1) void k_two(float *out, unsigned long long *cyc, int iters, int roles)
{
    const int wv = threadIdx.x >> 6;
    float s = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wv < 4) {
        if (roles & 1) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float a = threadIdx.x * 0.001f, b = 1.0001f;
            bf16x8 a8, b8;
            for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (KIND == 0) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
                    else acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[m], 0, 0, 0);
                }
            }
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        if (roles & 2) {
            float v[16];
            for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);   // 64 independent-enough FMAs
            }
            for (int i = 0; i < 16; ++i) s += v[i];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}

// (2) one wave per SIMD: every MFMA followed by NV independent v_fma_f32 of the same wave
template <int KIND, int NV>
__global__ __launch_bounds__(256, 1) void k_same(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == 0) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            else acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[m], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// (3) accuracy: D[16][16] = A[16][KK] B[KK][16], KK = 64: FP32 MFMA chain vs six-term bf16 split
//     (round-to-nearest parts: x = h + m + l to 2^-27; terms hh, hm, mh, hl, lh, mm)
__device__ inline void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l)
{
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}
__global__ __launch_bounds__(64) void k_acc(const float *A, const float *B, float *D32, float *Dbf, int KK, int order)
{
    const int l = threadIdx.x, p = l & 15, q = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KK; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[p * KK + k0 + q], B[(k0 + q) * 16 + p], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D32[(4 * q + r) * 16 + p] = acc[r];
    f32x4 ab = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KK; k0 += 32) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int i = 0; i < 8; ++i) {
            __bf16 h, m, lo;
            split3(A[p * KK + k0 + 8 * q + i], h, m, lo); ah[i] = h; am[i] = m; al[i] = lo;
            split3(B[(k0 + 8 * q + i) * 16 + p], h, m, lo); bh[i] = h; bm[i] = m; bl[i] = lo;
        }
        if (order == 0) {   // small terms first
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, ab, 0, 0, 0);
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, ab, 0, 0, 0);
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, ab, 0, 0, 0);
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, ab, 0, 0, 0);
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, ab, 0, 0, 0);
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, ab, 0, 0, 0);
        } else {
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, ab, 0, 0, 0);
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, ab, 0, 0, 0);
            ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, ab, 0, 0, 0);
            if (order == 1) {
                ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, ab, 0, 0, 0);
                ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, ab, 0, 0, 0);
                ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, ab, 0, 0, 0);
            }
        }
    }
    for (int r = 0; r < 4; ++r) Dbf[(4 * q + r) * 16 + p] = ab[r];
}

template <int KIND> static void two(float *out, unsigned long long *cyc, int roles)
{
    const int iters = 2000;
    (void)hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL(k_two<KIND>, dim3(1), dim3(512), 0, 0, out, cyc, iters, roles);
    (void)hipDeviceSynchronize();
    unsigned long long c[8];
    (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("two waves per SIMD, %s, roles=%d: MFMA wave %.1f ticks per 8 MFMAs, VALU wave %.1f ticks per 64 FMAs\n",
           KIND ? "bf16 16x16x32" : "f32 16x16x4", roles, (double)c[0] / iters, (double)c[4] / iters);
}
template <int KIND, int NV> static void same(float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k_same<KIND, NV>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("one wave, %s + %2d FMAs: %.1f ticks per (MFMA + FMAs)\n", KIND ? "bf16 16x16x32" : "f32 16x16x4 ", NV, (double)c / (iters * 8.0));
}

int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&cyc, 64);
    two<0>(out, cyc, 1); two<0>(out, cyc, 2); two<0>(out, cyc, 3);
    two<1>(out, cyc, 1); two<1>(out, cyc, 2); two<1>(out, cyc, 3);
    same<0, 0>(out, cyc); same<0, 4>(out, cyc); same<0, 8>(out, cyc); same<0, 12>(out, cyc);
    same<1, 0>(out, cyc); same<1, 2>(out, cyc); same<1, 4>(out, cyc); same<1, 6>(out, cyc); same<1, 8>(out, cyc); same<1, 12>(out, cyc); same<1, 16>(out, cyc);

    const int KK = 64;
    std::vector<float> A(16 * KK), B(KK * 16);
    float *dA, *dB, *d32, *dbf;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&d32, 1024); (void)hipMalloc(&dbf, 1024);
    for (int order = 0; order < 3; ++order) {
        double e32 = 0, ebf = 0, nrm = 0;
        srand(7);
        for (int trial = 0; trial < 200; ++trial) {
            for (auto &v : A) v = (float)((rand() / (double)RAND_MAX - 0.5) * 4.0 * std::pow(10.0, (rand() % 5) - 2));
            for (auto &v : B) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2.0);
            (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, d32, dbf, KK, order);
            float h32[256], hbf[256];
            (void)hipMemcpy(h32, d32, 1024, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hbf, dbf, 1024, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < KK; ++k) { ref += (double)A[i * KK + k] * B[k * 16 + j]; mag += std::fabs((double)A[i * KK + k] * B[k * 16 + j]); }
                    e32 += std::pow((h32[i * 16 + j] - ref) / mag, 2); ebf += std::pow((hbf[i * 16 + j] - ref) / mag, 2); nrm += 1;
                }
        }
        printf("accuracy K=%d (%s): rms error / sum|a b|: f32 MFMA %.3e, bf16 split %.3e\n", KK,
               order == 0 ? "six terms, small first" : order == 1 ? "six terms, large first" : "three terms", std::sqrt(e32 / nrm), std::sqrt(ebf / nrm));
    }
    return 0;
}
