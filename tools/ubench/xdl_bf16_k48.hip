// K = 48 as one v_mfma_f32_16x16x32_bf16 plus one v_mfma_f32_16x16x16_bf16 per product term: operand layouts, accuracy of the
// six-term three-way split against the FP32 MFMA, and the issue rate of the x16 form (two waves per SIMD, chip-wide wall clock).
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/xdl_bf16_k48 tools/ubench/xdl_bf16_k48.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ inline void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l)
{
    h = (__bf16)x; const float r1 = x - (float)h; m = (__bf16)r1; const float r2 = r1 - (float)m; l = (__bf16)r2;
}
__device__ inline f32x4 mfma16(bf16x4 a, bf16x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__global__ __launch_bounds__(64) void k_acc(const float *A, const float *B, float *D32, float *Dbf)
{
    constexpr int KK = 48;
    const int l = threadIdx.x, p = l & 15, q = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KK; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[p * KK + k0 + q], B[(k0 + q) * 16 + p], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D32[(4 * q + r) * 16 + p] = acc[r];
    bf16x8 ah, am, al, bh, bm, bl;
    bf16x4 ch, cm, cl, dh, dm, dl;
    for (int i = 0; i < 8; ++i) {
        __bf16 h, m, lo;
        split3(A[p * KK + 8 * q + i], h, m, lo); ah[i] = h; am[i] = m; al[i] = lo;
        split3(B[(8 * q + i) * 16 + p], h, m, lo); bh[i] = h; bm[i] = m; bl[i] = lo;
    }
    for (int i = 0; i < 4; ++i) {
        __bf16 h, m, lo;
        split3(A[p * KK + 32 + 4 * q + i], h, m, lo); ch[i] = h; cm[i] = m; cl[i] = lo;
        split3(B[(32 + 4 * q + i) * 16 + p], h, m, lo); dh[i] = h; dm[i] = m; dl[i] = lo;
    }
    f32x4 ab = {0.f, 0.f, 0.f, 0.f};
    // small terms first: (m,m) (l,h) (h,l) (m,h) (h,m) (h,h)
#define TERM(x8, y8, x4, y4) do { ab = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, ab, 0, 0, 0); ab = mfma16(x4, y4, ab); } while (0)
    TERM(am, bm, cm, dm); TERM(al, bh, cl, dh); TERM(ah, bl, ch, dl); TERM(am, bh, cm, dh); TERM(ah, bm, ch, dm); TERM(ah, bh, ch, dh);
    for (int r = 0; r < 4; ++r) Dbf[(4 * q + r) * 16 + p] = ab[r];
}

template <int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_rate(float *out, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a8, b8; bf16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a8), "v"(b8));
            else asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a4), "v"(b4));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int KIND, int WAVES> static void rate(float *out)
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
    printf("chip, %d wave(s)/SIMD, back-to-back %s: %.3f ms, %.2f ns per MFMA and SIMD\n", WAVES / 4, KIND ? "bf16 16x16x16" : "bf16 16x16x32",
           best, best * 1e6 / (8.0 * iters * (WAVES / 4)));
}

int main()
{
    const int KK = 48;
    std::vector<float> A(16 * KK), B(KK * 16);
    float *dA, *dB, *d32, *dbf, *out;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&d32, 1024); (void)hipMalloc(&dbf, 1024);
    (void)hipMalloc(&out, 4096);
    double e32 = 0, ebf = 0, nrm = 0;
    srand(7);
    for (int trial = 0; trial < 200; ++trial) {
        for (auto &v : A) v = (float)((rand() / (double)RAND_MAX - 0.5) * 4.0 * std::pow(10.0, (rand() % 5) - 2));
        for (auto &v : B) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2.0);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, d32, dbf);
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
    printf("accuracy K=48 (x32 + x16, six terms, small first): rms error / sum|a b|: f32 MFMA %.3e, bf16 split %.3e\n", std::sqrt(e32 / nrm), std::sqrt(ebf / nrm));
    rate<0, 8>(out); rate<1, 8>(out); rate<1, 4>(out);
    return 0;
}
