// Does the jets' own arithmetic (the real tanh_jet_fwd / tanh_jet_bwd of pde_amd/csrc/jet_math.hpp, order 7) hide a GEMM's
// MFMAs when ONE wave issues both, interleaved - the premise of the next kernel shape?
// Per "neuron slot" a wave runs the forward + reverse tanh jets of one neuron (~330 VALU instructions) and NM MFMAs
// (27 = a 48-wide layer's share per neuron for 9 components) on three accumulators; the MFMAs are spread through
// the jet code by sched_group_barrier.  Variants: jets alone, MFMAs alone, both, for the FP32 and the bf16 form, at one
// and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -w -fno-slp-vectorize -I pde_amd/csrc -o tools/ubench/jet_under_mfma tools/ubench/jet_under_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "jet_math.hpp"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using namespace pinn;

// KIND 0: no MFMAs; 1: f32 16x16x4; 2: bf16 16x16x32.  JETS: run the jet code.  MIX: interleave (else MFMAs after the jets)
template <int KIND, bool JETS, bool MIX, int NM>
__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
    constexpr int K = 7, KT = 1, J = K + 1 + KT, NN = 4;
    float Z[NN][J];
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
        for (int j = 0; j < J; ++j) Z[n][j] = seed * (0.01f * (threadIdx.x & 63) + 0.1f * n + 0.03f * j) - 0.2f;
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a1 = threadIdx.x * 0.001f, b1 = 1.0001f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            if (JETS) {
                float y[J], w[K + 1], zb[J];
                tanh_jet_fwd<K, KT>(Z[n], y, w);
                tanh_jet_bwd<K, KT>(Z[n], y, w, y, zb);
#pragma unroll
                for (int j = 0; j < J; ++j) Z[n][j] = 0.5f * Z[n][j] + 1e-3f * zb[j];
            }
            if (KIND != 0) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if (KIND == 1) acc[m % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[m % 3], 0, 0, 0);
                    else acc[m % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[m % 3], 0, 0, 0);
                }
            }
            if (MIX && JETS && KIND != 0) {
                // one MFMA, then 216 / NM VALU instructions (the jets of one neuron are ~220), NM times
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 216 / NM, 0);
                }
            } else {
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int n = 0; n < NN; ++n) for (int j = 0; j < J; ++j) s += Z[n][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

static float *out; static unsigned long long *cyc;
template <int KIND, bool JETS, bool MIX, int NM> static void run(const char *name, int waves)
{
    const int iters = 300;
    (void)hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL((k<KIND, JETS, MIX, NM>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters, 1.0f);
    (void)hipDeviceSynchronize();
    unsigned long long c[8] = {0}; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int i = 0; i < waves; ++i) if (c[i] > mx) mx = c[i];
    printf("%-64s %d wave(s)/SIMD: %8.1f ticks per neuron slot and wave (%.1f per SIMD)\n", name, waves / 4, mx / (iters * 4.0),
           mx / (iters * 4.0) / (waves / 4));
}

int main()
{
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64);
    for (int waves = 4; waves <= 8; waves += 4) {
        run<0, true, false, 27>("jets alone (tanh_jet_fwd + tanh_jet_bwd, order 7)", waves);
        run<1, false, false, 27>("27 FP32 MFMAs alone", waves);
        run<2, false, false, 27>("27 bf16 MFMAs alone", waves);
        run<1, true, false, 27>("jets, then 27 FP32 MFMAs", waves);
        run<1, true, true, 27>("jets with 27 FP32 MFMAs interleaved", waves);
        run<2, true, false, 27>("jets, then 27 bf16 MFMAs", waves);
        run<2, true, true, 27>("jets with 27 bf16 MFMAs interleaved", waves);
        run<2, true, true, 54>("jets with 54 bf16 MFMAs interleaved (two GEMM directions)", waves);
        run<2, true, true, 81>("jets with 81 bf16 MFMAs interleaved (all three GEMMs)", waves);
    }
    return 0;
}
