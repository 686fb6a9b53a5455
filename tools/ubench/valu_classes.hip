// Issue cost of the VALU opcode classes the jet arithmetic is made of, two waves per SIMD (the kernels' occupancy), written
// in inline assembly: is a plain multiply / add / convert / bit operation as cheap as an FMA, and does the operand
// pattern (accumulating into its own destination vs. two distinct sources and a third destination) matter?
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/valu_classes tools/ubench/valu_classes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters)
{
    float v[16], u[16];
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x + i; u[i] = i; }
    float c1 = 1.0001f, c2 = 0.5f;
    const unsigned long long mask = __builtin_amdgcn_read_exec() ^ (unsigned long long)iters;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = (i + 1) & 15;
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                if (KIND == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(c1));
                if (KIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[i]) : "v"(c1));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                if (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(u[i]) : "v"(v[i]), "v"(v[n]));
                if (KIND == 5) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[n]));
                if (KIND == 6) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(v[i]), "v"(v[n]), "v"(c2));
                if (KIND == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(v[i]), "v"(v[n]));
                if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(v[i]));
                if (KIND == 9) asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(v[i]));
                if (KIND == 10) asm volatile("v_mul_f32 %0, 2.0, %1" : "=v"(u[i]) : "v"(v[i]));
                if (KIND == 11) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double *)&v[2 * (i & 7)]) : "v"(*(double *)&u[0]), "v"(*(double *)&u[2]));
                if (KIND == 12) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*(double *)&u[2 * (i & 7)]) : "v"(v[i]));
                if (KIND == 13) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[i]) : "v"(v[i]), "v"(v[n]), "s"(mask));
                if (KIND == 14) asm volatile("v_min_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[n]));
                if (KIND == 15) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(c1), "v"(v[i]), "v"(v[n]));
                if (KIND == 16) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(v[i]), "v"(v[n]) : "vcc");
                if (KIND == 17) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(v[i]) : "v"(c1), "s"(mask));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
static float *out; static unsigned long long *cyc;
template <int KIND> void run(const char *name)
{
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(512), 0, 0, out, cyc, 2000);
    (void)hipDeviceSynchronize();
    unsigned long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-58s %.2f cycles per instruction and SIMD (waves 0 / 4: %.2f / %.2f per wave)\n", name, (c[0] > c[4] ? c[0] : c[4]) / (2000.0 * 128),
           c[0] / (2000.0 * 64), c[4] / (2000.0 * 64));
}
int main()
{
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64);
    run<0>("v_fmac_f32 d, c1, c2 (accumulate)");
    run<1>("v_mul_f32 d, c1, d");
    run<2>("v_add_f32 d, c1, d");
    run<3>("v_fma_f32 d, c1, d, c2 (VOP3)");
    run<4>("v_fmac_f32 u, v, v' (distinct registers)");
    run<5>("v_mul_f32 u, v, v' (distinct registers)");
    run<6>("v_fma_f32 u, v, v', c (VOP3, distinct registers)");
    run<7>("v_cndmask_b32 u, v, v', vcc");
    run<8>("v_mov_b32 u, v");
    run<9>("v_add_f32_dpp u, v, v quad_perm");
    run<10>("v_mul_f32 u, 2.0, v");
    run<11>("v_pk_fma_f32 (two FMAs per lane)");
    run<12>("v_cvt_f64_f32");
    run<13>("v_cndmask_b32_e64 u, v, v', s[mask]");
    run<14>("v_min_f32 u, v, v'");
    run<15>("v_bfi_b32 u, c, v, v'");
    run<16>("v_cmp_gt_f32 vcc + v_cndmask_b32 u, v, v', vcc (pair)");
    run<17>("v_cndmask_b32_e64 d, c1, d, s[mask] (own destination)");
    return 0;
}
