// Kernel-side descriptor shared by every kernel family.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pinn {

constexpr int MAX_LAYERS = 16;

// passed to kernels by value (kernarg segment, read through scalar loads)
struct KDesc {
    int n_lin;                    // Linear layers
    int widths[MAX_LAYERS + 1];   // layers list
    int woff[MAX_LAYERS];         // offset of W_l in theta (floats)
    int boff[MAX_LAYERS];         // offset of b_l in theta
    int P;                        // parameter count
    int nout;                     // widths[n_lin]
    float lbx, lbt, ubx, ubt;     // Scale bounds
    float coef[8];                // residual coefficients
    float sx, st;                 // 2/(ubx-lbx), 2/(ubt-lbt): d(scaled coordinate)/d(coordinate)
};

enum { MODE_FWD = 0, MODE_VJP = 1, MODE_LOSS = 2 };

// per-call arguments common to all residual kernels
struct CallArgs {
    const float *theta;
    const float *x, *t;
    long long stride;
    long long n;
    const float *gbar;   // MODE_VJP: [n][nf]
    float scale;         // MODE_LOSS: cotangent = 2*scale*f
    float *f_out;        // MODE_FWD: [n][nf] or null
    float *derivs_out;   // MODE_FWD: [n][nout][J] or null
    float *slab;         // MODE_VJP/LOSS: per-workgroup partials [grid][P+2]
    int *tile_ctr;       // MFMA kernels: [2] zeroed counters (tiles handed out, waves finished) of the launch's workspace, or null
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

}  // namespace pinn
