// Fused Adam / AdamW step over the flat parameter vector.
//
// The reference's Adam phase is torch.optim.Adam / AdamW(weight_decay=1e-2) on the ten parameter tensors of the
// net (7kdv/PI-7kdv3soliton.py:138,154-157; 7kdv/sw73-adamW.py:170).  torch runs it as ~15 foreach launches per
// step; here the parameters are one flat buffer (pde_amd/nn.py), so one launch does the whole update.  Same
// arithmetic as torch's _single_tensor_adam (default flags): exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2 v + (1-b2) g g;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= lr/(1-b1^t) * m/denom; weight decay added to g (Adam) or applied to p
// first (AdamW).  The step count and the running powers b1^t, b2^t live on the device in float64, so the launch
// takes no host scalar that changes from step to step and can be replayed from a captured HIP graph.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pinnjet.h"

namespace {

// state: [0] step t, [1] b1^t, [2] b2^t (float64), then one uint32 ticket counter at byte 24
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                               float *__restrict__ v, double *state, long long n, float lr, double b1d,
                                               double b2d, float eps, float wd, int decoupled)
{
    // every thread derives this step's scalars from the state the previous launch left
    // (torch forms 1 - beta in double and rounds once: 1 - 0.999 as a float product of float(0.999) would be 1e-5 off)
    const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    const double t = state[0] + 1.0;
    const double p1 = (state[0] == 0.0 ? 1.0 : state[1]) * b1d;
    const double p2 = (state[0] == 0.0 ? 1.0 : state[2]) * b2d;
    const float step_size = (float)((double)lr / (1.0 - p1));
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - p2));
    const float decay = 1.0f - lr * wd;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float pi = p[i], gi = g[i];
        if (wd != 0.0f) {
            if (decoupled) pi *= decay;
            else gi = fmaf(wd, pi, gi);
        }
        float mi = m[i], vi = v[i];
        mi = fmaf(gi - mi, omb1, mi);
        vi = fmaf(omb2 * gi, gi, vi * b2);
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        pi = fmaf(-step_size, mi / denom, pi);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
    // the workgroup that finishes last publishes the new state: every other workgroup has read the old one by then
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int *ticket = (unsigned int *)(state + 3);
        __threadfence();
        const unsigned int done = atomicAdd(ticket, 1u);
        if (done == gridDim.x - 1) {
            state[0] = t; state[1] = p1; state[2] = p2;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        (void)t;
    }
}

}  // namespace

extern "C" int pinn_adam_step(float *theta, const float *grad, float *exp_avg, float *exp_avg_sq, double *state,
                              int64_t n, float lr, double beta1, double beta2, float eps, float weight_decay,
                              int32_t decoupled, void *stream)
{
    if (!theta || !grad || !exp_avg || !exp_avg_sq || !state) return PINN_ERR_NULL;
    if (n < 0 || n > ((int64_t)1 << 31)) return PINN_ERR_INVALID_DESC;
    if (n == 0) return PINN_OK;
    const long long nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, theta, grad, exp_avg, exp_avg_sq,
                       state, (long long)n, lr, beta1, beta2, eps, weight_decay, (int)decoupled);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}
