// The data terms of the reference's backward() on the outputs of one net_u batch, one launch each way.
//
// The reference forms loss0 = mean((u0_pred - u0)^2) [+ the same for v], loss_b = mean((u_lb_pred - u_ub_pred)^2)
// [+ v] (7kdv/PI-7kdv3soliton.py:99-105; two outputs: NLS/sw-GNSS-adamW.py:114-129) or, for Burgers,
// mean(u_lb_pred^2) + mean(u_ub_pred^2) (Burgers/Burgers_equation(PINN).py:109) as a chain of ~12 small torch
// kernels forward and as many backward.  At the scripts' sizes (150 data points) that chain, not the arithmetic,
// is what a closure waits for.  Here the batch is X = [IC rows (n0) | lower-boundary rows (nb) | upper-boundary
// rows (nb)], `out` = net_u(X) as pinn_forward wrote it ([rows][nout]), and
//   k_data_loss       : losses[0] = sum_{rows < n0, cols} (out - tgt0)^2 / n0
//                       losses[1] = sum_{i < nb, cols} (out[n0+i] - out[n0+nb+i])^2 / nb      (periodic)
//                                   sum_{n0 <= rows, cols} out^2 / nb                        (dirichlet)
//   k_data_cotangent  : gbar = g0 * d losses[0]/d out + gb * d losses[1]/d out   (g0, gb: device scalars),
// which pinn_forward_vjp turns into the weight gradient.  Sums in float64, one workgroup, fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pinnjet.h"

namespace {

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

constexpr int LOSS_THREADS = 1024;      // one workgroup (fixed summation order); the bc classes' loss1 batch is ~10^5 rows

__global__ __launch_bounds__(LOSS_THREADS) void k_data_loss(const float *__restrict__ out, const float *__restrict__ tgt0, int n0, int nb,
                                                    int nout, int dirichlet, float *__restrict__ losses)
{
    __shared__ double part[2][LOSS_THREADS / 64];
    double s0 = 0.0, sb = 0.0;
    const int e0 = n0 * nout;
    // eight elements per thread and pass, all loads issued before the first use (the bc classes' loss1 batch is 77 312
    // rows: as a plain loop each of its 75 passes waited for its own two loads); same summation order as the plain loop
    constexpr int U = 8;
    for (int base = threadIdx.x; base < e0; base += LOSS_THREADS * U) {
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + LOSS_THREADS * u, ii = i < e0 ? i : e0 - 1;
            a[u] = out[ii];
            b[u] = tgt0[ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + LOSS_THREADS * u < e0) {
                const float r = a[u] - b[u];
                s0 += (double)r * r;
            }
    }
    const int eb = (dirichlet ? 2 * nb : nb) * nout;
    for (int base = threadIdx.x; base < eb; base += LOSS_THREADS * U) {
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + LOSS_THREADS * u, ii = i < eb ? i : eb - 1;
            a[u] = out[e0 + ii];
            b[u] = dirichlet ? 0.0f : out[e0 + nb * nout + ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + LOSS_THREADS * u < eb) {
                const float r = a[u] - b[u];
                sb += (double)r * r;
            }
    }
    s0 = wave_sum_d(s0);
    sb = wave_sum_d(sb);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wv] = s0; part[1][wv] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < LOSS_THREADS / 64; ++w) { a += part[0][w]; b += part[1][w]; }
        losses[0] = n0 > 0 ? (float)(a / n0) : 0.0f;
        losses[1] = nb > 0 ? (float)(b / nb) : 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_data_cotangent(const float *__restrict__ out, const float *__restrict__ tgt0, int n0,
                                                         int nb, int nout, int dirichlet, const float *__restrict__ g0,
                                                         const float *__restrict__ gb, float *__restrict__ gbar)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int e0 = n0 * nout, eb = nb * nout;
    if (i >= e0 + 2 * eb) return;
    if (i < e0) {
        gbar[i] = (out[i] - tgt0[i]) * (g0[0] * (2.0f / (float)n0));
        return;
    }
    const float c = gb[0] * (2.0f / (float)nb);
    if (dirichlet) {
        gbar[i] = out[i] * c;
    } else {
        const int j = i - e0;                                 // lower rows first, then upper rows
        const int k = j < eb ? j : j - eb;
        const float t = (out[e0 + k] - out[e0 + eb + k]) * c;
        gbar[i] = j < eb ? t : -t;
    }
}

// The tail of backward(): `loss = loss0 + loss_b [+ loss1] [+ loss_prev] [+ loss_s] + loss_f; loss.backward()`
// (7kdv/PI-7kdv3soliton.py:110-111, 7kdv/bc-7kdv3soliton.py:252-253, 7kdv/sw-73.py:142-143) once every term's value and weight gradient exist: the float32 sums in the
// order the reference adds them, and the two flat gradients (data terms, collocation term) added into `.grad`.
__global__ __launch_bounds__(256) void k_closure_sum(const float *__restrict__ terms, int n_terms, float *__restrict__ sums,
                                                      int nf, float n_total, float *__restrict__ loss2,
                                                      const float *__restrict__ ga, float *__restrict__ gb,
                                                      float *__restrict__ grad, long long P, int accumulate, int clear)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < P) {
        float v = ga[i];
        if (gb) {
            v += gb[i];
            if (clear) gb[i] = 0.0f;
        }
        grad[i] = accumulate ? grad[i] + v : v;
    }
    if (i == 0) {
        float lf = 0.0f;
        if (nf > 0) {
            lf = sums[0];
            for (int c = 1; c < nf; ++c) lf += sums[c];
            lf = lf / n_total;
            if (clear)
                for (int c = 0; c < nf; ++c) sums[c] = 0.0f;
        }
        // the reference's order: loss0 + loss_b + loss_f + (loss1 | loss_prev | loss_s)
        float l = n_terms > 0 ? terms[0] : 0.0f;
        if (n_terms > 1) l += terms[1];
        l += lf;
        for (int k = 2; k < n_terms; ++k) l += terms[k];
        loss2[0] = l;
        loss2[1] = lf;
    }
}

}  // namespace

extern "C" {

int pinn_closure_sum(const float *terms, int32_t n_terms, float *sums, int32_t nf, float n_total, float *loss2,
                     const float *ga, float *gb, float *grad, int64_t P, int32_t flags, void *stream)
{
    const int accumulate = flags & 1, clear = (flags >> 1) & 1;
    if (!loss2 || !ga || !grad || (n_terms > 0 && !terms) || (nf > 0 && !sums)) return PINN_ERR_NULL;
    if (n_terms < 0 || n_terms > 16 || nf < 0 || nf > 3 || P <= 0 || !(n_total > 0.0f) || (flags & ~3)) return PINN_ERR_INVALID_DESC;
    if (P > (1LL << 30)) return PINN_ERR_TOO_LARGE;
    hipLaunchKernelGGL(k_closure_sum, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, terms, n_terms, sums,
                       nf, n_total, loss2, ga, gb, grad, (long long)P, accumulate, clear);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) fprintf(stderr, "pinn_closure_sum: %s\n", hipGetErrorString(e));
    return e == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

int pinn_data_loss(const float *out, const float *tgt0, int32_t n0, int32_t nb, int32_t nout, int32_t dirichlet,
                   float *losses, void *stream)
{
    if (!out || !losses || (n0 > 0 && !tgt0)) return PINN_ERR_NULL;
    if (n0 < 0 || nb < 0 || nout < 1 || nout > PINN_MAX_OUT) return PINN_ERR_INVALID_DESC;
    hipLaunchKernelGGL(k_data_loss, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, out, tgt0, n0, nb, nout, dirichlet, losses);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

int pinn_data_loss_cotangent(const float *out, const float *tgt0, int32_t n0, int32_t nb, int32_t nout, int32_t dirichlet,
                             const float *g0, const float *gb, float *gbar, void *stream)
{
    if (!out || !gbar || !g0 || !gb || (n0 > 0 && !tgt0)) return PINN_ERR_NULL;
    if (n0 < 0 || nb < 0 || nout < 1 || nout > PINN_MAX_OUT) return PINN_ERR_INVALID_DESC;
    const long long total = ((long long)n0 + 2LL * nb) * nout;
    if (total == 0) return PINN_OK;
    if (total > (1LL << 30)) return PINN_ERR_TOO_LARGE;
    hipLaunchKernelGGL(k_data_cotangent, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, tgt0, n0,
                       nb, nout, dirichlet, g0, gb, gbar);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

}  // extern "C"
