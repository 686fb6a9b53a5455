// L-BFGS search direction on the device.
//
// The reference finishes every training run with one torch.optim.LBFGS.step(closure)
// (7kdv/PI-7kdv3soliton.py:138-161) of up to 50 000 closure evaluations.  torch's
// implementation runs the two-loop recursion as ~6 tiny tensor ops per history pair
// (300-900 launches per iteration at history 50-150) plus a host sync for every
// scalar it branches on; at the scripts' native sizes (N_f = 20 000) that costs several
// times more than the closure itself.  Here one launch does, in the same order of
// operations as the textbook recursion:
//     y = g - g_prev, s = t*d, curvature test y.s > 1e-10, history ring update,
//     H = y.s / y.y, q = -g, first loop (newest -> oldest), r = H q,
//     second loop (oldest -> newest), d = r, g_prev = g,
// and leaves the scalars the host's line search needs (g.d, sum|g|, max|d|, max|g|)
// in a small device buffer.  P is 7 851 .. 30 802 floats, so one 1024-thread
// workgroup streams the history out of L2; every reduction is a fixed-order tree,
// which keeps the result deterministic and identical on every rank.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pinnjet.h"

namespace {

constexpr int LB_THREADS = 1024;
constexpr int LB_WAVES = LB_THREADS / 64;
constexpr int LB_HDR = 16;          // floats: [0] pairs held, [1] ring head, [2] H_diag
constexpr int LB_MAX_HISTORY = 512;

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Block-wide reduction of NV sums and NM maxima; every thread leaves with the same values
// (all of them add the per-wave partials in the same order).
template <int NV, int NM>
__device__ inline void block_reduce(float (&s)[NV], float (&m)[NM > 0 ? NM : 1], float *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) s[k] = wave_sum(s[k]);
#pragma unroll
    for (int k = 0; k < NM; ++k) m[k] = wave_max(m[k]);
    __syncthreads();    // the previous reduction's readers are done with `red`
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[w * (NV + NM) + k] = s[k];
#pragma unroll
        for (int k = 0; k < NM; ++k) red[w * (NV + NM) + NV + k] = m[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float a = 0.f;
        for (int i = 0; i < LB_WAVES; ++i) a += red[i * (NV + NM) + k];
        s[k] = a;
    }
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        float a = red[NV + k];
        for (int i = 1; i < LB_WAVES; ++i) a = fmaxf(a, red[i * (NV + NM) + NV + k]);
        m[k] = a;
    }
}

struct LbLayout {
    int64_t ro, prev_g, q, S, Y, total;
};
inline __host__ __device__ LbLayout lb_layout(int64_t n, int hist) {
    LbLayout L;
    const int64_t np = (n + 3) & ~int64_t(3);
    L.ro = LB_HDR;
    L.prev_g = L.ro + ((hist + 3) & ~3);
    L.q = L.prev_g + np;
    L.S = L.q + np;
    L.Y = L.S + int64_t(hist) * np;
    L.total = L.Y + int64_t(hist) * np;
    return L;
}

__global__ __launch_bounds__(LB_THREADS) void k_lbfgs_direction(float *__restrict__ ws, int n, int hist,
                                                                const float *__restrict__ g, float *__restrict__ d,
                                                                float t, int first, float *__restrict__ stats) {
    __shared__ float red[LB_WAVES * 4];
    __shared__ float al_s[LB_MAX_HISTORY];
    const LbLayout L = lb_layout(n, hist);
    const int np = (n + 3) & ~3;
    float *ro = ws + L.ro, *prev_g = ws + L.prev_g, *q = ws + L.q, *S = ws + L.S, *Y = ws + L.Y;
    const int tid = threadIdx.x;
    float dummy[1] = {0.f};

    int count = first ? 0 : (int)ws[0];
    int head = first ? 0 : (int)ws[1];
    float Hd = first ? 1.f : ws[2];

    if (!first) {
        // curvature pair of the step just taken
        float a[2] = {0.f, 0.f};
        for (int i = tid; i < n; i += LB_THREADS) {
            const float y = g[i] - prev_g[i], s = d[i] * t;
            a[0] = fmaf(y, s, a[0]);
            a[1] = fmaf(y, y, a[1]);
        }
        block_reduce<2, 0>(a, dummy, red);
        const float ys = a[0];
        if (ys > 1e-10f) {      // same value on every thread
            int slot;
            if (count == hist) {
                slot = head;                     // overwrite the oldest pair
                head = (head + 1) % hist;
            } else {
                slot = (head + count) % hist;
                ++count;
            }
            float *Ys = Y + int64_t(slot) * np, *Ss = S + int64_t(slot) * np;
            for (int i = tid; i < n; i += LB_THREADS) {
                Ys[i] = g[i] - prev_g[i];
                Ss[i] = d[i] * t;
            }
            if (tid == 0) ro[slot] = 1.f / ys;
            Hd = ys / a[1];
            __syncthreads();                     // ro[slot] visible to the whole workgroup
        }
    }

    // q = -g, fused with the first dot product of the first loop
    float dot[1] = {0.f};
    {
        const float *Sn = count > 0 ? S + int64_t((head + count - 1) % hist) * np : nullptr;
        for (int i = tid; i < n; i += LB_THREADS) {
            const float qi = -g[i];
            q[i] = qi;
            if (Sn) dot[0] = fmaf(Sn[i], qi, dot[0]);
        }
        if (count > 0) block_reduce<1, 0>(dot, dummy, red);
    }
    // first loop, newest -> oldest: al_k = ro_k s_k.q ; q -= al_k y_k   (the pass that updates q also
    // accumulates the next dot product: s_{k-1}.q, or y_0.(H q) when the loop ends)
    for (int k = count - 1; k >= 0; --k) {
        const int slot = (head + k) % hist;
        const float al = dot[0] * ro[slot];
        if (tid == 0) al_s[k] = al;
        const float *Yk = Y + int64_t(slot) * np;
        const float *nxt = k > 0 ? S + int64_t((head + k - 1) % hist) * np : Y + int64_t(head % hist) * np;
        const float scale = k > 0 ? 1.f : Hd;
        float nd[1] = {0.f};
        for (int i = tid; i < n; i += LB_THREADS) {
            const float qi = fmaf(-al, Yk[i], q[i]);
            q[i] = qi;
            nd[0] = fmaf(nxt[i], qi * scale, nd[0]);
        }
        block_reduce<1, 0>(nd, dummy, red);
        dot[0] = nd[0];
    }
    __syncthreads();    // al_s complete
    // r = H q ; second loop, oldest -> newest: be_k = ro_k y_k.r ; r += (al_k - be_k) s_k
    for (int i = tid; i < n; i += LB_THREADS) q[i] *= Hd;
    for (int k = 0; k < count; ++k) {
        const int slot = (head + k) % hist;
        const float coef = al_s[k] - dot[0] * ro[slot];
        const float *Sk = S + int64_t(slot) * np;
        const float *nxt = k + 1 < count ? Y + int64_t((head + k + 1) % hist) * np : nullptr;
        float nd[1] = {0.f};
        for (int i = tid; i < n; i += LB_THREADS) {
            const float ri = fmaf(coef, Sk[i], q[i]);
            q[i] = ri;
            if (nxt) nd[0] = fmaf(nxt[i], ri, nd[0]);
        }
        if (nxt) {
            block_reduce<1, 0>(nd, dummy, red);
            dot[0] = nd[0];
        }
    }
    // d = r, g_prev = g, and the scalars the host needs
    float s2[2] = {0.f, 0.f}, m2[2] = {0.f, 0.f};
    for (int i = tid; i < n; i += LB_THREADS) {
        const float di = q[i], gi = g[i];
        d[i] = di;
        prev_g[i] = gi;
        s2[0] = fmaf(gi, di, s2[0]);
        s2[1] += fabsf(gi);
        m2[0] = fmaxf(m2[0], fabsf(di));
        m2[1] = fmaxf(m2[1], fabsf(gi));
    }
    block_reduce<2, 2>(s2, m2, red);
    if (tid == 0) {
        ws[0] = (float)count;
        ws[1] = (float)head;
        ws[2] = Hd;
        stats[0] = s2[0];         // g.d
        stats[1] = s2[1];         // sum |g|
        stats[2] = m2[0];         // max |d|
        stats[3] = m2[1];         // max |g|
        stats[4] = (float)count;
        stats[5] = Hd;
        stats[6] = 0.f;
        stats[7] = 0.f;
    }
}

// out = [loss, g.d, max|g|, sum|g|] after one closure evaluation of the line search
__global__ __launch_bounds__(LB_THREADS) void k_lbfgs_eval_stats(const float *__restrict__ g, const float *__restrict__ d,
                                                                 int n, const float *__restrict__ loss,
                                                                 float *__restrict__ out) {
    __shared__ float red[LB_WAVES * 3];
    float s[2] = {0.f, 0.f}, m[1] = {0.f};
    for (int i = threadIdx.x; i < n; i += LB_THREADS) {
        const float gi = g[i];
        s[0] = fmaf(gi, d[i], s[0]);
        s[1] += fabsf(gi);
        m[0] = fmaxf(m[0], fabsf(gi));
    }
    block_reduce<2, 1>(s, m, red);
    if (threadIdx.x == 0) {
        out[0] = loss ? loss[0] : 0.f;
        out[1] = s[0];
        out[2] = m[0];
        out[3] = s[1];
    }
}

}  // namespace

extern "C" {

int64_t pinn_lbfgs_workspace_floats(int64_t n, int32_t history) {
    if (n <= 0 || history <= 0 || history > LB_MAX_HISTORY || n > (int64_t(1) << 24)) return PINN_ERR_INVALID_DESC;
    return lb_layout(n, history).total;
}

int pinn_lbfgs_direction(float *workspace, int64_t n, int32_t history, const float *g, float *d, float t,
                         int32_t first, float *stats8, void *stream) {
    if (!workspace || !g || !d || !stats8) return PINN_ERR_NULL;
    if (n <= 0 || history <= 0 || history > LB_MAX_HISTORY || n > (int64_t(1) << 24)) return PINN_ERR_INVALID_DESC;
    hipLaunchKernelGGL(k_lbfgs_direction, dim3(1), dim3(LB_THREADS), 0, (hipStream_t)stream, workspace, (int)n,
                       (int)history, g, d, t, (int)first, stats8);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

int pinn_lbfgs_eval_stats(const float *g, const float *d, int64_t n, const float *loss, float *out4, void *stream) {
    if (!g || !d || !out4) return PINN_ERR_NULL;
    if (n <= 0 || n > (int64_t(1) << 24)) return PINN_ERR_INVALID_DESC;
    hipLaunchKernelGGL(k_lbfgs_eval_stats, dim3(1), dim3(LB_THREADS), 0, (hipStream_t)stream, g, d, (int)n, loss,
                       out4);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

}  // extern "C"
