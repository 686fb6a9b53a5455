// L-BFGS search direction on the device.
//
// The reference finishes every training run with one torch.optim.LBFGS.step(closure)
// (7kdv/PI-7kdv3soliton.py:138-161) of up to 50 000 closure evaluations.  torch's
// implementation runs the two-loop recursion as ~6 tiny tensor ops per history pair
// (300-900 launches per iteration at history 50-150) plus a host sync for every
// scalar it branches on; at the scripts' native sizes (N_f = 20 000) that costs several
// times more than the closure itself.
//
// The two-loop recursion is sequential only in its SCALARS: every inner product it takes
// (s_k.q, y_k.r with q, r partially updated) is a linear combination of inner products
// among {g, s_j, y_j}.  So one direction update is arranged as four chip-wide launches
//   k_dots     every live history row (and the candidate pair y = g - g_prev, s = t d,
//              formed on the fly) against g and the candidate: 5 dot products per row,
//              float products accumulated in float64, per-chunk partials (no atomics);
//   k_scalars  one workgroup: curvature test y.s > 1e-10, history bookkeeping, the new
//              row/column of the Gram matrices S.Y^T and Y.Y^T, H = y.s/y.y, and the
//              two-loop recursion itself on (history x history) float64 scalars;
//   k_combine  d = -H g + sum_k aY_k y_k + sum_k aS_k s_k per element, stores the accepted
//              pair, g_prev = g, per-block partials of g.d, sum|g|, max|d|, max|g|;
//   k_stats    one workgroup folds those partials into the 8 floats the host reads.
// It is the same recursion as torch's (same pairs, same curvature test, same H), evaluated
// in exact-product float64 instead of float32, so it tracks the exact-arithmetic direction
// more closely than the float32 loop does.  All reductions have a fixed order: the result
// is deterministic and identical on every rank holding the same gradient.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pinnjet.h"

namespace {

constexpr int LB_MAX_HISTORY = 256;
constexpr int LB_CHUNK = 2048;      // elements per workgroup of k_dots / k_combine
constexpr int LB_BLOCK = 256;
constexpr int LB_EPT = LB_CHUNK / LB_BLOCK;
constexpr int LB_NDOT = 5;

// workspace header (ints), then float64 scalars, then float vectors; offsets in BYTES
struct LbLayout {
    int64_t hdr;        // int[8]: count, free slot, accept flag
    int64_t ord;        // int[hist+1]: physical slot of logical pair k (oldest first)
    int64_t H;          // double[2]: H_diag
    int64_t ro;         // double[hist+1] by physical slot
    int64_t b, c;       // double[hist+1]: s_j.g, y_j.g by physical slot
    int64_t SY, YY;     // double[(hist+1)^2] by physical slots: s_a.y_b, y_a.y_b
    int64_t coefY, coefS;  // double[hist+1] by logical index, coefG = coefY[hist+1 ...]
    int64_t part;       // double[(hist+1) * nchunk * LB_NDOT]
    int64_t spart;      // double[ncomb * 4]
    int64_t prev_g;     // float[np]
    int64_t S, Y;       // float[(hist+1) * np]
    int64_t total;
    int np, nchunk, ncomb, slots;
};
inline __host__ __device__ int64_t al256(int64_t v) { return (v + 255) & ~int64_t(255); }
inline __host__ __device__ LbLayout lb_layout(int64_t n, int hist) {
    LbLayout L;
    L.slots = hist + 1;
    L.np = (int)((n + 63) & ~int64_t(63));
    L.nchunk = (int)((n + LB_CHUNK - 1) / LB_CHUNK);
    L.ncomb = (int)((n + LB_BLOCK - 1) / LB_BLOCK);
    int64_t o = 0;
    L.hdr = o; o = al256(o + 8 * 4);
    L.ord = o; o = al256(o + int64_t(L.slots) * 4);
    L.H = o; o = al256(o + 16);
    L.ro = o; o = al256(o + int64_t(L.slots) * 8);
    L.b = o; o = al256(o + int64_t(L.slots) * 8);
    L.c = o; o = al256(o + int64_t(L.slots) * 8);
    L.SY = o; o = al256(o + int64_t(L.slots) * L.slots * 8);
    L.YY = o; o = al256(o + int64_t(L.slots) * L.slots * 8);
    L.coefY = o; o = al256(o + int64_t(L.slots + 1) * 8);
    L.coefS = o; o = al256(o + int64_t(L.slots + 1) * 8);
    L.part = o; o = al256(o + int64_t(L.slots) * L.nchunk * LB_NDOT * 8);
    L.spart = o; o = al256(o + int64_t(L.ncomb) * 4 * 8);
    L.prev_g = o; o = al256(o + int64_t(L.np) * 4);
    L.S = o; o = al256(o + int64_t(L.slots) * L.np * 4);
    L.Y = o; o = al256(o + int64_t(L.slots) * L.np * 4);
    L.total = o;
    return L;
}

template <typename T> __device__ inline T *at(char *ws, int64_t off) { return reinterpret_cast<T *>(ws + off); }

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// block-wide reduction (LB_BLOCK threads = 4 waves) of NV sums then NM maxima; valid on thread 0
template <int NV, int NM>
__device__ inline void block_reduce(double (&v)[NV + NM], double *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
#pragma unroll
    for (int k = NV; k < NV + NM; ++k) v[k] = wave_max(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV + NM; ++k) red[w * (NV + NM) + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV + NM; ++k) {
            double a = red[k];
            for (int i = 1; i < LB_BLOCK / 64; ++i)
                a = k < NV ? a + red[i * (NV + NM) + k] : fmax(a, red[i * (NV + NM) + k]);
            v[k] = a;
        }
    }
}

// ---- k_dots: grid (nchunk, slots).  Row = a live pair read from memory, or the candidate formed on the fly.
__global__ __launch_bounds__(LB_BLOCK) void k_dots(char *__restrict__ ws, int n, int hist, const float *__restrict__ g,
                                                   const float *__restrict__ d, float t) {
    __shared__ double red[(LB_BLOCK / 64) * LB_NDOT];
    const LbLayout L = lb_layout(n, hist);
    const int *hdr = at<int>(ws, L.hdr);
    const int *ord = at<int>(ws, L.ord);
    const int count = hdr[0], cand = hdr[1];
    const int slot = blockIdx.y, chunk = blockIdx.x;
    bool live = slot == cand;
    for (int k = 0; k < count && !live; ++k) live = ord[k] == slot;
    if (!live) return;                                   // uniform over the workgroup
    const float *prev_g = at<float>(ws, L.prev_g);
    const float *Sj = at<float>(ws, L.S) + int64_t(slot) * L.np;
    const float *Yj = at<float>(ws, L.Y) + int64_t(slot) * L.np;
    double a[LB_NDOT] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < LB_EPT; ++e) {
        const int i = chunk * LB_CHUNK + e * LB_BLOCK + threadIdx.x;
        if (i < n) {
            const float gi = g[i];
            const float yc = gi - prev_g[i], sc = d[i] * t;
            const float sj = slot == cand ? sc : Sj[i];
            const float yj = slot == cand ? yc : Yj[i];
            a[0] = fma((double)sj, (double)gi, a[0]);    // s_j.g
            a[1] = fma((double)yj, (double)gi, a[1]);    // y_j.g
            a[2] = fma((double)sj, (double)yc, a[2]);    // s_j.y_c
            a[3] = fma((double)sc, (double)yj, a[3]);    // s_c.y_j
            a[4] = fma((double)yj, (double)yc, a[4]);    // y_j.y_c
        }
    }
    block_reduce<LB_NDOT, 0>(a, red);
    if (threadIdx.x == 0) {
        double *p = at<double>(ws, L.part) + (int64_t(slot) * L.nchunk + chunk) * LB_NDOT;
#pragma unroll
        for (int k = 0; k < LB_NDOT; ++k) p[k] = a[k];
    }
}

// ---- k_scalars: one workgroup of 64 threads
// `ldsn` > 0: the live part of S.Y^T is staged (in logical order) into ldsn x ldsn doubles of dynamic LDS, so
// that a step of the sequential loops waits on LDS instead of on L2.
__global__ __launch_bounds__(64) void k_scalars(char *__restrict__ ws, int n, int hist, int first, int ldsn) {
    extern __shared__ double SYl[];
    __shared__ double al[LB_MAX_HISTORY + 1], cc[LB_MAX_HISTORY + 1], pd[(LB_MAX_HISTORY + 1) * LB_NDOT];
    __shared__ double Ak[LB_MAX_HISTORY + 1], rok[LB_MAX_HISTORY + 1], bk[LB_MAX_HISTORY + 1];
    __shared__ int sord[LB_MAX_HISTORY + 1];
    const LbLayout L = lb_layout(n, hist);
    int *hdr = at<int>(ws, L.hdr);
    int *ord = at<int>(ws, L.ord);
    double *Hp = at<double>(ws, L.H), *ro = at<double>(ws, L.ro), *b = at<double>(ws, L.b), *c = at<double>(ws, L.c);
    double *SY = at<double>(ws, L.SY), *YY = at<double>(ws, L.YY);
    double *coefY = at<double>(ws, L.coefY), *coefS = at<double>(ws, L.coefS);
    const double *part = at<double>(ws, L.part);
    const int lane = threadIdx.x, slots = L.slots;

    int count = first ? 0 : hdr[0];
    int cand = first ? 0 : hdr[1];
    double H = first ? 1.0 : Hp[0];
    int accept = 0;
    for (int k = lane; k < count; k += 64) sord[k] = ord[k];
    __syncthreads();

    if (!first) {
        // fold the per-chunk partials of every live row and of the candidate, in chunk order
        for (int r = lane; r <= count; r += 64) {
            const int slot = r < count ? sord[r] : cand;
            double a[LB_NDOT] = {0, 0, 0, 0, 0};
            for (int ch = 0; ch < L.nchunk; ++ch)
                for (int k = 0; k < LB_NDOT; ++k) a[k] += part[(int64_t(slot) * L.nchunk + ch) * LB_NDOT + k];
            for (int k = 0; k < LB_NDOT; ++k) pd[r * LB_NDOT + k] = a[k];
        }
        __syncthreads();
        const double ys = pd[count * LB_NDOT + 2], yy = pd[count * LB_NDOT + 4];
        accept = (float)ys > 1e-10f;
        // s_j.g and y_j.g of the live rows
        for (int r = lane; r < count; r += 64) {
            b[sord[r]] = pd[r * LB_NDOT + 0];
            c[sord[r]] = pd[r * LB_NDOT + 1];
        }
        if (accept) {
            for (int r = lane; r < count; r += 64) {
                const int j = sord[r];
                SY[int64_t(j) * slots + cand] = pd[r * LB_NDOT + 2];
                SY[int64_t(cand) * slots + j] = pd[r * LB_NDOT + 3];
                YY[int64_t(j) * slots + cand] = pd[r * LB_NDOT + 4];
                YY[int64_t(cand) * slots + j] = pd[r * LB_NDOT + 4];
            }
            if (lane == 0) {
                SY[int64_t(cand) * slots + cand] = ys;
                YY[int64_t(cand) * slots + cand] = yy;
                ro[cand] = 1.0 / ys;
                b[cand] = pd[count * LB_NDOT + 0];
                c[cand] = pd[count * LB_NDOT + 1];
            }
            H = ys / yy;
            __syncthreads();
            // history bookkeeping: drop the oldest pair when full; its slot becomes the next free one
            int freed;
            if (count == hist) {
                freed = sord[0];
                __syncthreads();
                int keep[(LB_MAX_HISTORY + 63) / 64];
                int m = 0;
                for (int k = lane; k + 1 < count; k += 64) keep[m++] = sord[k + 1];
                __syncthreads();
                m = 0;
                for (int k = lane; k + 1 < count; k += 64) sord[k] = keep[m++];
                if (lane == 0) sord[count - 1] = cand;
            } else {
                if (lane == 0) sord[count] = cand;
                ++count;
                freed = count;            // slots are handed out in order 0,1,2,... until the ring is full
            }
            cand = freed;
        }
        __syncthreads();
    }

    // two-loop recursion on scalars.  Logical k = 0 oldest .. count-1 newest, j = sord[k].
    const bool staged = ldsn >= count && ldsn > 0;
    if (staged)
        // eight loads in flight per lane (one at a time, every L2 round trip is on the critical path)
        for (int base = lane; base < count * count; base += 64 * 8) {
            double v[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + 64 * u;
                const int k = idx < count * count ? idx / count : 0, m = idx < count * count ? idx - k * count : 0;
                dst[u] = idx < count * count ? k * ldsn + m : -1;
                v[u] = SY[int64_t(sord[k]) * slots + sord[m]];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (dst[u] >= 0) SYl[dst[u]] = v[u];
        }
    for (int k = lane; k < count; k += 64) {
        rok[k] = ro[sord[k]];
        bk[k] = b[sord[k]];
    }
    __syncthreads();
    // Both loops are run "right-looking": lane l owns the rows k = l, l+64, ... and keeps their partial sums in
    // registers; a step computes ONE scalar on its owner lane, publishes it through LDS, and every lane folds
    // it into its own rows.  (The textbook left-looking form needs a wave-wide float64 reduction per step:
    // twelve cross-lane shuffles on the critical path of each of the 2*count steps.)
    constexpr int NK = (LB_MAX_HISTORY + 64) / 64;      // rows per lane, at most
    auto sy = [&](int k, int m) -> double {             // s_k . y_m, logical indices
        return staged ? SYl[k * ldsn + m] : SY[int64_t(sord[k]) * slots + sord[m]];
    };
    double acc[NK];
    const int nk = (count + 63) >> 6;                   // rows per lane actually in use (wave-uniform)
    //   al_m = ro_m ( -s_m.g - sum_{m'>m} al_m' s_m.y_m' ),  m = count-1 .. 0
#pragma unroll
    for (int i = 0; i < NK; ++i) acc[i] = 0.0;
    for (int m = count - 1; m >= 0; --m) {
        if (lane == (m & 63)) {
            double a = 0.0;
#pragma unroll
            for (int i = 0; i < NK; ++i) a = (i == (m >> 6)) ? acc[i] : a;
            al[m] = rok[m] * (-bk[m] - a);
        }
        __syncthreads();
        const double alm = al[m];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int k = lane + 64 * i;
            if (i < nk && k < m) acc[i] = fma(alm, sy(k, m), acc[i]);
        }
    }
    __syncthreads();
    //   be_k = ro_j y_j.r_k,  r_k = H(-g - sum_m al_m y_m) + sum_{m<k} cc_m s_m,  cc_k = al_k - be_k
    //   the y_j.(H q) part does not depend on earlier steps: all k in parallel first
    for (int k = lane; k < count; k += 64) {
        const int j = sord[k];
        double a = 0.0;
        for (int m0 = 0; m0 < count; m0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = YY[int64_t(j) * slots + sord[m0 + u < count ? m0 + u : 0]];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (m0 + u < count) a = fma(al[m0 + u], v[u], a);
        }
        Ak[k] = H * (-c[j] - a);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NK; ++i) acc[i] = 0.0;
    for (int m = 0; m < count; ++m) {
        if (lane == (m & 63)) {
            double e = 0.0;
#pragma unroll
            for (int i = 0; i < NK; ++i) e = (i == (m >> 6)) ? acc[i] : e;
            cc[m] = al[m] - rok[m] * (Ak[m] + e);
        }
        __syncthreads();
        const double ccm = cc[m];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int k = lane + 64 * i;
            if (i < nk && k > m && k < count) acc[i] = fma(ccm, sy(m, k), acc[i]);
        }
    }
    __syncthreads();
    for (int k = lane; k < count; k += 64) {
        coefY[k] = -H * al[k];
        coefS[k] = cc[k];
        ord[k] = sord[k];
    }
    if (lane == 0) {
        coefY[slots] = -H;     // coefficient of g
        hdr[0] = count;
        hdr[1] = cand;
        hdr[2] = accept;
        Hp[0] = H;
    }
}

// ---- k_combine: grid (ncomb), one element per thread.  `newest` = physical slot of the just-accepted pair
// (its values are still in registers here; this kernel stores them)
__global__ __launch_bounds__(LB_BLOCK) void k_combine(char *__restrict__ ws, int n, int hist, const float *__restrict__ g,
                                                      float *__restrict__ d, float t, int first) {
    __shared__ double red[(LB_BLOCK / 64) * 4];
    __shared__ double cY[LB_MAX_HISTORY], cS[LB_MAX_HISTORY];
    __shared__ int sord[LB_MAX_HISTORY];
    const LbLayout L = lb_layout(n, hist);
    const int *hdr = at<int>(ws, L.hdr);
    const int *ord = at<int>(ws, L.ord);
    const int count = hdr[0], accept = hdr[2];
    const double *coefY = at<double>(ws, L.coefY), *coefS = at<double>(ws, L.coefS);
    for (int k = threadIdx.x; k < count; k += LB_BLOCK) {
        cY[k] = coefY[k];
        cS[k] = coefS[k];
        sord[k] = ord[k];
    }
    __syncthreads();
    const int newest = accept ? sord[count - 1] : -1;
    const int stored = accept ? count - 1 : count;      // pairs whose rows are in memory (the newest one is not yet)
    const double cG = coefY[L.slots];
    float *prev_g = at<float>(ws, L.prev_g);
    float *S = at<float>(ws, L.S), *Y = at<float>(ws, L.Y);
    double st[4] = {0, 0, 0, 0};    // g.d, sum|g|, max|d|, max|g|
    const int i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i < n) {
        const float gi = g[i];
        const float yc = first ? 0.f : gi - prev_g[i];
        const float sc = first ? 0.f : d[i] * t;
        double acc = cG * (double)gi;
        int k = 0;
        for (; k + 4 <= stored; k += 4) {               // 8 independent loads in flight
            float yv[4], sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t row = int64_t(sord[k + u]) * L.np + i;
                yv[u] = Y[row];
                sv[u] = S[row];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = fma(cY[k + u], (double)yv[u], acc);
                acc = fma(cS[k + u], (double)sv[u], acc);
            }
        }
        for (; k < stored; ++k) {
            const int64_t row = int64_t(sord[k]) * L.np + i;
            acc = fma(cY[k], (double)Y[row], acc);
            acc = fma(cS[k], (double)S[row], acc);
        }
        if (newest >= 0) {
            acc = fma(cY[count - 1], (double)yc, acc);
            acc = fma(cS[count - 1], (double)sc, acc);
            Y[int64_t(newest) * L.np + i] = yc;
            S[int64_t(newest) * L.np + i] = sc;
        }
        const float di = (float)acc;
        d[i] = di;
        prev_g[i] = gi;
        st[0] = (double)gi * (double)di;
        st[1] = fabs((double)gi);
        st[2] = fabs((double)di);
        st[3] = fabs((double)gi);
    }
    block_reduce<2, 2>(st, red);
    if (threadIdx.x == 0) {
        double *p = at<double>(ws, L.spart) + int64_t(blockIdx.x) * 4;
        p[0] = st[0]; p[1] = st[1]; p[2] = st[2]; p[3] = st[3];
    }
}

// one wave: lane l folds partials l, l+64, ... in order, then the lanes are folded in a fixed tree
__global__ __launch_bounds__(64) void k_stats(const char *__restrict__ ws, int n, int hist, float *__restrict__ stats) {
    const LbLayout L = lb_layout(n, hist);
    const double *p = reinterpret_cast<const double *>(ws + L.spart);
    double a0 = 0, a1 = 0, m2 = 0, m3 = 0;
    for (int ch = threadIdx.x; ch < L.ncomb; ch += 64) {
        a0 += p[ch * 4 + 0];
        a1 += p[ch * 4 + 1];
        m2 = fmax(m2, p[ch * 4 + 2]);
        m3 = fmax(m3, p[ch * 4 + 3]);
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    m2 = wave_max(m2);
    m3 = wave_max(m3);
    if (threadIdx.x != 0) return;
    const int *hdr = reinterpret_cast<const int *>(ws + L.hdr);
    stats[0] = (float)a0;           // g.d
    stats[1] = (float)a1;           // sum |g|
    stats[2] = (float)m2;           // max |d|
    stats[3] = (float)m3;           // max |g|
    stats[4] = (float)hdr[0];       // pairs held
    stats[5] = (float)reinterpret_cast<const double *>(ws + L.H)[0];
    stats[6] = (float)hdr[2];       // the last pair was accepted
    stats[7] = 0.f;
}

// out = [loss, g.d, max|g|, sum|g|] after one closure evaluation of the line search
__global__ __launch_bounds__(1024) void k_lbfgs_eval_stats(const float *__restrict__ g, const float *__restrict__ d, int n,
                                                           const float *__restrict__ loss, float *__restrict__ out) {
    __shared__ double red[16 * 3];
    double s0 = 0, s1 = 0, m = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float gi = g[i];
        s0 = fma((double)gi, (double)d[i], s0);
        s1 += fabs((double)gi);
        m = fmax(m, fabs((double)gi));
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    m = wave_max(m);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        red[w * 3 + 0] = s0;
        red[w * 3 + 1] = s1;
        red[w * 3 + 2] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) {
            s0 += red[i * 3 + 0];
            s1 += red[i * 3 + 1];
            m = fmax(m, red[i * 3 + 2]);
        }
        out[0] = loss ? loss[0] : 0.f;
        out[1] = (float)s0;
        out[2] = (float)m;
        out[3] = (float)s1;
    }
}

inline bool bad_sizes(int64_t n, int32_t history) {
    return n <= 0 || history <= 0 || history > LB_MAX_HISTORY || n > (int64_t(1) << 24);
}

}  // namespace

extern "C" {

int64_t pinn_lbfgs_workspace_floats(int64_t n, int32_t history) {
    if (bad_sizes(n, history)) return PINN_ERR_INVALID_DESC;
    return lb_layout(n, history).total / 4;
}

int pinn_lbfgs_direction(float *workspace, int64_t n, int32_t history, const float *g, float *d, float t,
                         int32_t first, float *stats8, void *stream) {
    if (!workspace || !g || !d || !stats8) return PINN_ERR_NULL;
    if (bad_sizes(n, history)) return PINN_ERR_INVALID_DESC;
    if ((reinterpret_cast<uintptr_t>(workspace) & 7) != 0) return PINN_ERR_INVALID_DESC;
    const LbLayout L = lb_layout(n, history);
    char *ws = reinterpret_cast<char *>(workspace);
    hipStream_t s = (hipStream_t)stream;
    if (!first)
        hipLaunchKernelGGL(k_dots, dim3(L.nchunk, L.slots), dim3(LB_BLOCK), 0, s, ws, (int)n, (int)history, g, d, t);
    const int ldsn = (history + 1) * (history + 1) * 8 <= 40 * 1024 ? history + 1 : 0;   // static LDS of k_scalars: 22 KB
    hipLaunchKernelGGL(k_scalars, dim3(1), dim3(64), size_t(ldsn) * ldsn * 8, s, ws, (int)n, (int)history, (int)first,
                       ldsn);
    hipLaunchKernelGGL(k_combine, dim3(L.ncomb), dim3(LB_BLOCK), 0, s, ws, (int)n, (int)history, g, d, t, (int)first);
    hipLaunchKernelGGL(k_stats, dim3(1), dim3(64), 0, s, ws, (int)n, (int)history, stats8);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

int pinn_lbfgs_eval_stats(const float *g, const float *d, int64_t n, const float *loss, float *out4, void *stream) {
    if (!g || !d || !out4) return PINN_ERR_NULL;
    if (n <= 0 || n > (int64_t(1) << 24)) return PINN_ERR_INVALID_DESC;
    hipLaunchKernelGGL(k_lbfgs_eval_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, d, (int)n, loss, out4);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

}  // extern "C"
