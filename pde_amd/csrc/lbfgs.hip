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

#include <type_traits>

#include "../../include/pinnjet.h"

namespace {

constexpr int LB_MAX_HISTORY = 256;
constexpr int LB_CHUNK = 2048;      // elements per workgroup of k_dots / k_combine
constexpr int LB_BLOCK = 256;
constexpr int LB_EPT = LB_CHUNK / LB_BLOCK;
constexpr int LB_NDOT = 5;

// workspace header (ints), then float64 scalars, then float vectors; offsets in BYTES
struct LbLayout {
    int64_t hdr;        // int[8]: count, free slot, accept flag, ticket of k_combine's last-block fold
    int64_t ord;        // int[hist+1]: physical slot of logical pair k (oldest first)
    int64_t live;       // int[hist+1]: physical slot holds a stored pair
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
    L.ncomb = (int)((n + 63) / 64);
    int64_t o = 0;
    L.hdr = o; o = al256(o + 8 * 4);
    L.ord = o; o = al256(o + int64_t(L.slots) * 4);
    L.live = o; o = al256(o + int64_t(L.slots) * 4);
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
    const int cand = hdr[1];
    const int slot = blockIdx.y, chunk = blockIdx.x;
    if (slot != cand && at<int>(ws, L.live)[slot] == 0) return;      // uniform over the workgroup
    const float *prev_g = at<float>(ws, L.prev_g);
    const float *Sj = at<float>(ws, L.S) + int64_t(slot) * L.np;
    const float *Yj = at<float>(ws, L.Y) + int64_t(slot) * L.np;
    double a[LB_NDOT] = {0, 0, 0, 0, 0};
    // every load of the thread's LB_EPT elements is issued before the first use (indices past the end are clamped and
    // their products dropped): one memory round trip per workgroup instead of one to three per element
    float gv[LB_EPT], pv[LB_EPT], dv[LB_EPT], sv[LB_EPT], yv[LB_EPT];
    const bool is_cand = slot == cand;                   // uniform over the workgroup
#pragma unroll
    for (int e = 0; e < LB_EPT; ++e) {
        const int i = chunk * LB_CHUNK + e * LB_BLOCK + threadIdx.x;
        const int ii = i < n ? i : n - 1;
        gv[e] = g[ii];
        pv[e] = prev_g[ii];
        dv[e] = d[ii];
    }
    if (!is_cand) {
#pragma unroll
        for (int e = 0; e < LB_EPT; ++e) {
            const int i = chunk * LB_CHUNK + e * LB_BLOCK + threadIdx.x;
            const int ii = i < n ? i : n - 1;
            sv[e] = Sj[ii];
            yv[e] = Yj[ii];
        }
    }
#pragma unroll
    for (int e = 0; e < LB_EPT; ++e) {
        const int i = chunk * LB_CHUNK + e * LB_BLOCK + threadIdx.x;
        if (i < n) {
            const float gi = gv[e];
            const float yc = gi - pv[e], sc = dv[e] * t;
            const float sj = is_cand ? sc : sv[e];
            const float yj = is_cand ? yc : yv[e];
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

// ---- k_scalars: one workgroup of LB_BLOCK threads.
// The parallel parts (folding the partials, the new Gram row/column, staging, the y.(Hq) products) use all four
// waves; the two sequential loops run on wave 0 alone, with no LDS round trip and no barrier on their critical
// path: lane l owns the rows k = l, l+64, ... and keeps their partial sums in registers ("right-looking" form);
// a step computes ONE scalar on its owner lane, v_readlane broadcasts it, and every lane folds it into its own
// rows.  (The textbook left-looking form needs a wave-wide float64 reduction per step.)  The columns (first
// loop) / rows (second loop) of S.Y^T that a run of steps needs are staged by all waves, in logical order, into
// a panel of `pcols` x ldc doubles of dynamic LDS; history 50 fits one panel, history 150 takes two.
__device__ inline double readlane_f64(double v, int l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int NK>       // rows of the history per lane of wave 0: 1 up to history 63, 5 up to LB_MAX_HISTORY
__global__ __launch_bounds__(LB_BLOCK) void k_scalars(char *__restrict__ ws, int n, int hist, int first, int pcols, int whole) {
    extern __shared__ double panel[];
    __shared__ double al[LB_MAX_HISTORY + 1], cc[LB_MAX_HISTORY + 1], Ak[LB_MAX_HISTORY + 1];
    __shared__ double rok[LB_MAX_HISTORY + 1], bk[LB_MAX_HISTORY + 1], ck[LB_MAX_HISTORY + 1];
    __shared__ double pd[(LB_MAX_HISTORY + 1) * LB_NDOT];       // folded partials; later: the 4 part sums of y.(Hq)
    __shared__ int sord[LB_MAX_HISTORY + 1], snew[LB_MAX_HISTORY + 1];
    const LbLayout L = lb_layout(n, hist);
    int *hdr = at<int>(ws, L.hdr);
    int *ord = at<int>(ws, L.ord);
    int *livef = at<int>(ws, L.live);
    double *Hp = at<double>(ws, L.H), *ro = at<double>(ws, L.ro), *b = at<double>(ws, L.b), *c = at<double>(ws, L.c);
    double *SY = at<double>(ws, L.SY), *YY = at<double>(ws, L.YY);
    double *coefY = at<double>(ws, L.coefY), *coefS = at<double>(ws, L.coefS);
    const double *part = at<double>(ws, L.part);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slots = L.slots;
    const int ldc = slots | 1;

    int count = first ? 0 : hdr[0];
    int cand = first ? 0 : hdr[1];
    double H = first ? 1.0 : Hp[0];
    int accept = 0;
    for (int k = tid; k < count; k += LB_BLOCK) sord[k] = ord[k];
    __syncthreads();

    if (!first) {
        // fold the per-chunk partials of every live row and of the candidate, in chunk order
        for (int idx = tid; idx < (count + 1) * LB_NDOT; idx += LB_BLOCK) {
            const int r = idx / LB_NDOT, k = idx - r * LB_NDOT;
            const int slot = r < count ? sord[r] : cand;
            const double *p = part + int64_t(slot) * L.nchunk * LB_NDOT + k;
            double a = 0.0;
            for (int ch = 0; ch < L.nchunk; ++ch) a += p[ch * LB_NDOT];
            pd[idx] = a;
        }
        __syncthreads();
        const double ys = pd[count * LB_NDOT + 2], yy = pd[count * LB_NDOT + 4];
        accept = (float)ys > 1e-10f;
        // s_j.g and y_j.g of the live rows
        for (int r = tid; r < count; r += LB_BLOCK) {
            b[sord[r]] = pd[r * LB_NDOT + 0];
            c[sord[r]] = pd[r * LB_NDOT + 1];
        }
        if (accept) {
            for (int r = tid; r < count; r += LB_BLOCK) {
                const int j = sord[r];
                SY[int64_t(j) * slots + cand] = pd[r * LB_NDOT + 2];
                SY[int64_t(cand) * slots + j] = pd[r * LB_NDOT + 3];
                YY[int64_t(j) * slots + cand] = pd[r * LB_NDOT + 4];
                YY[int64_t(cand) * slots + j] = pd[r * LB_NDOT + 4];
            }
            if (tid == 0) {
                SY[int64_t(cand) * slots + cand] = ys;
                YY[int64_t(cand) * slots + cand] = yy;
                ro[cand] = 1.0 / ys;
                b[cand] = pd[count * LB_NDOT + 0];
                c[cand] = pd[count * LB_NDOT + 1];
            }
            H = ys / yy;
            // history bookkeeping: drop the oldest pair when full; its slot becomes the next free one
            int freed;
            if (count == hist) {
                freed = sord[0];
                for (int k = tid; k + 1 < count; k += LB_BLOCK) snew[k] = sord[k + 1];
                if (tid == 0) snew[count - 1] = cand;
            } else {
                for (int k = tid; k < count; k += LB_BLOCK) snew[k] = sord[k];
                if (tid == 0) snew[count] = cand;
                ++count;
                freed = count;            // slots are handed out in order 0,1,2,... until the ring is full
            }
            __syncthreads();
            for (int k = tid; k < count; k += LB_BLOCK) sord[k] = snew[k];
            cand = freed;
        }
        __syncthreads();
    }

    for (int k = tid; k < count; k += LB_BLOCK) {
        const int j = sord[k];
        rok[k] = ro[j];
        bk[k] = b[j];
        ck[k] = c[j];
    }
    __syncthreads();

    // stage `nl` lines of a Gram matrix G (columns m when `cols`, rows otherwise) starting at logical index m0:
    // dst[(m - m0) * ldc + k] = G[k][m] (cols) or G[m][k] (rows), k < count, logical indices; threads t0, t0+nt, ...
    // of the workgroup take part; sixteen loads in flight per thread
    auto stage = [&](const double *G, double *dst, int m0, int nl, bool cols, int t0, int nt) {
        constexpr int U = 16;
        const int tot = nl * count;
        for (int base = tid - t0; base < tot; base += nt * U) {
            double v[U];
            int di[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + nt * u;
                const bool ok = idx < tot;
                const int mm = ok ? idx / count : 0, k = ok ? idx - mm * count : 0;
                di[u] = ok ? mm * ldc + k : -1;
                const int jm = sord[m0 + mm], jk = sord[k];
                v[u] = cols ? G[int64_t(jk) * slots + jm] : G[int64_t(jm) * slots + jk];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (di[u] >= 0) dst[di[u]] = v[u];
        }
    };

    double acc[NK], alr[NK], own_ro[NK], own_b[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int k = lane + 64 * i;
        acc[i] = 0.0;
        alr[i] = 0.0;
        own_ro[i] = k < count ? rok[k] : 0.0;
        own_b[i] = k < count ? bk[k] : 0.0;
    }
    // The sequential loops on wave 0, in blocks of 64 steps whose owner row index OI = m >> 6 is a compile-time constant:
    // no select over the rows a lane owns, only the rows that can take part are touched (first loop: rows i <= OI,
    // second loop: rows i >= OI), and the panel values of the next step are loaded before the current step's arithmetic
    // (their addresses do not depend on the recursion; left to itself every step waited for its own LDS reads).
    //   first loop:  al_m = ro_m ( -s_m.g - sum_{m'>m} al_m' s_m.y_m' ),  m descending
    auto loop1_blk = [&](auto oic, const double *pan, int mlo, int lo, int hi) {
        constexpr int OI = decltype(oic)::value;
        if constexpr (OI < NK) {
            double cur[OI + 1], nxt[OI + 1];
            auto fetch = [&](int m, double *dst) {
#pragma unroll
                for (int i = 0; i <= OI; ++i) {
                    const int k = lane + 64 * i;
                    dst[i] = pan[(m - mlo) * ldc + (k < count ? k : count - 1)];
                }
            };
            fetch(hi - 1, cur);
            for (int m = hi - 1; m >= lo; --m) {
                const int ol = __builtin_amdgcn_readfirstlane(m & 63);
                fetch(m > lo ? m - 1 : m, nxt);
                const double alm = readlane_f64(own_ro[OI] * (-own_b[OI] - acc[OI]), ol);
                if (lane == ol) alr[OI] = alm;
#pragma unroll
                for (int i = 0; i < OI; ++i) acc[i] = fma(alm, cur[i], acc[i]);
                if (lane < ol) acc[OI] = fma(alm, cur[OI], acc[OI]);
#pragma unroll
                for (int i = 0; i <= OI; ++i) cur[i] = nxt[i];
            }
        }
    };
    auto loop1 = [&](const double *pan, int mlo, int mhi) {
        auto blk = [&](auto oic) {
            constexpr int OI = decltype(oic)::value;
            const int lo = mlo > 64 * OI ? mlo : 64 * OI, hi = mhi < 64 * (OI + 1) ? mhi : 64 * (OI + 1);
            if (lo < hi) loop1_blk(oic, pan, mlo, lo, hi);
        };
        blk(std::integral_constant<int, 4>{});
        blk(std::integral_constant<int, 3>{});
        blk(std::integral_constant<int, 2>{});
        blk(std::integral_constant<int, 1>{});
        blk(std::integral_constant<int, 0>{});
    };
    double own_A[NK], ccr[NK];
    //   second loop:  cc_m = al_m - ro_m (A_m + sum_{m'<m} cc_m' s_m'.y_m),  m ascending
    auto loop2_blk = [&](auto oic, const double *pan, int mlo, int lo, int hi) {
        constexpr int OI = decltype(oic)::value;
        if constexpr (OI < NK) {
            double cur[NK - OI], nxt[NK - OI];
            auto fetch = [&](int m, double *dst) {
#pragma unroll
                for (int i = OI; i < NK; ++i) {
                    const int k = lane + 64 * i;
                    dst[i - OI] = pan[(m - mlo) * ldc + (k < count ? k : count - 1)];
                }
            };
            fetch(lo, cur);
            for (int m = lo; m < hi; ++m) {
                const int ol = __builtin_amdgcn_readfirstlane(m & 63);
                fetch(m + 1 < hi ? m + 1 : m, nxt);
                const double ccm = readlane_f64(alr[OI] - own_ro[OI] * (own_A[OI] + acc[OI]), ol);
                if (lane == ol) ccr[OI] = ccm;
                if (lane > ol) acc[OI] = fma(ccm, cur[0], acc[OI]);
#pragma unroll
                for (int i = OI + 1; i < NK; ++i) acc[i] = fma(ccm, cur[i - OI], acc[i]);   // (rows k >= count: never read)
#pragma unroll
                for (int i = 0; i < NK - OI; ++i) cur[i] = nxt[i];
            }
        }
    };
    auto loop2 = [&](const double *pan, int mlo, int mhi) {
        auto blk = [&](auto oic) {
            constexpr int OI = decltype(oic)::value;
            const int lo = mlo > 64 * OI ? mlo : 64 * OI, hi = mhi < 64 * (OI + 1) ? mhi : 64 * (OI + 1);
            if (lo < hi) loop2_blk(oic, pan, mlo, lo, hi);
        };
        blk(std::integral_constant<int, 0>{});
        blk(std::integral_constant<int, 1>{});
        blk(std::integral_constant<int, 2>{});
        blk(std::integral_constant<int, 3>{});
        blk(std::integral_constant<int, 4>{});
    };
    auto publish_al = [&]() {
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int k = lane + 64 * i;
                if (k < count) al[k] = alr[i];
            }
        }
    };
    //   be_k = ro_j y_j.r_k,  r_k = H(-g - sum_m al_m y_m) + sum_{m<k} cc_m s_m,  cc_k = al_k - be_k
    //   the y_j.(H q) part A_k does not depend on earlier steps: all k at once, four threads per row (m = part, part+4,
    //   ...), the four sums folded in a fixed order.  `yy`: Y.Y^T staged in LDS by rows, or NULL (read through L2)
    auto fold_A = [&](const double *yy) {
        for (int k = tid >> 2; k < count; k += LB_BLOCK / 4) {
            const int j = sord[k], pt = tid & 3;
            double a = 0.0;
            if (yy) {
                for (int m = pt; m < count; m += 4) a = fma(al[m], yy[k * ldc + m], a);
            } else {
                for (int m0 = pt; m0 < count; m0 += 64) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = YY[int64_t(j) * slots + sord[m0 + 4 * u < count ? m0 + 4 * u : 0]];
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (m0 + 4 * u < count) a = fma(al[m0 + 4 * u], v[u], a);
                }
            }
            pd[k * 4 + pt] = a;
        }
        __syncthreads();
        for (int k = tid; k < count; k += LB_BLOCK)
            Ak[k] = H * (-ck[k] - ((pd[k * 4] + pd[k * 4 + 1]) + (pd[k * 4 + 2] + pd[k * 4 + 3])));
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int k = lane + 64 * i;
            acc[i] = 0.0;
            ccr[i] = 0.0;
            own_A[i] = k < count ? Ak[k] : 0.0;
        }
    };

    if (whole) {
        // everything fits: three images (columns of S.Y^T, its rows, Y.Y^T).  While wave 0 runs the first loop the
        // other three waves stage what the rest needs.
        double *pan2 = panel + pcols * ldc, *yyl = pan2 + pcols * ldc;
        stage(SY, panel, 0, count, true, 0, LB_BLOCK);
        __syncthreads();
        if (wave == 0) {
            loop1(panel, 0, count);
        } else {
            stage(SY, pan2, 0, count, false, 64, LB_BLOCK - 64);
            stage(YY, yyl, 0, count, false, 64, LB_BLOCK - 64);
        }
        publish_al();
        __syncthreads();
        fold_A(yyl);
        if (wave == 0) loop2(pan2, 0, count);
    } else if (count > 0) {
        // Larger histories: half panels, one read by wave 0 while the other three waves stage the next one (the last
        // panel of the first loop is overlapped with the first panel of the second loop: rows 0 .. pc-1).
        const int pc = pcols >> 1;
        double *pb[2] = {panel, panel + pc * ldc};
        const int np = (count + pc - 1) / pc;
        stage(SY, pb[0], count > pc ? count - pc : 0, count > pc ? pc : count, true, 0, LB_BLOCK);
        __syncthreads();
        for (int i = 0; i < np; ++i) {
            const int mhi = count - i * pc, mlo = mhi > pc ? mhi - pc : 0;
            if (wave == 0) {
                loop1(pb[i & 1], mlo, mhi);
            } else if (i + 1 < np) {
                const int nlo = mlo > pc ? mlo - pc : 0;
                stage(SY, pb[(i + 1) & 1], nlo, mlo - nlo, true, 64, LB_BLOCK - 64);
            } else {
                stage(SY, pb[(i + 1) & 1], 0, count > pc ? pc : count, false, 64, LB_BLOCK - 64);
            }
            __syncthreads();
        }
        publish_al();
        __syncthreads();
        fold_A(nullptr);
        for (int i = 0; i < np; ++i) {
            const int mlo = i * pc, mhi = mlo + pc < count ? mlo + pc : count;
            const int cur = (np + i) & 1;
            if (wave == 0) {
                loop2(pb[cur], mlo, mhi);
            } else if (i + 1 < np) {
                const int nhi = mhi + pc < count ? mhi + pc : count;
                stage(SY, pb[cur ^ 1], mhi, nhi - mhi, false, 64, LB_BLOCK - 64);
            }
            __syncthreads();
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int k = lane + 64 * i;
            if (k < count) {
                coefY[k] = -H * alr[i];
                coefS[k] = ccr[i];
            }
        }
    }
    for (int sl = tid; sl < slots; sl += LB_BLOCK) livef[sl] = 0;
    __syncthreads();
    for (int k = tid; k < count; k += LB_BLOCK) {
        ord[k] = sord[k];
        livef[sord[k]] = 1;
    }
    if (tid == 0) {
        coefY[slots] = -H;     // coefficient of g
        hdr[0] = count;
        hdr[1] = cand;
        hdr[2] = accept;
        hdr[3] = 0;            // ticket of k_combine's last-block fold
        Hp[0] = H;
    }
}

// ---- k_combine: grid (ncomb), 64 elements per workgroup: wave w sums the stored rows k = w, w+4, ... of its element
// (four times as many loads in flight as one thread per element would have), wave 0 adds the four sums in a fixed order
// and finishes the element.  `newest` = physical slot of the just-accepted pair (its values are still in registers here;
// this kernel stores them)
__global__ __launch_bounds__(LB_BLOCK) void k_combine(char *__restrict__ ws, int n, int hist, const float *__restrict__ g,
                                                      float *__restrict__ d, float t, int first, float *__restrict__ stats,
                                                      float *__restrict__ x, float *__restrict__ x0, float t_next) {
    __shared__ double red[(LB_BLOCK / 64) * 4];
    __shared__ int last_block;
    __shared__ double cY[LB_MAX_HISTORY], cS[LB_MAX_HISTORY];
    __shared__ double psum[3][64];
    __shared__ int sord[LB_MAX_HISTORY];
    const LbLayout L = lb_layout(n, hist);
    int *hdr = at<int>(ws, L.hdr);
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
    const int wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ic = i < n ? i : n - 1;
    // wave 0's own operands are requested before the history sums (nothing of theirs is needed until after the barrier)
    float gi = 0.f, pgi = 0.f, dold = 0.f, xi = 0.f;
    if (wv == 0) {
        gi = g[ic];
        if (!first) {
            pgi = prev_g[ic];
            dold = d[ic];
        }
        if (x) xi = x[ic];
    }
    double acc = 0.0;
    // the wave's rows k = wv, wv+4, ...: sixteen rows (32 loads) in flight per pass - one pass up to history 64
    for (int kb = wv; kb < stored; kb += 64) {
        float yv[16], sv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = kb + 4 * u;
            const int64_t row = int64_t(sord[k < stored ? k : kb]) * L.np + ic;
            yv[u] = Y[row];
            sv[u] = S[row];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = kb + 4 * u;
            if (k < stored) {
                acc = fma(cY[k], (double)yv[u], acc);
                acc = fma(cS[k], (double)sv[u], acc);
            }
        }
    }
    if (wv > 0) psum[wv - 1][threadIdx.x & 63] = acc;
    __syncthreads();
    if (wv == 0 && i < n) {
        const float yc = first ? 0.f : gi - pgi;
        const float sc = first ? 0.f : dold * t;
        const int e = threadIdx.x;
        acc = cG * (double)gi + ((acc + psum[0][e]) + (psum[1][e] + psum[2][e]));
        if (newest >= 0) {
            acc = fma(cY[count - 1], (double)yc, acc);
            acc = fma(cS[count - 1], (double)sc, acc);
            Y[int64_t(newest) * L.np + i] = yc;
            S[int64_t(newest) * L.np + i] = sc;
        }
        const float di = (float)acc;
        d[i] = di;
        prev_g[i] = gi;
        if (x) {                                        // the line search's first trial point, x0 kept for the others
            x0[i] = xi;
            x[i] = xi + t_next * di;
        }
        st[0] = (double)gi * (double)di;
        st[1] = fabs((double)gi);
        st[2] = fabs((double)di);
        st[3] = fabs((double)gi);
    }
    block_reduce<2, 2>(st, red);
    if (threadIdx.x == 0) {
        double *p = at<double>(ws, L.spart) + int64_t(blockIdx.x) * 4;
        p[0] = st[0]; p[1] = st[1]; p[2] = st[2]; p[3] = st[3];
        // the workgroup that takes the last ticket folds everybody's partials (in index order: the result does
        // not depend on which workgroup that is)
        __threadfence();
        last_block = atomicAdd(&hdr[3], 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_block || threadIdx.x >= 64) return;
    __threadfence();
    // one wave: lane l folds partials l, l+64, ... in order, then the lanes are folded in a fixed tree
    const double *p = reinterpret_cast<const double *>(ws + L.spart);
    double a0 = 0, a1 = 0, m2 = 0, m3 = 0;
    for (int ch = threadIdx.x; ch < L.ncomb; ch += 64) {
        a0 += __hip_atomic_load(p + ch * 4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a1 += __hip_atomic_load(p + ch * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m2 = fmax(m2, __hip_atomic_load(p + ch * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        m3 = fmax(m3, __hip_atomic_load(p + ch * 4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    m2 = wave_max(m2);
    m3 = wave_max(m3);
    if (threadIdx.x != 0) return;
    stats[0] = (float)a0;           // g.d
    stats[1] = (float)a1;           // sum |g|
    stats[2] = (float)m2;           // max |d|
    stats[3] = (float)m3;           // max |g|
    stats[4] = (float)hdr[0];       // pairs held
    stats[5] = (float)reinterpret_cast<const double *>(ws + L.H)[0];
    stats[6] = (float)hdr[2];       // the last pair was accepted
    stats[7] = 0.f;
    hdr[3] = 0;
}

// out = [loss, g.d, max|g|, sum|g|] after one closure evaluation of the line search
__global__ __launch_bounds__(1024) void k_lbfgs_eval_stats(const float *__restrict__ g, const float *__restrict__ d, int n,
                                                           const float *__restrict__ loss, float *__restrict__ out,
                                                           float *__restrict__ g_copy, const float *__restrict__ stats8,
                                                           float *host13, int ticket) {
    __shared__ double red[16 * 3];
    double s0 = 0, s1 = 0, m = 0;
    // eight elements per thread and pass, all loads first (left as a plain loop every element waited for its own two loads)
    for (int base = threadIdx.x; base < n; base += 1024 * 8) {
        float gv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 1024 * u, ii = i < n ? i : n - 1;
            gv[u] = g[ii];
            dv[u] = d[ii];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 1024 * u;
            if (i < n) {
                const float gi = gv[u];
                if (g_copy) g_copy[i] = gi;
                s0 = fma((double)gi, (double)dv[u], s0);
                s1 += fabs((double)gi);
                m = fmax(m, fabs((double)gi));
            }
        }
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
        const float o0 = loss ? loss[0] : 0.f, o1 = (float)s0, o2 = (float)m, o3 = (float)s1;
        out[0] = o0;
        out[1] = o1;
        out[2] = o2;
        out[3] = o3;
        if (host13) {
            // the host's copy, in pinned host memory: the direction's statistics (an earlier launch of this stream
            // wrote them), this evaluation's, then - after a system-scope fence - the ticket the host polls for
            if (stats8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) host13[k] = stats8[k];
            }
            host13[8] = o0;
            host13[9] = o1;
            host13[10] = o2;
            host13[11] = o3;
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<int *>(host13) + 12, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
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

int pinn_lbfgs_direction_advance(float *workspace, int64_t n, int32_t history, const float *g, float *d, float t,
                                 int32_t first, float *stats8, float *x, float *x0, float t_next, void *stream) {
    if (!workspace || !g || !d || !stats8) return PINN_ERR_NULL;
    if ((x == nullptr) != (x0 == nullptr)) return PINN_ERR_NULL;
    if (bad_sizes(n, history)) return PINN_ERR_INVALID_DESC;
    if ((reinterpret_cast<uintptr_t>(workspace) & 7) != 0) return PINN_ERR_INVALID_DESC;
    const LbLayout L = lb_layout(n, history);
    char *ws = reinterpret_cast<char *>(workspace);
    hipStream_t s = (hipStream_t)stream;
    if (!first)
        hipLaunchKernelGGL(k_dots, dim3(L.nchunk, L.slots), dim3(LB_BLOCK), 0, s, ws, (int)n, (int)history, g, d, t);
    // panel of k_scalars: as many lines of S.Y^T as fit 96 KB of dynamic LDS (its static arrays take 31 KB)
    // (or, when three whole images fit - history <= 72 - all of S.Y^T by columns, by rows, and Y.Y^T)
    const int ldc = L.slots | 1;
    int pcols = (96 * 1024) / (ldc * 8);
    if (pcols > L.slots) pcols = L.slots;
    const int whole = size_t(3) * L.slots * ldc * 8 <= 128 * 1024;
    const size_t dyn = size_t(whole ? 3 : 1) * pcols * ldc * 8;
    auto kern = L.slots <= 64 ? k_scalars<1> : k_scalars<(LB_MAX_HISTORY + 64) / 64>;
    if (dyn > 32 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
        (void)hipGetLastError();
        return PINN_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(LB_BLOCK), dyn, s, ws, (int)n, (int)history, (int)first, pcols, whole);
    hipLaunchKernelGGL(k_combine, dim3(L.ncomb), dim3(LB_BLOCK), 0, s, ws, (int)n, (int)history, g, d, t, (int)first,
                       stats8, x, x0, t_next);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

int pinn_lbfgs_direction(float *workspace, int64_t n, int32_t history, const float *g, float *d, float t,
                         int32_t first, float *stats8, void *stream) {
    return pinn_lbfgs_direction_advance(workspace, n, history, g, d, t, first, stats8, nullptr, nullptr, 0.f, stream);
}

int pinn_lbfgs_eval_stats_keep(const float *g, const float *d, int64_t n, const float *loss, float *out4, float *g_copy,
                               const float *stats8, float *host13, int32_t ticket, void *stream) {
    if (!g || !d || !out4) return PINN_ERR_NULL;
    if (n <= 0 || n > (int64_t(1) << 24)) return PINN_ERR_INVALID_DESC;
    hipLaunchKernelGGL(k_lbfgs_eval_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, d, (int)n, loss, out4, g_copy,
                       stats8, host13, (int)ticket);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

int pinn_lbfgs_eval_stats(const float *g, const float *d, int64_t n, const float *loss, float *out4, void *stream) {
    return pinn_lbfgs_eval_stats_keep(g, d, n, loss, out4, nullptr, nullptr, nullptr, 0, stream);
}

// Host side of one line-search evaluation in the usual case, without the interpreter between the arrival of the
// statistics and the next launch: wait for the ticket of pinn_lbfgs_eval_stats_keep's host mirror, copy the 12 floats
// out, and - when the first trial point t = lr satisfies both Wolfe conditions and no stopping rule of
// torch.optim.LBFGS.step fires - launch the next direction update at once.  The tests are those of pde_amd/lbfgs.py
// (torch/optim/lbfgs.py: _strong_wolfe's first two tests with c1 = 1e-4, c2 = 0.9, then step()'s), in double precision
// and one rounding per operation like the interpreter's (volatile temporaries: no fused multiply-add).
int pinn_lbfgs_poll_advance(const float *host13, int32_t ticket, int64_t max_spins, double f_prev, double lr, double tol_x,
                            double tol_g, int32_t allow, float *workspace, int64_t n, int32_t history, const float *g_keep,
                            float *d, float *stats8, float *x, float *x0, void *stream, float *vals12) {
    if (!host13 || !vals12) return PINN_ERR_NULL;
    const volatile int *tk = reinterpret_cast<const volatile int *>(host13) + 12;
    int64_t spins = 0;
    while (*tk != ticket) {
        if (++spins > max_spins) return 2;              // not there yet: the caller waits on the stream and calls again
#if !defined(__HIP_DEVICE_COMPILE__)
        __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const volatile float *hv = host13;
    for (int k = 0; k < 12; ++k) vals12[k] = hv[k];
    if (!allow) return 0;
    const double gtd = vals12[0], dmax = vals12[2], fs = vals12[8], gtds = vals12[9], gmaxs = vals12[10];
    if (gtd > -tol_x) return 0;                         // not a descent direction: the caller undoes the evaluation
    volatile double a = 1e-4 * lr;
    a = a * gtd;
    a = f_prev + a;
    if (fs > a) return 0;                               // sufficient decrease fails: the caller brackets / zooms
    volatile double c = -0.9 * gtd;
    if (!(__builtin_fabs(gtds) <= c)) return 0;         // curvature condition fails
    if (gmaxs <= tol_g) return 0;                       // step()'s stopping rules
    volatile double m = dmax * __builtin_fabs(lr);
    if (m <= tol_x) return 0;
    volatile double df = fs - f_prev;
    if (__builtin_fabs(df) < tol_x) return 0;
    const int rc = pinn_lbfgs_direction_advance(workspace, n, history, g_keep, d, (float)lr, 0, stats8, x, x0, (float)lr, stream);
    return rc == PINN_OK ? 1 : rc;
}

int pinn_host_device_pointer(void *host, void **device) {
    if (!host || !device) return PINN_ERR_NULL;
    *device = nullptr;
    if (hipHostGetDevicePointer(device, host, 0) != hipSuccess || !*device) {
        (void)hipGetLastError();
        *device = nullptr;
        return PINN_ERR_HIP;
    }
    return PINN_OK;
}

}  // extern "C"
