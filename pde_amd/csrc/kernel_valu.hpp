// Generic VALU kernels: one collocation point per lane, any layer widths.
//
// This is the fallback/narrow-net path (hidden width < 32, or shapes the MFMA
// path does not cover).  Per-point jets of every layer live in a global
// scratch laid out [slot][thread] so that each access is one coalesced
// 256-B wave transaction; weights are wave-uniform reads; weight gradients
// are reduced across the wave with DPP shuffles, accumulated per workgroup in
// LDS (float atomics from the four waves: arrival order, so the last bits differ run to run) and
// written as one partial slab per workgroup; only the sum over workgroups (k_reduce_slabs) is fixed-order.
//
// Reference semantics: forward = Scale + DNN (7kdv/PI-7kdv3soliton.py:34-61),
// jets = net_f's derivative chain (:122-132), reverse = loss.backward() (:111).
#pragma once
#include "jet_math.hpp"
#include "pinn_common.hpp"

namespace pinn {

constexpr int VALU_BLOCK = 256;
constexpr int VALU_OB = 4;  // output neurons blocked per pass over the inputs

// scratch slot map (in units of NT floats), computed identically on host and device
struct ValuSlots {
    int ain;                 // 2*J   network input jets
    int z[MAX_LAYERS];       // out_l*J
    int y[MAX_LAYERS];       // out_l*J (hidden layers only)
    int b0, b1;              // maxw*J adjoint ping-pong
    int total;
};

__host__ __device__ inline ValuSlots valu_slots(const KDesc &d, int J)
{
    ValuSlots s;
    int off = 0, maxw = 2;
    s.ain = off; off += 2 * J;
    for (int l = 0; l < d.n_lin; ++l) {
        const int w = d.widths[l + 1];
        if (w > maxw) maxw = w;
        s.z[l] = off; off += w * J;
        s.y[l] = off; if (l < d.n_lin - 1) off += w * J;
    }
    s.b0 = off; off += maxw * J;
    s.b1 = off; off += maxw * J;
    s.total = off;
    return s;
}

template <int EQ, int MODE>
__global__ __launch_bounds__(VALU_BLOCK) void k_valu(KDesc d, CallArgs a, float *scratch)
{
    constexpr int K = EqTraits<EQ>::K, KT = EqTraits<EQ>::KT, J = K + 1 + KT;
    constexpr bool PRECISE = EqTraits<EQ>::PRECISE;      // jet_math.hpp: float64 accumulation in the top two Linear layers
    extern __shared__ float lds_g[];  // [P+2] workgroup partial of (grad, loss sums)
    const int P = d.P;
    if (MODE != MODE_FWD) {
        for (int i = threadIdx.x; i < P + MAXF; i += VALU_BLOCK) lds_g[i] = 0.0f;
        __syncthreads();
    }
    const long long NT = (long long)gridDim.x * VALU_BLOCK;
    const long long tid = (long long)blockIdx.x * VALU_BLOCK + threadIdx.x;
    float *S = scratch + tid;
    const ValuSlots sl = valu_slots(d, J);
    const int lane = threadIdx.x & 63;
    const int nout = d.nout;
    const int nf = EqTraits<EQ>::NF ? EqTraits<EQ>::NF : (nout < MAXOUT ? nout : MAXOUT);
    float lsum[MAXF] = {0.0f, 0.0f, 0.0f};

#define SL(slot) S[(long long)(slot) * NT]

    for (long long base = (long long)blockIdx.x * VALU_BLOCK; base < a.n; base += NT) {
        const long long p = base + threadIdx.x;
        const bool valid = p < a.n;
        const long long pc = valid ? p : a.n - 1;
        const float xv = a.x[pc * a.stride], tv = a.t[pc * a.stride];

        // ---- network input jets (Scale, :39-41) ----
        {
            float ax[J], at[J];
#pragma unroll
            for (int j = 0; j < J; ++j) { ax[j] = 0.0f; at[j] = 0.0f; }
            ax[0] = 2.0f * (xv - d.lbx) / (d.ubx - d.lbx) - 1.0f;
            if (K >= 1) ax[1] = 2.0f / (d.ubx - d.lbx);
            at[0] = 2.0f * (tv - d.lbt) / (d.ubt - d.lbt) - 1.0f;
            if (KT >= 1) at[K + 1] = 2.0f / (d.ubt - d.lbt);
#pragma unroll
            for (int j = 0; j < J; ++j) { SL(sl.ain + j) = ax[j]; SL(sl.ain + J + j) = at[j]; }
        }

        // ---- forward through the layers ----
        double ud[MAXOUT][J];      // PRECISE: the output layer's float64 sums (the residual polynomial takes them as they are)
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o)
#pragma unroll
            for (int j = 0; j < J; ++j) ud[o][j] = 0.0;
        for (int l = 0; l < d.n_lin; ++l) {
            const int nin = d.widths[l], no = d.widths[l + 1];
            const float *W = a.theta + d.woff[l], *B = a.theta + d.boff[l];
            const int in_slot = (l == 0) ? sl.ain : sl.y[l - 1];
            const bool hidden = l < d.n_lin - 1;
            const bool top = PRECISE && l >= 1 && l >= d.n_lin - 2;
            for (int o0 = 0; o0 < no; o0 += VALU_OB) {
                float acc[VALU_OB][J];
                if (PRECISE && top) {
                    double accd[VALU_OB][J];
#pragma unroll
                    for (int ob = 0; ob < VALU_OB; ++ob) {
#pragma unroll
                        for (int j = 0; j < J; ++j) accd[ob][j] = 0.0;
                        accd[ob][0] = (o0 + ob < no) ? (double)B[o0 + ob] : 0.0;
                    }
                    for (int i = 0; i < nin; ++i) {
                        double av[J];
#pragma unroll
                        for (int j = 0; j < J; ++j) av[j] = (double)SL(in_slot + i * J + j);
#pragma unroll
                        for (int ob = 0; ob < VALU_OB; ++ob) {
                            const double w = (o0 + ob < no) ? (double)W[(o0 + ob) * nin + i] : 0.0;
#pragma unroll
                            for (int j = 0; j < J; ++j) accd[ob][j] = fma(w, av[j], accd[ob][j]);
                        }
                    }
#pragma unroll
                    for (int ob = 0; ob < VALU_OB; ++ob)
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            acc[ob][j] = (float)accd[ob][j];
                            if (!hidden && o0 + ob < MAXOUT) ud[(o0 + ob) < MAXOUT ? o0 + ob : 0][j] = accd[ob][j];
                        }
                } else {
#pragma unroll
                for (int ob = 0; ob < VALU_OB; ++ob) {
#pragma unroll
                    for (int j = 0; j < J; ++j) acc[ob][j] = 0.0f;
                    acc[ob][0] = (o0 + ob < no) ? B[o0 + ob] : 0.0f;
                }
                for (int i = 0; i < nin; ++i) {
                    float av[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) av[j] = SL(in_slot + i * J + j);
#pragma unroll
                    for (int ob = 0; ob < VALU_OB; ++ob) {
                        const float w = (o0 + ob < no) ? W[(o0 + ob) * nin + i] : 0.0f;
#pragma unroll
                        for (int j = 0; j < J; ++j) acc[ob][j] = fmaf(w, av[j], acc[ob][j]);
                    }
                }
                }
#pragma unroll
                for (int ob = 0; ob < VALU_OB; ++ob) {
                    if (o0 + ob < no) {
                        const int o = o0 + ob;
#pragma unroll
                        for (int j = 0; j < J; ++j) SL(sl.z[l] + o * J + j) = acc[ob][j];
                        if (hidden) {
                            float y[J], w[K + 1];
                            tanh_jet_fwd<K, KT>(acc[ob], y, w);
#pragma unroll
                            for (int j = 0; j < J; ++j) SL(sl.y[l] + o * J + j) = y[j];
                        }
                    }
                }
            }
        }

        // ---- outputs -> true derivatives -> residual ----
        float D[MAXOUT][J], T[MAXOUT][J];
        const int zl = sl.z[d.n_lin - 1];
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o) {
#pragma unroll
            for (int k = 0; k <= K; ++k) D[o][k] = (o < nout) ? factorial(k) * SL(zl + o * J + k) : 0.0f;
#pragma unroll
            for (int k = 0; k < KT; ++k) T[o][k] = (o < nout) ? factorial(k) * SL(zl + o * J + K + 1 + k) : 0.0f;
        }
        float f[MAXF] = {0.0f, 0.0f, 0.0f};
        if (PRECISE) {
            double Dd[MAXOUT][J], Td[MAXOUT][J], fd[MAXF] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int o = 0; o < MAXOUT; ++o) {
#pragma unroll
                for (int j = 0; j < J; ++j) { Dd[o][j] = 0.0; Td[o][j] = 0.0; }
#pragma unroll
                for (int k = 0; k <= K; ++k) Dd[o][k] = (double)factorial(k) * ud[o][k];
#pragma unroll
                for (int k = 0; k < KT; ++k) Td[o][k] = (double)factorial(k) * ud[o][K + 1 + k];
            }
            Residual<EQ>::eval(Dd, Td, d.coef, nout, fd);
#pragma unroll
            for (int c = 0; c < MAXF; ++c) f[c] = (float)fd[c];
        } else {
            Residual<EQ>::eval(D, T, d.coef, nout, f);
        }

        if (MODE == MODE_FWD) {
            if (valid) {
                if (a.f_out)
                    for (int c = 0; c < nf; ++c) a.f_out[p * nf + c] = f[c];
                if (a.derivs_out)
                    for (int o = 0; o < nout; ++o) {
#pragma unroll
                        for (int k = 0; k <= K; ++k) a.derivs_out[(p * nout + o) * J + k] = D[o][k];
#pragma unroll
                        for (int k = 0; k < KT; ++k) a.derivs_out[(p * nout + o) * J + K + 1 + k] = T[o][k];
                    }
            }
            continue;
        }

        // ---- cotangent of f ----
        float fb[MAXF] = {0.0f, 0.0f, 0.0f};
        for (int c = 0; c < nf; ++c) {
            if (MODE == MODE_LOSS) {
                fb[c] = valid ? 2.0f * a.scale * f[c] : 0.0f;
                lsum[c] += valid ? f[c] * f[c] : 0.0f;
            } else {
                fb[c] = valid ? a.gbar[p * nf + c] : 0.0f;
            }
        }
        float Db[MAXOUT][J], Tb[MAXOUT][J];
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o)
#pragma unroll
            for (int j = 0; j < J; ++j) { Db[o][j] = 0.0f; Tb[o][j] = 0.0f; }
        Residual<EQ>::adj(D, T, d.coef, nout, fb, Db, Tb);
        int cur = sl.b0, nxt = sl.b1;
        for (int o = 0; o < nout; ++o) {
#pragma unroll
            for (int k = 0; k <= K; ++k) SL(cur + o * J + k) = factorial(k) * Db[o][k];
#pragma unroll
            for (int k = 0; k < KT; ++k) SL(cur + o * J + K + 1 + k) = factorial(k) * Tb[o][k];
        }

        // ---- reverse through the layers ----
        for (int l = d.n_lin - 1; l >= 0; --l) {
            const int nin = d.widths[l], no = d.widths[l + 1];
            const float *W = a.theta + d.woff[l];
            const int in_slot = (l == 0) ? sl.ain : sl.y[l - 1];
            // (a) weight and bias gradients: reduce over the wave's 64 points
            for (int o0 = 0; o0 < no; o0 += VALU_OB) {
                float zb[VALU_OB][J];
#pragma unroll
                for (int ob = 0; ob < VALU_OB; ++ob) {
#pragma unroll
                    for (int j = 0; j < J; ++j) zb[ob][j] = (o0 + ob < no) ? SL(cur + (o0 + ob) * J + j) : 0.0f;
                    const float bs = wave_sum(zb[ob][0]);
                    if (lane == 0 && o0 + ob < no) atomicAdd(&lds_g[d.boff[l] + o0 + ob], bs);
                }
                for (int i = 0; i < nin; ++i) {
                    float av[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) av[j] = SL(in_slot + i * J + j);
#pragma unroll
                    for (int ob = 0; ob < VALU_OB; ++ob) {
                        float v = 0.0f;
#pragma unroll
                        for (int j = 0; j < J; ++j) v = fmaf(zb[ob][j], av[j], v);
                        v = wave_sum(v);
                        if (lane == 0 && o0 + ob < no) atomicAdd(&lds_g[d.woff[l] + (o0 + ob) * nin + i], v);
                    }
                }
            }
            if (l == 0) break;
            // (b) adjoint of the inputs, then back through tanh of layer l-1
            for (int i0 = 0; i0 < nin; i0 += VALU_OB) {
                float ab[VALU_OB][J];
#pragma unroll
                for (int ib = 0; ib < VALU_OB; ++ib)
#pragma unroll
                    for (int j = 0; j < J; ++j) ab[ib][j] = 0.0f;
                for (int o = 0; o < no; ++o) {
                    float zb[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) zb[j] = SL(cur + o * J + j);
#pragma unroll
                    for (int ib = 0; ib < VALU_OB; ++ib) {
                        const float w = (i0 + ib < nin) ? W[o * nin + i0 + ib] : 0.0f;
#pragma unroll
                        for (int j = 0; j < J; ++j) ab[ib][j] = fmaf(w, zb[j], ab[ib][j]);
                    }
                }
#pragma unroll
                for (int ib = 0; ib < VALU_OB; ++ib) {
                    if (i0 + ib < nin) {
                        const int i = i0 + ib;
                        // y and w = sech^2 jets are recomputed from the stashed z (w0 comes out of tanh_sech2
                        // directly; re-forming it as 1 - y0^2 from the stashed y would lose its accuracy)
                        float z[J], y[J], w[K + 1], zbn[J];
#pragma unroll
                        for (int j = 0; j < J; ++j) z[j] = SL(sl.z[l - 1] + i * J + j);
                        tanh_jet_fwd<K, KT>(z, y, w);
                        tanh_jet_bwd<K, KT>(z, y, w, ab[ib], zbn);
#pragma unroll
                        for (int j = 0; j < J; ++j) SL(nxt + i * J + j) = zbn[j];
                    }
                }
            }
            const int tmp = cur; cur = nxt; nxt = tmp;
        }
    }
#undef SL

    if (MODE != MODE_FWD) {
        if (MODE == MODE_LOSS) {
            for (int c = 0; c < nf; ++c) {
                const float s = wave_sum(lsum[c]);
                if (lane == 0) atomicAdd(&lds_g[P + c], s);
            }
        }
        __syncthreads();
        float *slab = a.slab + (long long)blockIdx.x * (P + MAXF);
        for (int i = threadIdx.x; i < P + MAXF; i += VALU_BLOCK) slab[i] = lds_g[i];
    }
}

}  // namespace pinn
