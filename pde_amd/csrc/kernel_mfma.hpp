// FP32-MFMA kernels: one wavefront owns a tile of 16 collocation points and
// carries the jets of a whole hidden layer in registers.
//
// Layout (v_mfma_f32_16x16x4_f32, MI355X guide §3): for a tile of 16 points x
// 16 neurons the accumulator of lane l holds rows 4*(l>>4)+r (r=0..3), column
// l&15.  We put NEURONS on rows and POINTS on columns:
//     lane (p = l&15, q = l>>4), register r  <->  neuron 16*T + 4*q + r, point p
// for neuron tile T.  An activation set is  v[T][k][r]  (k = jet component).
//
//  * forward  Z^T[o][p] = sum_i W[o][i] Y^T[i][p]:  A = W fragment, B = the
//    previous layer's accumulator registers used directly as the B operand
//    (k-step (Ti, r) covers input neurons 16Ti + 4q + r; the A fragment is read
//    from LDS in the matching order), so activations never leave registers
//    between layers;
//  * adjoint  Yb^T[i][p] = sum_o W[o][i] Zb^T[o][p]: same trick with W^T;
//  * weight gradient  Wb[o][i] = sum_{p,k} Zb_k[o][p] Y_k[i][p]: the reduction
//    runs over points, which sit on lanes, so both operands are needed
//    transposed: Zb goes through a 1-KiB-per-tile LDS staging image, Y is read
//    back from the stash the forward sweep wrote in [point][neuron] order, whose
//    transposed read is a plain coalesced dword load.
//
// Weights stay resident in LDS (zero-padded [16NT][S] images); the weight
// gradient is accumulated per workgroup in LDS and leaves as one partial slab
// per workgroup (summed in a fixed order by k_reduce_slabs).
// tanh jets / residuals: jet_math.hpp.  Reference semantics as kernel_valu.hpp.
#pragma once
#include "jet_math.hpp"
#include "pinn_common.hpp"

namespace pinn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MFMA_BLOCK = 256;
constexpr int MFMA_WAVES = MFMA_BLOCK / 64;

struct MfmaLayout {   // offsets in floats into dynamic LDS
    int S;            // row stride of the padded hidden weight images
    int wl;           // hidden->hidden weights, image l at wl + l*16NT*S
    int bl;           // hidden->hidden biases,  image l at bl + l*16NT
    int w0x, w0t, b0; // first layer columns and bias, [16NT] each
    int wout;         // last layer [MAXOUT][16NT]
    int bout;         // [MAXOUT]
    int zero_end;     // everything before this is zero-filled then populated
    int gacc;         // [P+2]
    int stage;        // per wave NT*256
    int total;
};

__host__ __device__ inline MfmaLayout mfma_layout(const KDesc &d, int NT)
{
    MfmaLayout L;
    const int NW = 16 * NT;
    const int nhh = d.n_lin - 2;
    L.S = NW + 4;
    int off = 0;
    L.wl = off; off += nhh * NW * L.S;
    L.bl = off; off += nhh * NW;
    L.w0x = off; off += NW;
    L.w0t = off; off += NW;
    L.b0 = off; off += NW;
    L.wout = off; off += MAXOUT * NW;
    L.bout = off; off += 4;
    L.zero_end = off;
    L.gacc = off; off += ((d.P + 2 + 3) / 4) * 4;
    L.stage = off; off += MFMA_WAVES * NT * 256;
    L.total = off;
    return L;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// sum over the 16 lanes of a DPP row (= the 16 points of the tile); every lane gets the sum
__device__ __forceinline__ float row_sum16(float v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true);  // row_half_mirror
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true);  // row_mirror
    v += __int_as_float(t);
    return v;
}

// sum over the four 16-lane rows (same point, different neuron quads)
__device__ __forceinline__ float quad_rows_sum(float v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

#define PINN_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int EQ, int NT, int MODE>
__global__ __launch_bounds__(MFMA_BLOCK, 1) void k_mfma(KDesc d, CallArgs a, float *stash)
{
    constexpr int K = EqTraits<EQ>::K, KT = EqTraits<EQ>::KT, J = K + 1 + KT;
    constexpr int NW = 16 * NT;
    constexpr int TILE = 256;                 // floats of one 16x16 stash/stage image
    constexpr int LAYER_STASH = NT * J * TILE;  // one of (z | y) of one hidden layer
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const MfmaLayout L = mfma_layout(d, NT);
    const int S = L.S;
    const int W = d.widths[1];
    const int n_lin = d.n_lin;
    const int NH = n_lin - 1;                 // tanh layers
    const int nout = d.nout;
    const int nf = (EQ == EQ_NONE) ? nout : (EQ == EQ_GNSS ? 2 : 1);
    const int P = d.P;
    const int tid = threadIdx.x;

    // ---- stage the weights in LDS (zero padded) ----
    for (int i = tid; i < L.zero_end; i += MFMA_BLOCK) lds[i] = 0.0f;
    if (MODE != MODE_FWD)
        for (int i = tid; i < P + 2; i += MFMA_BLOCK) lds[L.gacc + i] = 0.0f;
    __syncthreads();
    for (int l = 1; l <= n_lin - 2; ++l) {
        const float *Wg = a.theta + d.woff[l];
        float *img = lds + L.wl + (l - 1) * NW * S;
        for (int idx = tid; idx < W * W; idx += MFMA_BLOCK) {
            const int o = idx / W, i = idx - o * W;
            img[o * S + i] = Wg[idx];
        }
        for (int o = tid; o < W; o += MFMA_BLOCK) lds[L.bl + (l - 1) * NW + o] = a.theta[d.boff[l] + o];
    }
    for (int n = tid; n < W; n += MFMA_BLOCK) {
        lds[L.w0x + n] = a.theta[d.woff[0] + 2 * n];
        lds[L.w0t + n] = a.theta[d.woff[0] + 2 * n + 1];
        lds[L.b0 + n] = a.theta[d.boff[0] + n];
    }
    for (int idx = tid; idx < nout * W; idx += MFMA_BLOCK) {
        const int o = idx / W, n = idx - o * W;
        lds[L.wout + o * NW + n] = a.theta[d.woff[n_lin - 1] + idx];
    }
    if (tid < nout) lds[L.bout + tid] = a.theta[d.boff[n_lin - 1] + tid];
    __syncthreads();

    const int lane = tid & 63, wv = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const long long wave_g = (long long)blockIdx.x * MFMA_WAVES + wv;
    const long long nwaves = (long long)gridDim.x * MFMA_WAVES;
    const long long ntiles = (a.n + 15) / 16;
    float *const st = stash + wave_g * (long long)(NH * 2 * LAYER_STASH);
    float *const stage = lds + L.stage + wv * NT * TILE;
    float *const gacc = lds + L.gacc;
    const int nat = p * 16 + 4 * q;           // this lane's f32x4 slot inside a [point][neuron] image
    float lsum[MAXOUT] = {0.0f, 0.0f};

    const float sx = 2.0f / (d.ubx - d.lbx), stt = 2.0f / (d.ubt - d.lbt);

    for (long long tile = wave_g; tile < ntiles; tile += nwaves) {
        const long long pt = tile * 16 + p;
        const bool valid = pt < a.n;
        const long long pc = valid ? pt : a.n - 1;
        const float xv = a.x[pc * a.stride], tv = a.t[pc * a.stride];
        const float hx = 2.0f * (xv - d.lbx) / (d.ubx - d.lbx) - 1.0f;
        const float ht = 2.0f * (tv - d.lbt) / (d.ubt - d.lbt) - 1.0f;

        f32x4 A[NT][J];   // activations of the current hidden layer (tanh outputs)

        // ---- first layer (2 -> W), VALU: z0 = wx*hx + wt*ht + b, z1 = wx*sx, zt0 = wt*st ----
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const f32x4 wx = *(const f32x4 *)&lds[L.w0x + 16 * T + 4 * q];
            const f32x4 wt = *(const f32x4 *)&lds[L.w0t + 16 * T + 4 * q];
            const f32x4 b0 = *(const f32x4 *)&lds[L.b0 + 16 * T + 4 * q];
            f32x4 zq[J];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float z[J], y[J], w[K + 1];
#pragma unroll
                for (int j = 0; j < J; ++j) z[j] = 0.0f;
                z[0] = fmaf(wx[r], hx, fmaf(wt[r], ht, b0[r]));
                if (K >= 1) z[1] = wx[r] * sx;
                if (KT >= 1) z[K + 1] = wt[r] * stt;
                tanh_jet_fwd<K, KT>(z, y, w);
#pragma unroll
                for (int j = 0; j < J; ++j) { A[T][j][r] = y[j]; zq[j][r] = z[j]; }
            }
            if (MODE != MODE_FWD) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    *(f32x4 *)&st[(T * J + j) * TILE + nat] = zq[j];
                    *(f32x4 *)&st[LAYER_STASH + (T * J + j) * TILE + nat] = A[T][j];
                }
            }
        }

        // ---- hidden -> hidden layers on the matrix cores ----
        for (int l = 1; l <= n_lin - 2; ++l) {
            const float *img = lds + L.wl + (l - 1) * NW * S;
            const float *bimg = lds + L.bl + (l - 1) * NW;
            float *zs = st + (long long)(l * 2) * LAYER_STASH;
            float *ys = zs + LAYER_STASH;
            f32x4 B[NT][J];
#pragma unroll
            for (int To = 0; To < NT; ++To) {
                f32x4 acc[J];
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                acc[0] = *(const f32x4 *)&bimg[16 * To + 4 * q];
#pragma unroll
                for (int Ti = 0; Ti < NT; ++Ti) {
                    // A fragment: lane (m = p, q) holds W[16To + m][16Ti + 4q + r], r = 0..3
                    const f32x4 wf = *(const f32x4 *)&img[(16 * To + p) * S + 16 * Ti + 4 * q];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (Ti < NT - 1 || 16 * Ti + r < W) {
#pragma unroll
                            for (int j = 0; j < J; ++j) acc[j] = mfma4(wf[r], A[Ti][j][r], acc[j]);
                        }
                    }
                }
                if (MODE != MODE_FWD) {
#pragma unroll
                    for (int j = 0; j < J; ++j) *(f32x4 *)&zs[(To * J + j) * TILE + nat] = acc[j];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z[J], y[J], w[K + 1];
#pragma unroll
                    for (int j = 0; j < J; ++j) z[j] = acc[j][r];
                    tanh_jet_fwd<K, KT>(z, y, w);
#pragma unroll
                    for (int j = 0; j < J; ++j) B[To][j][r] = y[j];
                }
                if (MODE != MODE_FWD) {
#pragma unroll
                    for (int j = 0; j < J; ++j) *(f32x4 *)&ys[(To * J + j) * TILE + nat] = B[To][j];
                }
            }
#pragma unroll
            for (int T = 0; T < NT; ++T)
#pragma unroll
                for (int j = 0; j < J; ++j) A[T][j] = B[T][j];
        }

        // ---- output layer (W -> nout), VALU + cross-row reduction ----
        float D[MAXOUT][J], Tt[MAXOUT][J];
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o) {
            float u[J];
#pragma unroll
            for (int j = 0; j < J; ++j) u[j] = 0.0f;
            if (o < nout) {
#pragma unroll
                for (int T = 0; T < NT; ++T) {
                    const f32x4 wo = *(const f32x4 *)&lds[L.wout + o * NW + 16 * T + 4 * q];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < J; ++j) u[j] = fmaf(wo[r], A[T][j][r], u[j]);
                }
#pragma unroll
                for (int j = 0; j < J; ++j) u[j] = quad_rows_sum(u[j]);
                u[0] += lds[L.bout + o];
            }
#pragma unroll
            for (int k = 0; k <= K; ++k) D[o][k] = factorial(k) * u[k];
#pragma unroll
            for (int k = 0; k < KT; ++k) Tt[o][k] = factorial(k) * u[K + 1 + k];
        }
        float f[MAXOUT] = {0.0f, 0.0f};
        Residual<EQ>::eval(D, Tt, d.coef, nout, f);

        if (MODE == MODE_FWD) {
            if (valid && q == 0) {
                if (a.f_out)
                    for (int c = 0; c < nf; ++c) a.f_out[pt * nf + c] = f[c];
                if (a.derivs_out)
                    for (int o = 0; o < nout; ++o) {
#pragma unroll
                        for (int k = 0; k <= K; ++k) a.derivs_out[(pt * nout + o) * J + k] = D[o][k];
#pragma unroll
                        for (int k = 0; k < KT; ++k) a.derivs_out[(pt * nout + o) * J + K + 1 + k] = Tt[o][k];
                    }
            }
            continue;
        }

        // ---- cotangent of f -> adjoint of the output jets ----
        float fb[MAXOUT] = {0.0f, 0.0f};
        for (int c = 0; c < nf; ++c) {
            if (MODE == MODE_LOSS) {
                fb[c] = valid ? 2.0f * a.scale * f[c] : 0.0f;
                lsum[c] += (valid && q == 0) ? f[c] * f[c] : 0.0f;
            } else {
                fb[c] = valid ? a.gbar[pt * nf + c] : 0.0f;
            }
        }
        float Db[MAXOUT][J], Tb[MAXOUT][J];
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o)
#pragma unroll
            for (int j = 0; j < J; ++j) { Db[o][j] = 0.0f; Tb[o][j] = 0.0f; }
        Residual<EQ>::adj(D, Tt, d.coef, nout, fb, Db, Tb);
        float ub[MAXOUT][J];
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o) {
#pragma unroll
            for (int k = 0; k <= K; ++k) ub[o][k] = factorial(k) * Db[o][k];
#pragma unroll
            for (int k = 0; k < KT; ++k) ub[o][K + 1 + k] = factorial(k) * Tb[o][k];
        }

        // ---- output layer reverse: its weight gradient and the adjoint of the last tanh outputs ----
        f32x4 G[NT][J];
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int j = 0; j < J; ++j) G[T][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o) {
            if (o < nout) {
                const float bs = row_sum16(ub[o][0]);
                if (lane == 0) atomicAdd(&gacc[d.boff[n_lin - 1] + o], bs);
#pragma unroll
                for (int T = 0; T < NT; ++T) {
                    const f32x4 wo = *(const f32x4 *)&lds[L.wout + o * NW + 16 * T + 4 * q];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = 0.0f;
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            v = fmaf(ub[o][j], A[T][j][r], v);
                            G[T][j][r] = fmaf(wo[r], ub[o][j], G[T][j][r]);
                        }
                        v = row_sum16(v);
                        const int n = 16 * T + 4 * q + r;
                        if (p == 0 && n < W) atomicAdd(&gacc[d.woff[n_lin - 1] + o * W + n], v);
                    }
                }
            }
        }

        // ---- reverse sweep over the tanh layers hl = NH-1 .. 0 ----
        // on entry G = adjoint of y_hl; after the tanh reverse G = adjoint of z_hl
        for (int hl = NH - 1; hl >= 0; --hl) {
            const float *zs = st + (long long)(hl * 2) * LAYER_STASH;
            const float *ys = zs + LAYER_STASH;
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                f32x4 zq[J], yq[J];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    zq[j] = *(const f32x4 *)&zs[(T * J + j) * TILE + nat];
                    yq[j] = *(const f32x4 *)&ys[(T * J + j) * TILE + nat];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z[J], y[J], w[K + 1], yb[J], zb[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) { z[j] = zq[j][r]; y[j] = yq[j][r]; yb[j] = G[T][j][r]; }
                    w_from_y<K>(y, w);
                    tanh_jet_bwd<K, KT>(z, y, w, yb, zb);
#pragma unroll
                    for (int j = 0; j < J; ++j) G[T][j][r] = zb[j];
                }
                // bias gradient of the layer that produced z_hl
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float bs = row_sum16(G[T][0][r]);
                    const int n = 16 * T + 4 * q + r;
                    if (p == 0 && n < W) atomicAdd(&gacc[d.boff[hl] + n], bs);
                }
            }
            if (hl == 0) break;

            // Linear layer l = hl maps y_{hl-1} -> z_hl
            const int l = hl;
            const float *img = lds + L.wl + (l - 1) * NW * S;
            const float *yprev = st + (long long)((hl - 1) * 2 + 1) * LAYER_STASH;

            // Phase A: Wb[o][i] += sum_{p,k} Zb_k[o][p] * Y_k[i][p]   (reduction over points)
            {
                f32x4 wacc[NT][NT];
#pragma unroll
                for (int To = 0; To < NT; ++To)
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) wacc[To][Ti] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    float bop[NT][4];
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                        for (int s = 0; s < 4; ++s) bop[Ti][s] = yprev[(Ti * J + j) * TILE + 64 * s + lane];
#pragma unroll
                    for (int To = 0; To < NT; ++To) *(f32x4 *)&stage[To * TILE + nat] = G[To][j];
                    PINN_LDS_FENCE();
                    float aop[NT][4];
#pragma unroll
                    for (int To = 0; To < NT; ++To)
#pragma unroll
                        for (int s = 0; s < 4; ++s) aop[To][s] = stage[To * TILE + 64 * s + lane];
                    PINN_LDS_FENCE();
#pragma unroll
                    for (int To = 0; To < NT; ++To)
#pragma unroll
                        for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                            for (int s = 0; s < 4; ++s) wacc[To][Ti] = mfma4(aop[To][s], bop[Ti][s], wacc[To][Ti]);
                }
                // accumulator (To,Ti): lane (i = p, q) holds Wb[16To + 4q + r][16Ti + p]
#pragma unroll
                for (int To = 0; To < NT; ++To)
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int o = 16 * To + 4 * q + r, i = 16 * Ti + p;
                            if (o < W && i < W) atomicAdd(&gacc[d.woff[l] + o * W + i], wacc[To][Ti][r]);
                        }
            }

            // Phase B: Yb^T[i][p] = sum_o W[o][i] Zb^T[o][p]
            f32x4 H[NT][J];
#pragma unroll
            for (int Ti = 0; Ti < NT; ++Ti) {
                f32x4 acc[J];
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int To = 0; To < NT; ++To) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (To < NT - 1 || 16 * To + r < W) {
                            // A fragment: lane (m = p, q) holds W[16To + 4q + r][16Ti + m]
                            const float wf = img[(16 * To + 4 * q + r) * S + 16 * Ti + p];
#pragma unroll
                            for (int j = 0; j < J; ++j) acc[j] = mfma4(wf, G[To][j][r], acc[j]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < J; ++j) H[Ti][j] = acc[j];
            }
#pragma unroll
            for (int T = 0; T < NT; ++T)
#pragma unroll
                for (int j = 0; j < J; ++j) G[T][j] = H[T][j];
        }

        // ---- first layer gradients from G = adjoint of z_0 ----
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g0 = G[T][0][r];
                float gx = g0 * hx, gt = g0 * ht;
                if (K >= 1) gx = fmaf(G[T][K >= 1 ? 1 : 0][r], sx, gx);
                if (KT >= 1) gt = fmaf(G[T][KT >= 1 ? K + 1 : 0][r], stt, gt);
                gx = row_sum16(gx);
                gt = row_sum16(gt);
                const int n = 16 * T + 4 * q + r;
                if (p == 0 && n < W) {
                    atomicAdd(&gacc[d.woff[0] + 2 * n], gx);
                    atomicAdd(&gacc[d.woff[0] + 2 * n + 1], gt);
                }
            }
    }

    if (MODE != MODE_FWD) {
        if (MODE == MODE_LOSS) {
            for (int c = 0; c < nf; ++c) {
                const float s = wave_sum(lsum[c]);
                if (lane == 0) atomicAdd(&gacc[P + c], s);
            }
        }
        __syncthreads();
        float *slab = a.slab + (long long)blockIdx.x * (P + 2);
        for (int i = tid; i < P + 2; i += MFMA_BLOCK) slab[i] = gacc[i];
    }
}

// ------------------------------------------------------------------ host side
struct MfmaPlan {
    int NT = 0;
    int grid = 0;
    float *stash = nullptr;
    size_t stash_bytes = 0;
    float *slab = nullptr;
    size_t slab_bytes = 0;
};

inline int mfma_nt(const KDesc &kd) { return (kd.widths[1] + 15) / 16; }

inline bool mfma_supported(const KDesc &kd)
{
    if (kd.n_lin < 2) return false;
    const int W = kd.widths[1];
    for (int l = 1; l < kd.n_lin; ++l)
        if (kd.widths[l] != W) return false;
    const int NT = mfma_nt(kd);
    if (NT < 1 || NT > 4) return false;
    const MfmaLayout L = mfma_layout(kd, NT);
    return (size_t)L.total * sizeof(float) <= 160 * 1024;
}

inline int mfma_plan_create(MfmaPlan &pl, const KDesc &kd, int n_cu)
{
    pl.NT = mfma_nt(kd);
    pl.grid = n_cu;
    return 0;
}

inline void mfma_plan_destroy(MfmaPlan &pl)
{
    if (pl.stash) (void)hipFree(pl.stash);
    if (pl.slab) (void)hipFree(pl.slab);
    pl.stash = pl.slab = nullptr;
}

template <int EQ, int NT, int MODE>
int launch_mfma_t(MfmaPlan &pl, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    const MfmaLayout L = mfma_layout(kd, NT);
    const size_t ldsb = (size_t)L.total * sizeof(float);
    auto kern = k_mfma<EQ, NT, MODE>;
    if (ldsb > 48 * 1024) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) {
            (void)hipGetLastError();
            return -3;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MFMA_BLOCK), ldsb, st, kd, a, pl.stash);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int EQ, int MODE>
int launch_mfma_nt(MfmaPlan &pl, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    switch (pl.NT) {
    case 1: return launch_mfma_t<EQ, 1, MODE>(pl, kd, a, grid, st);
    case 2: return launch_mfma_t<EQ, 2, MODE>(pl, kd, a, grid, st);
    case 3: return launch_mfma_t<EQ, 3, MODE>(pl, kd, a, grid, st);
    case 4: return launch_mfma_t<EQ, 4, MODE>(pl, kd, a, grid, st);
    default: return -2;
    }
}

template <int MODE>
int launch_mfma_eq(MfmaPlan &pl, int eq, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    switch (eq) {
    case EQ_NONE:    return launch_mfma_nt<EQ_NONE, MODE>(pl, kd, a, grid, st);
    case EQ_BURGERS: return launch_mfma_nt<EQ_BURGERS, MODE>(pl, kd, a, grid, st);
    case EQ_KDV5:    return launch_mfma_nt<EQ_KDV5, MODE>(pl, kd, a, grid, st);
    case EQ_KDV7:    return launch_mfma_nt<EQ_KDV7, MODE>(pl, kd, a, grid, st);
    case EQ_CH:      return launch_mfma_nt<EQ_CH, MODE>(pl, kd, a, grid, st);
    case EQ_GNSS:    return launch_mfma_nt<EQ_GNSS, MODE>(pl, kd, a, grid, st);
    default: return -1;
    }
}

}  // namespace pinn
