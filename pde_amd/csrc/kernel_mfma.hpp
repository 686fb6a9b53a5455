// FP32-MFMA kernels: one wavefront owns a tile of 16 collocation points and
// carries the jets of a whole hidden layer in registers.
//
// Layout (v_mfma_f32_16x16x4_f32, MI355X guide §3): for a 16x16 tile the
// accumulator of lane l holds rows 4*(l>>4)+r (r=0..3), column l&15.  NEURONS sit
// on rows, POINTS on columns:
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
// Widths that are 16*NT + R with 1 <= R <= 4 (the reference's width 50 = 48 + 2)
// do not pad up to the next tile: the R "tail" neurons are carried per lane on
// the VALU (replicated over the four lane quads), which removes the padding
// MFMAs (width 50: 972 instead of 1584 MFMAs per layer and tile) and a quarter
// of the register footprint.
//
// Weights stay resident in LDS; the weight gradient is accumulated per
// workgroup in LDS and leaves as one partial slab per workgroup (summed in a
// fixed order by k_reduce_slabs).  tanh jets / residuals: jet_math.hpp.
// Reference semantics as kernel_valu.hpp.
#pragma once
#include "jet_math.hpp"
#include "pinn_common.hpp"

namespace pinn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MFMA_BLOCK = 256;
constexpr int MFMA_WAVES = MFMA_BLOCK / 64;

template <int W> struct WidthCfg {
    static constexpr int REM = W % 16;
    static constexpr bool TAIL = (REM != 0 && REM <= 4 && W > 16);
    static constexpr int NT = TAIL ? W / 16 : (W + 15) / 16;  // MFMA neuron tiles
    static constexpr int R = TAIL ? REM : 0;                   // tail neurons on the VALU
    static constexpr int RR = R > 0 ? R : 1;                   // array extent
    static constexpr int NM = 16 * NT;                         // neurons covered by tiles
    static constexpr int S = NM + 4;                           // row stride of the main weight image
    // one hidden->hidden layer image in LDS (floats)
    static constexpr int O_BIAS = NM * S;          // [NM]
    static constexpr int O_WTC = O_BIAS + NM;      // [R][NM]  W[o][NM+e], o < NM
    static constexpr int O_WTR = O_WTC + RR * NM;  // [R][NM]  W[NM+e][i], i < NM
    static constexpr int O_WTT = O_WTR + RR * NM;  // [4][4]   W[NM+e][NM+e']
    static constexpr int O_BT = O_WTT + 16;        // [4]      b[NM+e]
    static constexpr int HH = O_BT + 4;
    // first / last layer block
    static constexpr int O_W0X = 0, O_W0T = NM, O_B0 = 2 * NM;     // [NM] each
    static constexpr int O_W0TAIL = 3 * NM;                        // [3][4]: w0x, w0t, b0 of the tail neurons
    static constexpr int O_WOUT = O_W0TAIL + 12;                   // [MAXOUT][NM]
    static constexpr int O_WOUTT = O_WOUT + MAXOUT * NM;           // [MAXOUT][4]
    static constexpr int O_BOUT = O_WOUTT + MAXOUT * 4;            // [4]
    static constexpr int IO = O_BOUT + 4;
};

struct MfmaLayout {   // offsets in floats into dynamic LDS
    int hh;           // hidden->hidden images, layer l (1..n_lin-2) at hh + (l-1)*HH
    int io;           // first/last layer block
    int zero_end;     // everything before this is zero-filled then populated
    int gacc;         // [P+2]
    int stage;        // per wave NT*256
    int total;
};

template <int W> __host__ __device__ inline MfmaLayout mfma_layout(const KDesc &d)
{
    typedef WidthCfg<W> C;
    MfmaLayout L;
    int off = 0;
    L.hh = off; off += (d.n_lin - 2) * C::HH;
    L.io = off; off += C::IO;
    L.zero_end = off;
    L.gacc = off; off += ((d.P + 2 + 3) / 4) * 4;
    L.stage = off; off += MFMA_WAVES * C::NT * 256;
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

// sum over the four 16-lane rows (same point, different neuron quads); every lane gets the sum
__device__ __forceinline__ float quad_rows_sum(float v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Park a value in the accumulator half of the unified register file.  VALU instructions can only
// address the 256 architectural VGPRs; the big per-layer activation sets are touched by MFMAs (which
// read/write AGPRs directly) most of the time, so pinning them there keeps the VALU working set small
// (no scratch spills); a VALU use costs one v_accvgpr_read.
__device__ __forceinline__ void park(f32x4 &v) { asm volatile("" : "+a"(v)); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Stash accesses go through a buffer descriptor: one shared per-lane VGPR offset plus a SCALAR offset
// per access.  With plain pointers the compiler materialises one loop-invariant VGPR address per
// access (the constant part exceeds the 13-bit immediate), hoists all of them out of the tile loop and
// spills them.
__device__ __forceinline__ void sb_store4(__amdgpu_buffer_rsrc_t rs, int voff, int soff_floats, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, soff_floats * 4, 0);
}
__device__ __forceinline__ void sb_store1(__amdgpu_buffer_rsrc_t rs, int voff, int soff_floats, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rs, voff, soff_floats * 4, 0);
}
__device__ __forceinline__ f32x4 sb_load4(__amdgpu_buffer_rsrc_t rs, int voff, int soff_floats)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff_floats * 4, 0));
}
__device__ __forceinline__ float sb_load1(__amdgpu_buffer_rsrc_t rs, int voff, int soff_floats)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff_floats * 4, 0));
}

#define PINN_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PINN_SB() __builtin_amdgcn_sched_barrier(0)

// MFMA block of the forward GEMM for one output tile: acc[j] = bias + sum_{Ti,r} W-frag * A[Ti][j][r]
template <int W, int J>
__device__ __forceinline__ void fwd_mma_tile(const float *img, int To, int p, int q,
                                             const f32x4 (&A)[WidthCfg<W>::NT][J], f32x4 (&acc)[J])
{
    typedef WidthCfg<W> C;
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    acc[0] = *(const f32x4 *)&img[C::O_BIAS + 16 * To + 4 * q];
#pragma unroll
    for (int Ti = 0; Ti < C::NT; ++Ti) {
        // A fragment: lane (m = p, q) holds W[16To + m][16Ti + 4q + r], r = 0..3
        const f32x4 wf = *(const f32x4 *)&img[(16 * To + p) * C::S + 16 * Ti + 4 * q];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * Ti + r < W) {     // compile-time: k-steps that only hit padding are dropped
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] = mfma4(wf[r], A[Ti][j][r], acc[j]);
            }
        }
    }
}

template <int EQ, int W, int MODE>
__global__ __launch_bounds__(MFMA_BLOCK, 1) void k_mfma(KDesc d, CallArgs a, float *stash)
{
    typedef WidthCfg<W> C;
    constexpr int NT = C::NT, R = C::R, RR = C::RR, NM = C::NM, S = C::S;
    constexpr bool TAIL = C::TAIL;
    constexpr int K = EqTraits<EQ>::K, KT = EqTraits<EQ>::KT, J = K + 1 + KT;
    constexpr int TILE = 256;                      // floats of one 16x16 stash/stage image
    constexpr int ZY = NT * J * TILE;              // main z (or y) images of one hidden layer
    constexpr int TS = ((RR * J * 16 + 63) / 64) * 64;  // tail z (or y) of one hidden layer: [e][k][p]
    constexpr int LAYER_STASH = 2 * ZY + 2 * TS;   // z main | y main | z tail | y tail
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const MfmaLayout L = mfma_layout<W>(d);
    const int n_lin = d.n_lin;
    const int NH = n_lin - 1;                      // tanh layers
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
        float *img = lds + L.hh + (l - 1) * C::HH;
        for (int idx = tid; idx < W * W; idx += MFMA_BLOCK) {
            const int o = idx / W, i = idx - o * W;
            const float v = Wg[idx];
            if (o < NM && i < NM) img[o * S + i] = v;
            else if (o < NM) img[C::O_WTC + (i - NM) * NM + o] = v;
            else if (i < NM) img[C::O_WTR + (o - NM) * NM + i] = v;
            else img[C::O_WTT + (o - NM) * 4 + (i - NM)] = v;
        }
        for (int o = tid; o < W; o += MFMA_BLOCK) {
            const float v = a.theta[d.boff[l] + o];
            if (o < NM) img[C::O_BIAS + o] = v;
            else img[C::O_BT + (o - NM)] = v;
        }
    }
    {
        float *iow = lds + L.io;
        for (int n = tid; n < W; n += MFMA_BLOCK) {
            const float wx = a.theta[d.woff[0] + 2 * n], wt = a.theta[d.woff[0] + 2 * n + 1];
            const float b = a.theta[d.boff[0] + n];
            if (n < NM) { iow[C::O_W0X + n] = wx; iow[C::O_W0T + n] = wt; iow[C::O_B0 + n] = b; }
            else { iow[C::O_W0TAIL + (n - NM)] = wx; iow[C::O_W0TAIL + 4 + (n - NM)] = wt; iow[C::O_W0TAIL + 8 + (n - NM)] = b; }
        }
        for (int idx = tid; idx < nout * W; idx += MFMA_BLOCK) {
            const int o = idx / W, n = idx - o * W;
            const float v = a.theta[d.woff[n_lin - 1] + idx];
            if (n < NM) iow[C::O_WOUT + o * NM + n] = v;
            else iow[C::O_WOUTT + o * 4 + (n - NM)] = v;
        }
        if (tid < nout) iow[C::O_BOUT + tid] = a.theta[d.boff[n_lin - 1] + tid];
    }
    __syncthreads();

    // wave id through readfirstlane: makes every per-wave base pointer provably wave-uniform, so the
    // stash/stage accesses use scalar-base + lane-offset addressing instead of per-lane 64-bit pointers
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const long long wave_g = (long long)blockIdx.x * MFMA_WAVES + wv;
    const long long nwaves = (long long)gridDim.x * MFMA_WAVES;
    const long long ntiles = (a.n + 15) / 16;
    float *const st = stash + wave_g * (long long)(NH * LAYER_STASH);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(st, 0, NH * LAYER_STASH * 4, 0x00020000);
    float *const stage = lds + L.stage + wv * NT * TILE;
    float *const gacc = lds + L.gacc;
    const float *const io = lds + L.io;
    const int nat = p * 16 + 4 * q;               // this lane's f32x4 slot inside a [point][neuron] image
    const int vnat = nat * 4, vlane = lane * 4, vp = p * 4;   // per-lane byte offsets for the stash buffer ops
    float lsum[MAXOUT] = {0.0f, 0.0f};

    const float sx = 2.0f / (d.ubx - d.lbx), stt = 2.0f / (d.ubt - d.lbt);

    for (long long tile = wave_g; tile < ntiles; tile += nwaves) {
        const long long pt = tile * 16 + p;
        const bool valid = pt < a.n;
        const long long pc = valid ? pt : a.n - 1;
        const float xv = a.x[pc * a.stride], tv = a.t[pc * a.stride];
        const float hx = 2.0f * (xv - d.lbx) / (d.ubx - d.lbx) - 1.0f;
        const float ht = 2.0f * (tv - d.lbt) / (d.ubt - d.lbt) - 1.0f;

        f32x4 A[NT][J];        // tanh outputs of the current hidden layer, tiled neurons
        float At[RR][J];       // ... tail neurons (same value in all four lane quads)

        // ---- first layer (2 -> W), VALU: z0 = wx*hx + wt*ht + b, z1 = wx*sx, zt0 = wt*st ----
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const f32x4 wx = *(const f32x4 *)&io[C::O_W0X + 16 * T + 4 * q];
            const f32x4 wt = *(const f32x4 *)&io[C::O_W0T + 16 * T + 4 * q];
            const f32x4 b0 = *(const f32x4 *)&io[C::O_B0 + 16 * T + 4 * q];
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
                    sb_store4(srs, vnat, (T * J + j) * TILE, zq[j]);
                    sb_store4(srs, vnat, ZY + (T * J + j) * TILE, A[T][j]);
                }
            }
#pragma unroll
            for (int j = 0; j < J; ++j) park(A[T][j]);
        }
#pragma unroll
        for (int e = 0; e < RR; ++e)
#pragma unroll
            for (int j = 0; j < J; ++j) At[e][j] = 0.0f;
        if (TAIL) {
#pragma unroll
            for (int e = 0; e < R; ++e) {
                float z[J], w[K + 1];
#pragma unroll
                for (int j = 0; j < J; ++j) z[j] = 0.0f;
                const float wx = io[C::O_W0TAIL + e], wt = io[C::O_W0TAIL + 4 + e], b0 = io[C::O_W0TAIL + 8 + e];
                z[0] = fmaf(wx, hx, fmaf(wt, ht, b0));
                if (K >= 1) z[1] = wx * sx;
                if (KT >= 1) z[K + 1] = wt * stt;
                tanh_jet_fwd<K, KT>(z, At[e], w);
                if (MODE != MODE_FWD && q == 0) {
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        sb_store1(srs, vp, 2 * ZY + (e * J + j) * 16, z[j]);
                        sb_store1(srs, vp, 2 * ZY + TS + (e * J + j) * 16, At[e][j]);
                    }
                }
            }
        }

        // ---- hidden -> hidden layers on the matrix cores ----
        for (int l = 1; l <= n_lin - 2; ++l) {
            const float *img = lds + L.hh + (l - 1) * C::HH;
            const int zs = l * LAYER_STASH;   // stash offset (floats) of this layer
            f32x4 B[NT][J];
            float Bt[RR][J];
#pragma unroll
            for (int e = 0; e < RR; ++e)
#pragma unroll
                for (int j = 0; j < J; ++j) Bt[e][j] = 0.0f;
            // software pipeline: the MFMA block of output tile To+1 is issued before the VALU work
            // (tail neurons, tanh jets) of tile To, so the matrix pipe runs under the vector work
            f32x4 acc[2][J];
            fwd_mma_tile<W, J>(img, 0, p, q, A, acc[0]);
            if (TAIL) {
                // tail outputs: every lane sums over its 4*NT tiled inputs, then across the lane quads
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    float z[J], w[K + 1];
#pragma unroll
                    for (int j = 0; j < J; ++j) z[j] = 0.0f;
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) {
                        const f32x4 wr = *(const f32x4 *)&img[C::O_WTR + e * NM + 16 * Ti + 4 * q];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < J; ++j) z[j] = fmaf(wr[r], A[Ti][j][r], z[j]);
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) z[j] = quad_rows_sum(z[j]);
                    z[0] += img[C::O_BT + e];
#pragma unroll
                    for (int e2 = 0; e2 < R; ++e2) {
                        const float wtt = img[C::O_WTT + e * 4 + e2];
#pragma unroll
                        for (int j = 0; j < J; ++j) z[j] = fmaf(wtt, At[e2][j], z[j]);
                    }
                    tanh_jet_fwd<K, KT>(z, Bt[e], w);
                    if (MODE != MODE_FWD && q == 0) {
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            sb_store1(srs, vp, zs + 2 * ZY + (e * J + j) * 16, z[j]);
                            sb_store1(srs, vp, zs + 2 * ZY + TS + (e * J + j) * 16, Bt[e][j]);
                        }
                    }
                }
            }
#pragma unroll
            for (int To = 0; To < NT; ++To) {
                if (To + 1 < NT) fwd_mma_tile<W, J>(img, To + 1, p, q, A, acc[(To + 1) & 1]);
                f32x4 (&ac)[J] = acc[To & 1];
                if (TAIL) {
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        const f32x4 wc = *(const f32x4 *)&img[C::O_WTC + e * NM + 16 * To + 4 * q];
#pragma unroll
                        for (int j = 0; j < J; ++j) ac[j] += wc * At[e][j];
                    }
                }
                if (MODE != MODE_FWD) {
#pragma unroll
                    for (int j = 0; j < J; ++j) sb_store4(srs, vnat, zs + (To * J + j) * TILE, ac[j]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z[J], y[J], w[K + 1];
#pragma unroll
                    for (int j = 0; j < J; ++j) z[j] = ac[j][r];
                    tanh_jet_fwd<K, KT>(z, y, w);
#pragma unroll
                    for (int j = 0; j < J; ++j) B[To][j][r] = y[j];
                }
                if (MODE != MODE_FWD) {
#pragma unroll
                    for (int j = 0; j < J; ++j) sb_store4(srs, vnat, zs + ZY + (To * J + j) * TILE, B[To][j]);
                }
#pragma unroll
                for (int j = 0; j < J; ++j) park(B[To][j]);
            }
#pragma unroll
            for (int T = 0; T < NT; ++T)
#pragma unroll
                for (int j = 0; j < J; ++j) A[T][j] = B[T][j];
#pragma unroll
            for (int e = 0; e < RR; ++e)
#pragma unroll
                for (int j = 0; j < J; ++j) At[e][j] = Bt[e][j];
        }

        // ---- output layer (W -> nout), VALU + cross-quad reduction ----
        float D[MAXOUT][J], Tt[MAXOUT][J];
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o) {
            float u[J];
#pragma unroll
            for (int j = 0; j < J; ++j) u[j] = 0.0f;
            if (o < nout) {
#pragma unroll
                for (int T = 0; T < NT; ++T) {
                    const f32x4 wo = *(const f32x4 *)&io[C::O_WOUT + o * NM + 16 * T + 4 * q];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < J; ++j) u[j] = fmaf(wo[r], A[T][j][r], u[j]);
                }
#pragma unroll
                for (int j = 0; j < J; ++j) u[j] = quad_rows_sum(u[j]);
                if (TAIL) {
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        const float wo = io[C::O_WOUTT + o * 4 + e];
#pragma unroll
                        for (int j = 0; j < J; ++j) u[j] = fmaf(wo, At[e][j], u[j]);
                    }
                }
                u[0] += io[C::O_BOUT + o];
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
        float Gt[RR][J];
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int j = 0; j < J; ++j) G[T][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < RR; ++e)
#pragma unroll
            for (int j = 0; j < J; ++j) Gt[e][j] = 0.0f;
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o) {
            if (o < nout) {
                float *gw = gacc + d.woff[n_lin - 1] + o * W;
                const float bs = row_sum16(ub[o][0]);
                if (lane == 0) atomicAdd(&gacc[d.boff[n_lin - 1] + o], bs);
#pragma unroll
                for (int T = 0; T < NT; ++T) {
                    const f32x4 wo = *(const f32x4 *)&io[C::O_WOUT + o * NM + 16 * T + 4 * q];
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
                        if (p == 0 && n < W) atomicAdd(&gw[n], v);
                    }
                }
                if (TAIL) {
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        const float wo = io[C::O_WOUTT + o * 4 + e];
                        float v = 0.0f;
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            v = fmaf(ub[o][j], At[e][j], v);
                            Gt[e][j] = fmaf(wo, ub[o][j], Gt[e][j]);
                        }
                        v = row_sum16(v);
                        if (lane == 0) atomicAdd(&gw[NM + e], v);
                    }
                }
            }
        }

#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int j = 0; j < J; ++j) park(G[T][j]);

        // ---- reverse sweep over the tanh layers hl = NH-1 .. 0 ----
        // on entry G/Gt = adjoint of y_hl; after the tanh reverse they are the adjoint of z_hl.
        // Ct carries the tail rows of Zb of the layer above (hl+1) whose weight gradient against the
        // tiled y_hl is formed here, where y_hl is loaded anyway.
        float Ct[RR][J];
#pragma unroll
        for (int e = 0; e < RR; ++e)
#pragma unroll
            for (int j = 0; j < J; ++j) Ct[e][j] = 0.0f;

        f32x4 zq[2][J], yq[2][J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            zq[0][j] = sb_load4(srs, vnat, (NH - 1) * LAYER_STASH + (0 * J + j) * TILE);
            yq[0][j] = sb_load4(srs, vnat, (NH - 1) * LAYER_STASH + ZY + (0 * J + j) * TILE);
        }
        for (int hl = NH - 1; hl >= 0; --hl) {
            const int zs = hl * LAYER_STASH;
            const bool above = hl < NH - 1;     // Linear layer hl+1 (hidden->hidden) sits above
            float *gw_above = gacc + d.woff[above ? hl + 1 : 0];
            // tails first (their stash rows are tiny broadcast loads)
            if (TAIL) {
                float yt[RR][J];
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    float z[J], w[K + 1], zb[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        z[j] = sb_load1(srs, vp, zs + 2 * ZY + (e * J + j) * 16);
                        yt[e][j] = sb_load1(srs, vp, zs + 2 * ZY + TS + (e * J + j) * 16);
                    }
                    w_from_y<K>(yt[e], w);
                    tanh_jet_bwd<K, KT>(z, yt[e], w, Gt[e], zb);
#pragma unroll
                    for (int j = 0; j < J; ++j) Gt[e][j] = zb[j];
                    const float bs = row_sum16(zb[0]);
                    if (lane == 0) atomicAdd(&gacc[d.boff[hl] + NM + e], bs);
                }
                if (above) {
                    // tail x tail block of Wb_{hl+1}
#pragma unroll
                    for (int e = 0; e < R; ++e)
#pragma unroll
                        for (int e2 = 0; e2 < R; ++e2) {
                            float v = 0.0f;
#pragma unroll
                            for (int j = 0; j < J; ++j) v = fmaf(Ct[e][j], yt[e2][j], v);
                            v = row_sum16(v);
                            if (lane == 0) atomicAdd(&gw_above[(NM + e) * W + NM + e2], v);
                        }
                }
            }
            // tiled neurons: z/y tiles of this layer come back from the stash as f32x4 rows; tile T+1 is
            // in flight while tile T is processed, and tile 0 of the NEXT layer down is requested before
            // the MFMA phases below so that its latency hides under them.
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                if (T + 1 < NT) {
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        zq[(T + 1) & 1][j] = sb_load4(srs, vnat, zs + ((T + 1) * J + j) * TILE);
                        yq[(T + 1) & 1][j] = sb_load4(srs, vnat, zs + ZY + ((T + 1) * J + j) * TILE);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z[J], y[J], w[K + 1], yb[J], zb[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) { z[j] = zq[T & 1][j][r]; y[j] = yq[T & 1][j][r]; yb[j] = G[T][j][r]; }
                    w_from_y<K>(y, w);
                    tanh_jet_bwd<K, KT>(z, y, w, yb, zb);
#pragma unroll
                    for (int j = 0; j < J; ++j) G[T][j][r] = zb[j];
                    const int n = 16 * T + 4 * q + r;
                    // bias gradient of the layer that produced z_hl
                    const float bs = row_sum16(zb[0]);
                    if (p == 0 && n < W) atomicAdd(&gacc[d.boff[hl] + n], bs);
                    if (TAIL && above) {
                        // tail rows of Wb_{hl+1} against this tile of y_hl
#pragma unroll
                        for (int e = 0; e < R; ++e) {
                            float v = 0.0f;
#pragma unroll
                            for (int j = 0; j < J; ++j) v = fmaf(Ct[e][j], y[j], v);
                            v = row_sum16(v);
                            if (p == 0) atomicAdd(&gw_above[(NM + e) * W + n], v);
                        }
                    }
                    PINN_SB();
                }
#pragma unroll
                for (int j = 0; j < J; ++j) park(G[T][j]);
            }
            if (hl > 0) {
                const int zn = (hl - 1) * LAYER_STASH;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    zq[0][j] = sb_load4(srs, vnat, zn + (0 * J + j) * TILE);
                    yq[0][j] = sb_load4(srs, vnat, zn + ZY + (0 * J + j) * TILE);
                }
            }
            if (hl == 0) break;

            // Linear layer l = hl maps y_{hl-1} -> z_hl
            const int l = hl;
            const float *img = lds + L.hh + (l - 1) * C::HH;
            const int yprev = (hl - 1) * LAYER_STASH + ZY;
            float *gw = gacc + d.woff[l];

            // Phase A: Wb[o][i] += sum_{p,k} Zb_k[o][p] * Y_k[i][p]   (reduction over points)
            {
                f32x4 wacc[NT][NT];
#pragma unroll
                for (int To = 0; To < NT; ++To)
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) wacc[To][Ti] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                float bop[2][NT][4];
#pragma unroll
                for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                    for (int s = 0; s < 4; ++s) bop[0][Ti][s] = sb_load1(srs, vlane, yprev + (Ti * J + 0) * TILE + 64 * s);
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    if (j + 1 < J) {
#pragma unroll
                        for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                bop[(j + 1) & 1][Ti][s] = sb_load1(srs, vlane, yprev + (Ti * J + j + 1) * TILE + 64 * s);
                    }
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
                            for (int s = 0; s < 4; ++s)
                                wacc[To][Ti] = mfma4(aop[To][s], bop[j & 1][Ti][s], wacc[To][Ti]);
                    PINN_SB();
                }
                // accumulator (To,Ti): lane (i = p, q) holds Wb[16To + 4q + r][16Ti + p]
                float *gl = gw + 4 * q * W + p;
#pragma unroll
                for (int To = 0; To < NT; ++To)
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int o = 16 * To + 4 * q + r, i = 16 * Ti + p;
                            if (TAIL || (o < W && i < W)) atomicAdd(&gl[(16 * To + r) * W + 16 * Ti], wacc[To][Ti][r]);
                        }
                if (TAIL) {
                    // tiled rows x tail columns: Wb[o][NM+e] = sum_{p,k} Zb_k[o][p] * ytail_k[e][p]
                    float yp[RR][J];
#pragma unroll
                    for (int e = 0; e < R; ++e)
#pragma unroll
                        for (int j = 0; j < J; ++j) yp[e][j] = sb_load1(srs, vp, yprev + ZY + TS + (e * J + j) * 16);
#pragma unroll
                    for (int To = 0; To < NT; ++To)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int e = 0; e < R; ++e) {
                                float v = 0.0f;
#pragma unroll
                                for (int j = 0; j < J; ++j) v = fmaf(G[To][j][r], yp[e][j], v);
                                v = row_sum16(v);
                                if (p == 0) atomicAdd(&gw[(16 * To + 4 * q + r) * W + NM + e], v);
                            }
                }
            }

            // Phase B: Yb^T[i][p] = sum_o W[o][i] Zb^T[o][p]
            f32x4 H[NT][J];
            float Ht[RR][J];
#pragma unroll
            for (int e = 0; e < RR; ++e)
#pragma unroll
                for (int j = 0; j < J; ++j) Ht[e][j] = 0.0f;
            if (TAIL) {
                // adjoint of the tail inputs: sum over tiled outputs (cross-quad) + tail outputs
#pragma unroll
                for (int e2 = 0; e2 < R; ++e2) {
                    float h[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) h[j] = 0.0f;
#pragma unroll
                    for (int To = 0; To < NT; ++To) {
                        const f32x4 wc = *(const f32x4 *)&img[C::O_WTC + e2 * NM + 16 * To + 4 * q];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < J; ++j) h[j] = fmaf(wc[r], G[To][j][r], h[j]);
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) h[j] = quad_rows_sum(h[j]);
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        const float wtt = img[C::O_WTT + e * 4 + e2];
#pragma unroll
                        for (int j = 0; j < J; ++j) h[j] = fmaf(wtt, Gt[e][j], h[j]);
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) Ht[e2][j] = h[j];
                }
            }
#pragma unroll
            for (int Ti = 0; Ti < NT; ++Ti) {
                f32x4 acc[J];
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int To = 0; To < NT; ++To) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (16 * To + r < W) {
                            // A fragment: lane (m = p, q) holds W[16To + 4q + r][16Ti + m]
                            const float wf = img[(16 * To + 4 * q + r) * S + 16 * Ti + p];
#pragma unroll
                            for (int j = 0; j < J; ++j) acc[j] = mfma4(wf, G[To][j][r], acc[j]);
                        }
                    }
                }
                if (TAIL) {
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        const f32x4 wr = *(const f32x4 *)&img[C::O_WTR + e * NM + 16 * Ti + 4 * q];
#pragma unroll
                        for (int j = 0; j < J; ++j) acc[j] += wr * Gt[e][j];
                    }
                }
#pragma unroll
                for (int j = 0; j < J; ++j) { H[Ti][j] = acc[j]; park(H[Ti][j]); }
                PINN_SB();
            }
#pragma unroll
            for (int T = 0; T < NT; ++T)
#pragma unroll
                for (int j = 0; j < J; ++j) G[T][j] = H[T][j];
#pragma unroll
            for (int e = 0; e < RR; ++e)
#pragma unroll
                for (int j = 0; j < J; ++j) { Ct[e][j] = Gt[e][j]; Gt[e][j] = Ht[e][j]; }
        }

        // ---- first layer gradients from G/Gt = adjoint of z_0 ----
        {
            float *gw0 = gacc + d.woff[0];
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
                        atomicAdd(&gw0[2 * n], gx);
                        atomicAdd(&gw0[2 * n + 1], gt);
                    }
                }
            if (TAIL) {
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    const float g0 = Gt[e][0];
                    float gx = g0 * hx, gt = g0 * ht;
                    if (K >= 1) gx = fmaf(Gt[e][K >= 1 ? 1 : 0], sx, gx);
                    if (KT >= 1) gt = fmaf(Gt[e][KT >= 1 ? K + 1 : 0], stt, gt);
                    gx = row_sum16(gx);
                    gt = row_sum16(gt);
                    if (lane == 0) {
                        atomicAdd(&gw0[2 * (NM + e)], gx);
                        atomicAdd(&gw0[2 * (NM + e) + 1], gt);
                    }
                }
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
    int grid = 0;
    float *stash = nullptr;
    size_t stash_bytes = 0;
    float *slab = nullptr;
    size_t slab_bytes = 0;
};

// hidden widths with an MFMA instantiation; anything else runs on the VALU path
#ifdef PINN_DEV_KDV7_ONLY
#define PINN_MFMA_WIDTHS(X) X(50)
#else
#define PINN_MFMA_WIDTHS(X) X(50) X(64)
#endif

template <int W> size_t mfma_lds_bytes(const KDesc &kd) { return (size_t)mfma_layout<W>(kd).total * sizeof(float); }

inline bool mfma_supported(const KDesc &kd)
{
    if (kd.n_lin < 2) return false;
    const int W = kd.widths[1];
    for (int l = 1; l < kd.n_lin; ++l)
        if (kd.widths[l] != W) return false;
    size_t lds = 0;
    switch (W) {
#define X(w) case w: lds = mfma_lds_bytes<w>(kd); break;
        PINN_MFMA_WIDTHS(X)
#undef X
    default: return false;
    }
    return lds <= 160 * 1024;
}

// floats of stash one wave needs for equation jets J
template <int W> size_t mfma_stash_per_wave(const KDesc &kd, int J)
{
    typedef WidthCfg<W> C;
    const size_t ZY = (size_t)C::NT * J * 256;
    const size_t TS = (size_t)((C::RR * J * 16 + 63) / 64) * 64;
    return (size_t)(kd.n_lin - 1) * (2 * ZY + 2 * TS);
}

inline size_t mfma_stash_floats(const KDesc &kd, int J)
{
    switch (kd.widths[1]) {
#define X(w) case w: return mfma_stash_per_wave<w>(kd, J);
        PINN_MFMA_WIDTHS(X)
#undef X
    default: return 0;
    }
}

inline int mfma_plan_create(MfmaPlan &pl, const KDesc &, int n_cu)
{
    pl.grid = n_cu;
    return 0;
}

inline void mfma_plan_destroy(MfmaPlan &pl)
{
    if (pl.stash) (void)hipFree(pl.stash);
    if (pl.slab) (void)hipFree(pl.slab);
    pl.stash = pl.slab = nullptr;
}

template <int EQ, int W, int MODE>
int launch_mfma_t(MfmaPlan &pl, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    const size_t ldsb = mfma_lds_bytes<W>(kd);
    auto kern = k_mfma<EQ, W, MODE>;
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
int launch_mfma_w(MfmaPlan &pl, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    switch (kd.widths[1]) {
#define X(w) case w: return launch_mfma_t<EQ, w, MODE>(pl, kd, a, grid, st);
        PINN_MFMA_WIDTHS(X)
#undef X
    default: return -2;
    }
}

template <int MODE>
int launch_mfma_eq(MfmaPlan &pl, int eq, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    switch (eq) {
#ifndef PINN_DEV_KDV7_ONLY
    case EQ_NONE:    return launch_mfma_w<EQ_NONE, MODE>(pl, kd, a, grid, st);
    case EQ_BURGERS: return launch_mfma_w<EQ_BURGERS, MODE>(pl, kd, a, grid, st);
    case EQ_KDV5:    return launch_mfma_w<EQ_KDV5, MODE>(pl, kd, a, grid, st);
    case EQ_CH:      return launch_mfma_w<EQ_CH, MODE>(pl, kd, a, grid, st);
    case EQ_GNSS:    return launch_mfma_w<EQ_GNSS, MODE>(pl, kd, a, grid, st);
#endif
    case EQ_KDV7:    return launch_mfma_w<EQ_KDV7, MODE>(pl, kd, a, grid, st);
    default: return -1;
    }
}

}  // namespace pinn
