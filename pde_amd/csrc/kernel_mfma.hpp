// FP32-MFMA kernels: one wavefront owns a tile of 16 collocation points and
// carries the jets of a whole hidden layer in registers.
//
// What the MI355X measurements behind this design say (tools/ubench/README.md):
//  * v_mfma_f32_16x16x4_f32 executes on the same FMA lanes as the FP32 VALU: MFMA time and VALU time
//    ADD, within a wave and across the waves of a SIMD.  Nothing is gained by software-pipelining
//    MFMAs against VALU work; what counts is the instruction count and not stalling.
//  * With one wave per SIMD every latency (dependent VALU issue, LDS round trips, stash loads, MFMA
//    results) is exposed; with two waves per SIMD the other wave fills the gaps.  So: 8 waves per
//    workgroup, at most 256 registers per wave, and NO register spills.
//  * A 64-lane ds_add_f32 costs 192 cycles, serialised over the CU.  The weight-gradient tiles are
//    therefore accumulated by a plain read-add-write of a shared LDS image under a per-layer lock;
//    LDS float atomics are only used with <= 4 active lanes (bias / tail entries).
//
// Layout (v_mfma_f32_16x16x4_f32, MI355X guide section 3): for a 16x16 tile the accumulator of lane
// l holds rows 4*(l>>4)+r (r=0..3), column l&15.  NEURONS sit on rows, POINTS on columns:
//     lane (p = l&15, q = l>>4), register r  <->  neuron 16*T + 4*q + r, point p
// for neuron tile T.  An activation set is  A[T][k][r]  (k = jet component): NT*J*4 registers.
//
//  * forward  Z^T[o][p] = sum_i W[o][i] Y^T[i][p]: A operand = W fragment from LDS, B operand = the
//    previous layer's registers used directly (k-step (Ti, r) covers input neurons 16Ti + 4q + r).
//    The GEMM runs one jet component at a time and OVERWRITES that component in place (a linear layer
//    never mixes components), so a layer needs NT*J*4 + NT*4 registers, not twice NT*J*4;
//  * adjoint  Yb^T[i][p] = sum_o W[o][i] Zb^T[o][p]: same, with W^T fragments;
//  * weight gradient  Wb[o][i] = sum_{p,k} Zb_k[o][p] Y_k[i][p]: the reduction runs over points, which
//    sit on lanes, so both operands are needed transposed.  Zb goes through a padded LDS staging
//    image; Y comes back from the stash, which stores y as [neuron][point] so that the transposed
//    fragment is one b128 load per lane (MFMA k-index kk of step s  <->  point 4*kk + s).
//
// Widths that are 16*NT + R with 1 <= R <= 4 (the reference's width 50 = 48 + 2) do not pad up to the
// next tile: the R "tail" neurons are carried per lane on the VALU (replicated over the four lane
// quads).
//
// Stash (global, private per wave, re-used every tile): per tanh layer  z main | y main | z tail | y tail.
// The reverse sweep re-reads z (neuron by neuron, prefetched) and recomputes y and w = 1 - y^2 from it;
// the stashed y only feeds the weight-gradient GEMM.  z of the first layer is recomputed (affine in
// x,t); the last tanh layer keeps z in registers and is never stashed.
//
// tanh jets / residuals: jet_math.hpp.  Reference semantics as kernel_valu.hpp.
#pragma once
#include <type_traits>
#include "jet_math.hpp"
#include "pinn_common.hpp"

namespace pinn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#ifndef PINN_MFMA_BLOCK
#define PINN_MFMA_BLOCK 512
#endif
constexpr int MFMA_MAX_WAVES = 8;
constexpr int MFMA_LDS_LIMIT = 160 * 1024;
constexpr int STAGE_ROW = 20;                  // floats per point row of the staging image (16 + pad: no bank conflicts)
constexpr int STAGE_TILE = 16 * STAGE_ROW;

template <int W> struct WidthCfg {
    static constexpr int REM = W % 16;
    // tail neurons on the VALU pay from three MFMA tiles on (width 50 = 48 + 2: padding to 64 costs the 7kdv net +33 %
    // and spills); width 20 as 16 + 4 was 25-40 % SLOWER than padded to two tiles (r03: 10 000 points 133 -> 80 us,
    // 1M points 3.51 -> 2.64 ms): below 48 the tail rows / columns / replicated tanh outweigh the padded MFMAs
#ifndef PINN_TAIL_MIN_W
#define PINN_TAIL_MIN_W 32
#endif
    static constexpr bool TAIL = (REM != 0 && REM <= 4 && W > PINN_TAIL_MIN_W);
    static constexpr int NT = TAIL ? W / 16 : (W + 15) / 16;  // MFMA neuron tiles
    static constexpr int R = TAIL ? REM : 0;                   // tail neurons on the VALU
    static constexpr int RR = R > 0 ? R : 1;                   // array extent
    static constexpr int NM = 16 * NT;                         // neurons covered by tiles
    // Workgroup shape.  Up to width 64 the weights, the weight-gradient images and the staging area of
    // eight waves fit in LDS together: two waves per SIMD, <= 256 registers each.  Wider layers
    // (GNSS, width 100 = 96 + 4) fill LDS with the weights alone: four waves (one per SIMD, 512
    // registers: the 36 weight-gradient accumulator tiles of a layer live in AGPRs, nothing needs
    // parking in LDS), Zb staged STG tiles at a time, and the weight-gradient tiles go to the
    // workgroup's global slab with hardware float atomics (L2: 123 KB per workgroup) instead of LDS
    // images; the small entries (biases, tails) stay in the compact LDS block.
    //   Wide layers whose jet set is small (NT*J*4 <= 128 registers: GNSS, Burgers, plain net_u at
    // width 100) still run EIGHT waves of <= 256 registers: the weight-gradient GEMM then works on one
    // row of accumulator tiles at a time (STG = 1, flushed by atomics row by row) and one staging tile
    // and the parked tail rows per wave are all the per-wave LDS there is room for.
    static constexpr bool WIDE = W > 64;
    static constexpr bool wide8(int J) { return WIDE && NT * J * 4 <= 128; }
    static constexpr int waves(int J) { return (WIDE && !wide8(J)) ? 4 : PINN_MFMA_BLOCK / 64; }
    static constexpr int block(int J) { return 64 * waves(J); }
    static constexpr bool WGLOBAL = WIDE;
    static constexpr bool PARK = !WIDE;                        // park registers in LDS around the neuron loops
    // Zb tiles staged per pass of the weight-gradient GEMM
    static constexpr int stg(int J) { return wide8(J) ? 1 : (WIDE ? (NT + 2) / 3 : NT); }
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
    // gradient accumulators.  Everything except the NM x NM main block of the hidden->hidden weight
    // gradients is small and is accumulated with few-lane LDS atomics in the compact block `gs`:
    //   [W0: 2W][b0: W] | per hidden layer l: [b_l: W][tail rows: R x W][tail cols: NM x R] | [Wout][bout] | [loss: 2]
    int gs, gs_hid, gs_hid_stride, gs_out, gs_loss, gs_size;
    int wimg;         // main blocks as MFMA accumulator images [l-1][To][Ti][lane][r] (tile = 256 floats)
    int locks;        // one int per hidden->hidden layer
    int stage;        // per wave NT*STAGE_TILE
    int xt;           // per wave [2][64]: the scaled coordinates of the tile's points, parked between the sweeps
    int ztl;          // per wave [R][J][16]: tail rows of Zb, [component][point], for the weight-gradient phase
    int atl;          // per wave [R][J][16]: the tail registers, parked while the tiled neurons are processed
    int total;
    bool wglobal;     // main blocks of the weight gradients go to the workgroup's global slab by float atomics (no `wimg`)
};

template <int W> __host__ __device__ inline MfmaLayout mfma_layout(const KDesc &d, int J)
{
    typedef WidthCfg<W> C;
    MfmaLayout L;
    int off = 0;
    L.hh = off; off += (d.n_lin - 2) * C::HH;
    L.io = off; off += C::IO;
    L.zero_end = off;
    L.gs = off;
    L.gs_hid = 3 * W;
    L.gs_hid_stride = W + C::R * W + C::NM * C::R;
    L.gs_out = L.gs_hid + (d.n_lin - 2) * L.gs_hid_stride;
    L.gs_loss = L.gs_out + d.nout * W + d.nout;
    L.gs_size = ((L.gs_loss + MAXF + 3) / 4) * 4;
    off += L.gs_size;
    // The accumulator images are the one big LDS consumer that can move: wide layers never have them, and a
    // deep net whose weights + images + staging would not fit (64 x 5 / 64 x 6, 7kdv/bc-7kdv3soliton.py:366)
    // accumulates in its global slab like the wide shapes do, instead of falling back to the VALU kernels.
    const int images = (d.n_lin - 2) * C::NT * C::NT * 256;
    const int rest = ((d.n_lin - 2 + 3) / 4) * 4 + 4 + C::waves(J) * C::stg(J) * STAGE_TILE + (C::PARK ? C::waves(J) * 128 : 0)
                     + C::waves(J) * C::R * J * 16 * (C::PARK ? 2 : 1);
    L.wglobal = C::WGLOBAL || (off + images + rest) * 4 > MFMA_LDS_LIMIT;
    L.wimg = off; off += L.wglobal ? 0 : images;
    L.locks = off; off += ((d.n_lin - 2 + 3) / 4) * 4 + 4;
    L.stage = off; off += C::waves(J) * C::stg(J) * STAGE_TILE;
    L.xt = off; off += C::PARK ? C::waves(J) * 128 : 0;
    L.ztl = off; off += C::waves(J) * C::R * J * 16;
    L.atl = off; off += C::PARK ? C::waves(J) * C::R * J * 16 : 0;
    L.total = off;
    return L;
}

#ifndef PINN_ZPF
#define PINN_ZPF 2           // reverse sweep: z of a neuron is requested this many neurons ahead
#endif
#define PINN_ADD_S(addr, v) atomicAdd(addr, v)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f64_16x16x4_f64: A/B operands as the f32 form (one value per lane), but the result's rows sit
// differently: lane (p, q) register r holds row q + 4r (f32 form: 4q + r), column p.
__device__ __forceinline__ f64x4 mfma4d(double a, double b, f64x4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }

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
__device__ __forceinline__ double quad_rows_sum(double v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Stash accesses go through a buffer descriptor: one per-lane VGPR offset plus a SCALAR offset per
// access, so no per-access 64-bit address is ever materialised (or hoisted and spilled).
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
// Boundary between two jet components of an in-place GEMM: nothing is scheduled across it and LDS
// contents are re-read after it, so the weight fragments are NOT kept in registers over the whole
// component loop (36+ registers that the in-place scheme exists to save); LDS reads are cheap.
#define PINN_JB() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// Pin a value: it is computed before this point and not re-derived later.  Without it LLVM sinks the
// unrolled jet arithmetic of a neuron past the exec-masked atomics into the block of its first use
// (a whole phase later) and spills the inputs instead.
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(f32x4 &v) { asm volatile("" : "+v"(v)); }

// Workgroup-level spin lock in LDS (lane 0 spins, the wave stays converged).  All eight waves of the
// workgroup are resident and make progress independently, so the holder always gets to release.
__device__ __forceinline__ void lds_lock(int *lk, int lane)
{
    for (;;) {
        int old = 1;
        if (lane == 0) old = atomicCAS(lk, 0, 1);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == 0) break;
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_unlock(int *lk, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(lk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int EQ, int W, int MODE>
__global__ __launch_bounds__(WidthCfg<W>::block(EqTraits<EQ>::K + 1 + EqTraits<EQ>::KT), 1) void k_mfma(KDesc d, CallArgs a, float *stash)
{
    typedef WidthCfg<W> C;
    constexpr int NT = C::NT, R = C::R, RR = C::RR, NM = C::NM, S = C::S;
    constexpr bool TAIL = C::TAIL, WGLOBAL = C::WGLOBAL;
    constexpr int K = EqTraits<EQ>::K, KT = EqTraits<EQ>::KT, J = K + 1 + KT;
    constexpr int MFMA_BLOCK = C::block(J), MFMA_WAVES = C::waves(J), STG = C::stg(J);
    constexpr bool GROUPED = C::wide8(J);          // weight-gradient GEMM one row of accumulator tiles at a time
    // "precise top" (jet_math.hpp, EqTraits::PRECISE): float64 accumulation in the last hidden Linear layer (f64 MFMA),
    // the output layer and the residual polynomial; UT = the arithmetic type of the output sums
    constexpr bool PRECISE = EqTraits<EQ>::PRECISE && !C::WIDE;
    constexpr int NO = EqTraits<EQ>::NOUT;         // most outputs this equation's nets have: rows >= NO of u / D / ub are constant 0
    typedef std::conditional_t<PRECISE, double, float> UT;
#ifndef PINN_ZT_LDS_ALL
#define PINN_ZT_LDS_ALL 1
#endif
    // forward tail pre-activations wait in the (idle) staging area instead of in registers
    constexpr bool ZT_LDS = GROUPED || (PINN_ZT_LDS_ALL && TAIL && R * J * 16 <= STG * STAGE_TILE);
    // the first layer's pre-activation is affine in (x, t): nothing above x-order 1 / t-tangent order 0
    constexpr int ZN0 = K >= 1 ? 1 : 0, ZTN0 = KT >= 1 ? 1 : 0;
    constexpr int TILE = 256;                      // floats of one 16x16 stash image
    constexpr int ZY = NT * J * TILE;              // main z (or y) images of one hidden layer
    constexpr int TS = ((RR * J * 16 + 63) / 64) * 64;  // tail z (or y) of one hidden layer: [e][k][p]
    constexpr int LAYER_STASH = 2 * ZY + 2 * TS;   // z main | y main | z tail | y tail
    extern __shared__ __attribute__((aligned(16))) float lds[];

    constexpr int ZTL_J = EqTraits<EQ>::K + 1 + EqTraits<EQ>::KT;
    const MfmaLayout L = mfma_layout<W>(d, ZTL_J);
    const int n_lin = d.n_lin;
    const int NH = n_lin - 1;                      // tanh layers
    const int nout = d.nout;
    const int nf = EqTraits<EQ>::NF ? EqTraits<EQ>::NF : (nout < MAXOUT ? nout : MAXOUT);
    const int P = d.P;
    const int tid = threadIdx.x;

    // ---- stage the weights in LDS (zero padded) ----
    for (int i = tid; i < L.zero_end; i += MFMA_BLOCK) lds[i] = 0.0f;
    if (MODE != MODE_FWD)
        for (int i = L.gs + tid; i < L.stage; i += MFMA_BLOCK) lds[i] = 0.0f;   // gs, images, locks
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

    // wave id through readfirstlane: makes every per-wave base provably wave-uniform
    const int lane0 = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long wave_g = (long long)blockIdx.x * MFMA_WAVES + wv;
    const long long nwaves = (long long)gridDim.x * MFMA_WAVES;
    const long long ntiles = (a.n + 15) / 16;
    // per wave: the layers' images, then 128 floats where the eight-wave wide shape parks the tile's coordinates
    const int XT_OFF = NH * LAYER_STASH;
    float *const st = stash + wave_g * (long long)(NH * LAYER_STASH + 128);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(st, 0, (NH * LAYER_STASH + 128) * 4, 0x00020000);
    float *const stage = lds + L.stage + wv * STG * STAGE_TILE;
    float *const gs = lds + L.gs;
    const float *const io = lds + L.io;
    // compact gradient block of Linear layer l < n_lin-1: bias first, then (hidden layers) tail rows / cols
    auto gblock = [&](int l) { return gs + (l == 0 ? 2 * W : L.gs_hid + (l - 1) * L.gs_hid_stride); };
    // the small gradient entries (accumulated with few-lane LDS atomics)
    float *const slabp = (MODE != MODE_FWD) ? a.slab + (long long)blockIdx.x * (d.P + MAXF) : nullptr;
    auto p_b = [&](int l) { return gblock(l); };                      // bias of Linear layer l < n_lin-1: [n]
    auto p_tr = [&](int l) { return gblock(l) + W; };                 // tail rows of Wb_l: [e*W + i]
    auto p_tc = [&](int l) { return gblock(l) + W + R * W; };         // tail columns of Wb_l: [o*TCS + e]
    constexpr int TCS = R > 0 ? R : 1;
    float *const p_wout = gs + L.gs_out;                              // [o*W + n]
    float *const p_bout = gs + L.gs_out + d.nout * W;                 // [o]
    float *const p_w0 = gs;                                           // [2n + c]
    // Lane-derived indices and per-lane byte offsets into the stash.  They are re-derived from an opaque
    // copy of the lane id at the start of every phase (PINN_REFRESH): left alone, the compiler hoists a
    // few dozen loop-invariant per-lane addresses out of the tile loop, spills them, and every reload
    // then waits (vmcnt) behind the stash stores in flight.
    int lane, p, q;
    int vnat;   // b128 slot of a [point][neuron] image (z) = transposed b128 slot of a [neuron][point] image (y)
    int vyn;    // dword slot of neuron 4q (+r: add 16 floats) in a [neuron][point] image
    int vp;
#define PINN_REFRESH() do { lane = lane0; asm volatile("" : "+v"(lane)); p = lane & 15; q = lane >> 4; \
        vnat = (p * 16 + 4 * q) * 4; vyn = (q * 64 + p) * 4; vp = p * 4; } while (0)
    PINN_REFRESH();
    float lsum[MAXF] = {0.0f, 0.0f, 0.0f};

    // wave-uniform: keep them in SGPRs
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float dxr = uni(d.ubx - d.lbx), dtr = uni(d.ubt - d.lbt);
    // the chain-rule factors come in as kernel arguments (SGPRs): computed here, the compiler proves them
    // uniform, keeps the quotients in VGPRs, spills them and reloads them at every use
    const float sx = d.sx, stt = d.st;
    const float two_scale = uni(2.0f * a.scale);
    float *const xtp = lds + L.xt + wv * 128;
    float *const ztl = lds + L.ztl + wv * (R * ZTL_J * 16);
    // (eight-wave wide shape: no LDS left for a parking area of its own, but the staging tile is idle
    // whenever the tail registers are parked and has exactly their size)
    static_assert(!GROUPED || R * ZTL_J * 16 <= STAGE_TILE, "tail set does not fit the staging tile");
    float *const atl = GROUPED ? stage : lds + L.atl + wv * (R * ZTL_J * 16);

    // Tiles are dealt to workgroups first and to the waves of a workgroup second (tile = block + grid * wave):
    // a launch with fewer tiles than wave slots (the scripts' N_f = 10 000 .. 16 000) then spreads over all CUs
    // and over the four SIMDs of each CU before any SIMD has to interleave two tiles.  That is the FIRST tile of
    // every wave; a launch with more tiles than waves hands the rest out first come, first served from a counter
    // in the launch's workspace: dealt statically (tile += nwaves) the CUs drift apart by several tiles over the
    // 30 rounds of a 1M-point launch and the launch lasts as long as the slowest one (7kdv: 6.27 -> 5.85 ms).
    // The counters start at zero and the last wave to finish puts them back (no clearing launch, replays from a
    // HIP graph); launches that share a workspace must not overlap, which they must not anyway (stash, slab).
    const bool dyn = a.tile_ctr != nullptr && ntiles > nwaves;
    long long tile = blockIdx.x + (long long)gridDim.x * wv;
    for (bool first = true;; first = false) {
        if (!first) {
            if (dyn) {
                int t = 0;
                if (lane0 == 0) t = atomicAdd(&a.tile_ctr[0], 1);
                tile = nwaves + __builtin_amdgcn_readfirstlane(t);
            } else {
                tile += nwaves;
            }
        }
        if (tile >= ntiles) break;
        PINN_REFRESH();
        // (point index and validity are re-derived where they are needed again, not kept live)
        const long long pc = (tile * 16 + p < a.n) ? tile * 16 + p : a.n - 1;
        const float xv = a.x[pc * a.stride], tv = a.t[pc * a.stride];
        // scaled coordinates: needed by the first layer now and again at the very end of the reverse
        // sweep; in between they are parked in LDS (PINN_XT_PARK / PINN_XT_FETCH) instead of being
        // spilled to scratch by the register allocator
        float hx = 2.0f * (xv - d.lbx) / dxr - 1.0f;
        float ht = 2.0f * (tv - d.lbt) / dtr - 1.0f;
#define PINN_XT_PARK() do { if (C::PARK) { xtp[lane] = hx; xtp[64 + lane] = ht; } \
        else if (GROUPED) { sb_store1(srs, lane * 4, XT_OFF, hx); sb_store1(srs, lane * 4, XT_OFF + 64, ht); } } while (0)
#define PINN_XT_FETCH() do { if (C::PARK) { asm volatile("" ::: "memory"); hx = xtp[lane]; ht = xtp[64 + lane]; } \
        else if (GROUPED) { asm volatile("" ::: "memory"); hx = sb_load1(srs, lane * 4, XT_OFF); ht = sb_load1(srs, lane * 4, XT_OFF + 64); } } while (0)
        if (MODE != MODE_FWD) PINN_XT_PARK();

        // The one register-resident set: activations on the way up, adjoints on the way down.
        f32x4 A[NT][J];
        float At[RR][J];       // tail neurons (same value in all four lane quads)
        // The reverse sweep's per-neuron loops need every register they can get: the tail set is parked
        // in LDS around them.
#define PINN_AT_PARK() do { if (TAIL && (C::PARK || GROUPED)) { if (q == 0) { _Pragma("unroll") for (int e = 0; e < R; ++e) _Pragma("unroll") for (int j = 0; j < J; ++j) \
            atl[(e * ZTL_J + j) * 16 + p] = At[e][j]; } } } while (0)
#define PINN_AT_FETCH() do { if (TAIL && (C::PARK || GROUPED)) { asm volatile("" ::: "memory"); _Pragma("unroll") for (int e = 0; e < R; ++e) _Pragma("unroll") for (int j = 0; j < J; ++j) \
            At[e][j] = atl[(e * ZTL_J + j) * 16 + p]; } } while (0)

        // first-layer pre-activation jets (also recomputed in the reverse sweep: they are affine in (x,t),
        // so they are never stashed).  Weights of tile T as three b128 LDS reads, then neuron r of them.
        struct W0 { f32x4 wx, wt, b0; };
        auto w0_tile = [&](int T) __attribute__((always_inline)) {
            W0 w;
            w.wx = *(const f32x4 *)&io[C::O_W0X + 16 * T + 4 * q];
            w.wt = *(const f32x4 *)&io[C::O_W0T + 16 * T + 4 * q];
            w.b0 = *(const f32x4 *)&io[C::O_B0 + 16 * T + 4 * q];
            return w;
        };
        auto z0_main = [&](const W0 &w, int r, float (&z)[J]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < J; ++j) z[j] = 0.0f;
            z[0] = fmaf(w.wx[r], hx, fmaf(w.wt[r], ht, w.b0[r]));
            if (K >= 1) z[K >= 1 ? 1 : 0] = w.wx[r] * sx;
            if (KT >= 1) z[KT >= 1 ? K + 1 : 0] = w.wt[r] * stt;
        };
        auto z0_tail = [&](int e, float (&z)[J]) __attribute__((always_inline)) {
            const float wx = io[C::O_W0TAIL + e], wt = io[C::O_W0TAIL + 4 + e], b0 = io[C::O_W0TAIL + 8 + e];
#pragma unroll
            for (int j = 0; j < J; ++j) z[j] = 0.0f;
            z[0] = fmaf(wx, hx, fmaf(wt, ht, b0));
            if (K >= 1) z[K >= 1 ? 1 : 0] = wx * sx;
            if (KT >= 1) z[KT >= 1 ? K + 1 : 0] = wt * stt;
        };

        // Output-layer partial sums.  The LAST tanh layer keeps its pre-activations z in A/At (they go
        // straight into the reverse sweep, which recomputes y from z everywhere); its outputs y only
        // exist neuron by neuron, feeding these sums.
        UT u[MAXOUT][J];
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o)
#pragma unroll
            for (int j = 0; j < J; ++j) u[o][j] = (UT)0;
        // y of tiled neuron n (lane-local partial sums; reduced over the lane quads at the end)
        auto out_acc_main = [&](int n, const float (&y)[J]) __attribute__((always_inline)) {
#pragma unroll
            for (int o = 0; o < NO; ++o)
                if (o < nout) {
                    const UT wo = (UT)io[C::O_WOUT + o * NM + n];
#pragma unroll
                    for (int j = 0; j < J; ++j) u[o][j] = fma_t(wo, (UT)y[j], u[o][j]);
                }
        };
        // y of tail neuron e: the value is replicated over the four quads, so only quad 0 adds it
        auto out_acc_tail = [&](int e, const float (&y)[J]) __attribute__((always_inline)) {
#pragma unroll
            for (int o = 0; o < NO; ++o)
                if (o < nout) {
                    const UT wo = (UT)(q == 0 ? io[C::O_WOUTT + o * 4 + e] : 0.0f);
#pragma unroll
                    for (int j = 0; j < J; ++j) u[o][j] = fma_t(wo, (UT)y[j], u[o][j]);
                }
        };

        // ---- first layer (2 -> W), VALU ----
        {
            const bool topl = NH == 1;
            const bool keep_y = (MODE != MODE_FWD) && !topl;
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                const W0 w0 = w0_tile(T);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z[J], y[J], w[K + 1];
                    z0_main(w0, r, z);
                    tanh_jet_fwd<K, KT, ZN0, ZTN0>(z, y, w);
#pragma unroll
                    for (int j = 0; j < J; ++j) pin(y[j]);
                    // (a select, not two branches writing A: merging the branches made the compiler copy the whole
                    // tile's registers back and forth once per neuron)
                    if (topl) out_acc_main(16 * T + 4 * q + r, y);
#pragma unroll
                    for (int j = 0; j < J; ++j) A[T][j][r] = topl ? z[j] : y[j];
                    PINN_SB();
                }
                if (keep_y) {
#pragma unroll
                    for (int j = 0; j < J; ++j) sb_store4(srs, vnat, ZY + (T * J + j) * TILE, A[T][j]);
                }
            }
#pragma unroll
            for (int e = 0; e < RR; ++e)
#pragma unroll
                for (int j = 0; j < J; ++j) At[e][j] = 0.0f;
            if (TAIL) {
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    float z[J], y[J], w[K + 1];
                    z0_tail(e, z);
                    tanh_jet_fwd<K, KT, ZN0, ZTN0>(z, y, w);
                    if (topl) {
                        out_acc_tail(e, y);
#pragma unroll
                        for (int j = 0; j < J; ++j) At[e][j] = z[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < J; ++j) At[e][j] = y[j];
                        if (keep_y && q == 0) {
#pragma unroll
                            for (int j = 0; j < J; ++j) sb_store1(srs, vp, 2 * ZY + TS + (e * J + j) * 16, y[j]);
                        }
                    }
                }
            }
        }


        // ---- hidden -> hidden layers on the matrix cores, one jet component at a time, in place ----
        for (int l = 1; l <= n_lin - 2; ++l) {
            PINN_REFRESH();
            const float *img = lds + L.hh + (l - 1) * C::HH;
            const int zs = l * LAYER_STASH;   // stash offset (floats) of this layer
            const bool topl = l == NH - 1;
            const bool keep_y = (MODE != MODE_FWD) && !topl;
            float zt[RR][J];
#pragma unroll
            for (int e = 0; e < RR; ++e)
#pragma unroll
                for (int j = 0; j < J; ++j) zt[e][j] = 0.0f;
            // one jet component of the layer in float32: A[:, j] <- W A[:, j] (+ bias for j = 0)
            auto gemm_f32 = [&](int j) __attribute__((always_inline)) {
                f32x4 acc[NT];
#pragma unroll
                for (int To = 0; To < NT; ++To) {
                    acc[To] = (j == 0) ? *(const f32x4 *)&img[C::O_BIAS + 16 * To + 4 * q] : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                    if (TAIL) {
                        // tail inputs into the tiled outputs: part of the accumulator's initial value
#pragma unroll
                        for (int e = 0; e < R; ++e) {
                            const f32x4 wc = *(const f32x4 *)&img[C::O_WTC + e * NM + 16 * To + 4 * q];
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[To][r] = fmaf(wc[r], At[e][j], acc[To][r]);
                        }
                    }
                    // A fragments: lane (m = p, q) holds W[16To + m][16Ti + 4q + r], r = 0..3; the NT reads
                    // of this output tile are issued together and waited for once
                    f32x4 wf[NT];
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) wf[Ti] = *(const f32x4 *)&img[(16 * To + p) * S + 16 * Ti + 4 * q];
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) pin(wf[Ti]);
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * Ti + r < W) acc[To] = mfma4(wf[Ti][r], A[Ti][j][r], acc[To]);
                    PINN_SB();
                }
                PINN_SB();
                if (TAIL) {
                    // tail outputs: every lane sums over its 4*NT tiled inputs, then across the lane quads
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        float s = 0.0f;
#pragma unroll
                        for (int Ti = 0; Ti < NT; ++Ti) {
                            const f32x4 wr = *(const f32x4 *)&img[C::O_WTR + e * NM + 16 * Ti + 4 * q];
#pragma unroll
                            for (int r = 0; r < 4; ++r) s = fmaf(wr[r], A[Ti][j][r], s);
                        }
                        s = quad_rows_sum(s);
                        if (j == 0) s += img[C::O_BT + e];
#pragma unroll
                        for (int e2 = 0; e2 < R; ++e2) s = fmaf(img[C::O_WTT + e * 4 + e2], At[e2][j], s);
                        // (eight-wave wide shape: the tail pre-activations wait in the idle staging tile, not
                        // in registers, until the tiled neurons are done)
                        if (ZT_LDS) { if (q == 0) stage[(e * J + j) * 16 + p] = s; }
                        else zt[e][j] = s;
                        PINN_SB();
                    }
                }
#pragma unroll
                for (int To = 0; To < NT; ++To) A[To][j] = acc[To];
            };
            // "precise top": the same component accumulated in float64 on v_mfma_f64_16x16x4_f64 (half the rate of the
            // f32 form).  Lane (p, q) register r of that instruction's result is row q + 4r, so the A operand's rows
            // are dealt as 4 (p & 3) + (p >> 2): the result then holds neuron 16 To + 4q + r exactly where the f32
            // form puts it.  The activations are converted once per component, the weights once per use.
            auto gemm_f64 = [&](int j) __attribute__((always_inline)) {
                const int prow = 4 * (p & 3) + (p >> 2);
                double yd[NT][4];
#pragma unroll
                for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) yd[Ti][r] = (double)A[Ti][j][r];
                f64x4 accd[NT];
#pragma unroll
                for (int To = 0; To < NT; ++To) {
                    const f32x4 b4 = (j == 0) ? *(const f32x4 *)&img[C::O_BIAS + 16 * To + 4 * q] : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) accd[To][r] = (double)b4[r];
                    if (TAIL) {
#pragma unroll
                        for (int e = 0; e < R; ++e) {
                            const f32x4 wc = *(const f32x4 *)&img[C::O_WTC + e * NM + 16 * To + 4 * q];
                            const double ae = (double)At[e][j];
#pragma unroll
                            for (int r = 0; r < 4; ++r) accd[To][r] = fma((double)wc[r], ae, accd[To][r]);
                        }
                    }
                    f32x4 wf[NT];
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) wf[Ti] = *(const f32x4 *)&img[(16 * To + prow) * S + 16 * Ti + 4 * q];
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) pin(wf[Ti]);
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * Ti + r < W) accd[To] = mfma4d((double)wf[Ti][r], yd[Ti][r], accd[To]);
                    PINN_SB();
                }
                PINN_SB();
                if (TAIL) {
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        double sd = 0.0;
#pragma unroll
                        for (int Ti = 0; Ti < NT; ++Ti) {
                            const f32x4 wr = *(const f32x4 *)&img[C::O_WTR + e * NM + 16 * Ti + 4 * q];
#pragma unroll
                            for (int r = 0; r < 4; ++r) sd = fma((double)wr[r], yd[Ti][r], sd);
                        }
                        sd = quad_rows_sum(sd);
                        if (j == 0) sd += (double)img[C::O_BT + e];
#pragma unroll
                        for (int e2 = 0; e2 < R; ++e2) sd = fma((double)img[C::O_WTT + e * 4 + e2], (double)At[e2][j], sd);
                        const float s = (float)sd;
                        if (ZT_LDS) { if (q == 0) stage[(e * J + j) * 16 + p] = s; }
                        else zt[e][j] = s;
                        PINN_SB();
                    }
                }
#pragma unroll
                for (int To = 0; To < NT; ++To)
#pragma unroll
                    for (int r = 0; r < 4; ++r) A[To][j][r] = (float)accd[To][r];
            };
#pragma unroll
            for (int j = 0; j < J; ++j) {
                // Which components of the last hidden layer need float64: the value, the first x-derivative and the
                // first t-derivative of the pre-activation.  The tanh recurrences raise THEIR rounding errors to the
                // high orders (y_3 ~ z_1^3 ...), where the residual's large terms cancel; the high-order components'
                // own dot-product errors enter linearly and do not matter (tools/ch_noise_lab.py: {0, 1, K+1} in
                // float64 gives 1.09x the reference's noise, all seven 1.11x, {3, 6} alone 2.6x, none 4.1x).
                const bool pj = PRECISE && (j == 0 || j == 1 || j == K + 1);
                if (pj && topl) gemm_f64(j);
                else gemm_f32(j);
                PINN_JB();
            }
            // A = z of this layer.  Last tanh layer: z stays, y feeds the output sums.  Otherwise: stash z
            // and y, tanh in place, neuron by neuron.
            PINN_REFRESH();
#pragma unroll
            for (int To = 0; To < NT; ++To)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z[J], y[J], w[K + 1];
#pragma unroll
                    for (int j = 0; j < J; ++j) z[j] = A[To][j][r];
                    if (keep_y) {
#pragma unroll
                        for (int j = 0; j < J; ++j) sb_store1(srs, vyn, zs + (To * J + j) * TILE + r * 16, z[j]);
                    }
                    tanh_jet_fwd<K, KT>(z, y, w);
#pragma unroll
                    for (int j = 0; j < J; ++j) pin(y[j]);
                    if (topl) out_acc_main(16 * To + 4 * q + r, y);
#pragma unroll
                    for (int j = 0; j < J; ++j) A[To][j][r] = topl ? z[j] : y[j];
                    PINN_SB();
                    if (r == 3 && keep_y) {
#pragma unroll
                        for (int j = 0; j < J; ++j) sb_store4(srs, vnat, zs + ZY + (To * J + j) * TILE, A[To][j]);
                    }
                }
            if (TAIL) {
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    float y[J], w[K + 1];
                    if (ZT_LDS) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int j = 0; j < J; ++j) zt[e][j] = stage[(e * J + j) * 16 + p];
                    }
                    tanh_jet_fwd<K, KT>(zt[e], y, w);
                    if (topl) {
                        out_acc_tail(e, y);
#pragma unroll
                        for (int j = 0; j < J; ++j) At[e][j] = zt[e][j];
                    } else {
#pragma unroll
                        for (int j = 0; j < J; ++j) At[e][j] = y[j];
                        if (keep_y && q == 0) {
#pragma unroll
                            for (int j = 0; j < J; ++j) {
                                sb_store1(srs, vp, zs + 2 * ZY + (e * J + j) * 16, zt[e][j]);
                                sb_store1(srs, vp, zs + 2 * ZY + TS + (e * J + j) * 16, y[j]);
                            }
                        }
                    }
                }
            }
        }

        // ---- output layer (W -> nout), VALU + cross-quad reduction ----
        PINN_REFRESH();
        float D[MAXOUT][J], Tt[MAXOUT][J];
        float f[MAXF] = {0.0f, 0.0f, 0.0f};
        {
            // true derivatives and the residual polynomial in the arithmetic type of the output sums
            UT Du[MAXOUT][J], Tu[MAXOUT][J], fu[MAXF] = {(UT)0, (UT)0, (UT)0};
#pragma unroll
            for (int o = 0; o < MAXOUT; ++o) {
#pragma unroll
                for (int j = 0; j < J; ++j) { Du[o][j] = (UT)0; Tu[o][j] = (UT)0; D[o][j] = 0.0f; Tt[o][j] = 0.0f; }
                if (o < NO) {
                    if (o < nout) {
#pragma unroll
                        for (int j = 0; j < J; ++j) u[o][j] = quad_rows_sum(u[o][j]);
                        u[o][0] += (UT)io[C::O_BOUT + o];
                    }
#pragma unroll
                    for (int k = 0; k <= K; ++k) { Du[o][k] = (UT)factorial(k) * u[o][k]; D[o][k] = (float)Du[o][k]; }
#pragma unroll
                    for (int k = 0; k < KT; ++k) { Tu[o][k] = (UT)factorial(k) * u[o][K + 1 + k]; Tt[o][k] = (float)Tu[o][k]; }
                }
            }
            Residual<EQ>::eval(Du, Tu, d.coef, nout, fu);
#pragma unroll
            for (int c = 0; c < MAXF; ++c) f[c] = (float)fu[c];
        }
        const long long pt = tile * 16 + p;
        const bool valid = pt < a.n;

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
        float fb[MAXF] = {0.0f, 0.0f, 0.0f};
        for (int c = 0; c < nf; ++c) {
            if (MODE == MODE_LOSS) {
                fb[c] = valid ? two_scale * f[c] : 0.0f;
                lsum[c] += (valid && q == 0) ? f[c] * f[c] : 0.0f;
            } else {
                fb[c] = valid ? a.gbar[pt * nf + c] : 0.0f;
            }
        }
        float ub[MAXOUT][J];
        {
            float Db[MAXOUT][J], Tb[MAXOUT][J];
#pragma unroll
            for (int o = 0; o < MAXOUT; ++o)
#pragma unroll
                for (int j = 0; j < J; ++j) { Db[o][j] = 0.0f; Tb[o][j] = 0.0f; }
            Residual<EQ>::adj(D, Tt, d.coef, nout, fb, Db, Tb);
#pragma unroll
            for (int o = 0; o < MAXOUT; ++o) {
#pragma unroll
                for (int j = 0; j < J; ++j) ub[o][j] = 0.0f;
#pragma unroll
                for (int k = 0; k <= K; ++k) ub[o][k] = factorial(k) * Db[o][k];
#pragma unroll
                for (int k = 0; k < KT; ++k) ub[o][K + 1 + k] = factorial(k) * Tb[o][k];
            }
        }


        // ---- output layer: bias gradient ----
        PINN_REFRESH();
#pragma unroll
        for (int o = 0; o < NO; ++o)
            if (o < nout) {
                const float bs = row_sum16(ub[o][0]);
                if (lane == 0) PINN_ADD_S(&p_bout[o], bs);
            }

        // ---- top tanh layer (NH-1): A/At hold its pre-activations z; y is recomputed, its adjoint is the
        //      outer product Wout^T ub, and the output layer's weight gradient is formed on the way.
        //      A/At: z -> Zb. ----
        {
            const int hl = NH - 1;
            float *gbias = p_b(hl);
            PINN_AT_PARK();
            PINN_SB();
#pragma unroll
            for (int T = 0; T < NT; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z[J], y[J], w[K + 1], yb[J], zb[J];
                    const int n = 16 * T + 4 * q + r;
#pragma unroll
                    for (int j = 0; j < J; ++j) { z[j] = A[T][j][r]; yb[j] = 0.0f; }
                    tanh_jet_fwd<K, KT>(z, y, w);
#pragma unroll
                    for (int o = 0; o < NO; ++o)
                        if (o < nout) {
                            const float wo = io[C::O_WOUT + o * NM + n];
                            float v = 0.0f;
#pragma unroll
                            for (int j = 0; j < J; ++j) {
                                v = fmaf(ub[o][j], y[j], v);
                                yb[j] = fmaf(wo, ub[o][j], yb[j]);
                            }
                            v = row_sum16(v);
                            if (p == 0 && n < W) PINN_ADD_S(&p_wout[o * W + n], v);
                        }
                    tanh_jet_bwd<K, KT>(z, y, w, yb, zb);
#pragma unroll
                    for (int j = 0; j < J; ++j) pin(zb[j]);
#pragma unroll
                    for (int j = 0; j < J; ++j) A[T][j][r] = zb[j];
                    const float bs = row_sum16(zb[0]);
                    if (p == 0 && n < W) PINN_ADD_S(&gbias[n], bs);
                    PINN_SB();
                }
            PINN_AT_FETCH();
            if (TAIL) {
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    float y[J], w[K + 1], yb[J], zb[J];
                    tanh_jet_fwd<K, KT>(At[e], y, w);
#pragma unroll
                    for (int j = 0; j < J; ++j) yb[j] = 0.0f;
#pragma unroll
                    for (int o = 0; o < NO; ++o)
                        if (o < nout) {
                            const float wo = io[C::O_WOUTT + o * 4 + e];
                            float v = 0.0f;
#pragma unroll
                            for (int j = 0; j < J; ++j) {
                                v = fmaf(ub[o][j], y[j], v);
                                yb[j] = fmaf(wo, ub[o][j], yb[j]);
                            }
                            v = row_sum16(v);
                            if (lane == 0) PINN_ADD_S(&p_wout[o * W + NM + e], v);
                        }
                    tanh_jet_bwd<K, KT>(At[e], y, w, yb, zb);
#pragma unroll
                    for (int j = 0; j < J; ++j) At[e][j] = zb[j];
                    const float bs = row_sum16(zb[0]);
                    if (lane == 0) PINN_ADD_S(&gbias[NM + e], bs);
                }
            }
        }

        // ---- Linear layers l = NH-1 .. 1 (hidden -> hidden), each followed by the tanh layer below it ----
        for (int l = NH - 1; l >= 1; --l) {
            PINN_REFRESH();
            // on entry A/At = Zb (adjoint of z_l);  Linear layer l maps y_{l-1} -> z_l
            const float *img = lds + L.hh + (l - 1) * C::HH;
            const int yprev = (l - 1) * LAYER_STASH + ZY;
            float *gtc = p_tc(l);                                                 // tail columns of Wb_l: [o*TCS + e]
            float *gtr = p_tr(l);                                                 // tail rows of Wb_l: [e*W + i]
            float *wim = lds + L.wimg + (l - 1) * (NT * NT * TILE);              // main block of Wb_l (LDS images)

            // Phase A: Wb[o][i] += sum_{p,k} Zb_k[o][p] * Y_k[i][p]   (reduction over points)
            {
                // The tail rows of Zb are parked in LDS as [component][point]: their weight gradient against
                // the tiled y_{l-1} is formed in the tanh layer below, where y_{l-1} is recomputed anyway.
                if (TAIL) {
                    if (q == 0) {
#pragma unroll
                        for (int e = 0; e < R; ++e)
#pragma unroll
                            for (int j = 0; j < J; ++j) ztl[(e * ZTL_J + j) * 16 + p] = At[e][j];
                    }
                }
                if constexpr (GROUPED) {
                    // One row of accumulator tiles (output neurons 16To..16To+15) at a time: NT tiles of
                    // accumulators instead of NT*NT, at the price of re-reading y from the stash NT times.
                    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(slabp, 0, (d.P + MAXF) * 4, 0x00020000);
                    const int gvoff = (4 * q * W + p) * 4;
                    const int gsoff = d.woff[l] * 4;
#ifndef PINN_GR
#define PINN_GR 3
#endif
                    constexpr int GR = (NT % PINN_GR == 0) ? PINN_GR : (NT % 2 == 0 ? 2 : 1);     // rows of tiles per pass
#pragma unroll
                    for (int Tg = 0; Tg < NT; Tg += GR) {
                        f32x4 wrow[GR][NT];
#pragma unroll
                        for (int t = 0; t < GR; ++t)
#pragma unroll
                            for (int Ti = 0; Ti < NT; ++Ti) wrow[t][Ti] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            float bopg[NT][4];
                            // (scalar base re-derived per use: hoisted, the NT*J*4 stash offsets of a layer
                            // would stay live in SGPRs over all passes)
                            int ypo = yprev + j * TILE;
                            asm volatile("" : "+s"(ypo));
#pragma unroll
                            for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                                for (int s = 0; s < 4; ++s) bopg[Ti][s] = sb_load1(srs, vyn, ypo + Ti * J * TILE + s * 16);
#pragma unroll
                            for (int t = 0; t < GR; ++t) {
                                // the one staging tile is re-used row after row (LDS operations of a wave are in order)
                                *(f32x4 *)&stage[p * STAGE_ROW + 4 * q] = A[Tg + t][j];
                                PINN_LDS_FENCE();
                                float aopg[4];
#pragma unroll
                                for (int s = 0; s < 4; ++s) aopg[s] = stage[(4 * q + s) * STAGE_ROW + p];
#pragma unroll
                                for (int s = 0; s < 4; ++s) pin(aopg[s]);
                                PINN_SB();
#pragma unroll
                                for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                                    for (int s = 0; s < 4; ++s) wrow[t][Ti] = mfma4(aopg[s], bopg[Ti][s], wrow[t][Ti]);
                                PINN_SB();
                            }
                            PINN_JB();
                        }
                        {
#pragma unroll
                            for (int t = 0; t < GR; ++t) {
                                // scalar offset of this row of tiles, re-derived here: formed once per layer the
                                // per-tile offsets would all be live (and spilled) across the whole phase
                                int gso = gsoff + 16 * (Tg + t) * W * 4;
                                asm volatile("" : "+s"(gso));
#pragma unroll
                                for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        if (TAIL || (16 * (Tg + t) + 4 * q + r < W && 16 * Ti + p < W))
                                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wrow[t][Ti][r], grs, gvoff,
                                                                                            gso + (r * W + 16 * Ti) * 4, 0);
                            }
                        }
                    }
                } else {
                f32x4 wacc[NT][NT];
#pragma unroll
                for (int To = 0; To < NT; ++To)
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) wacc[To][Ti] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                // Software pipeline over the jet components: while the 36 MFMAs of component j run, the
                // operands of j+1 are already on their way (stash b128 loads; Zb through the staging image).
                //  B operand: lane (n = p, kk = q) gets y[16Ti + n][points 4kk .. 4kk+3]
                //  A operand: Zb through the staging image [point][neuron] -> lane (m = p, kk = q) reads
                //             Zb[16To + m][point 4kk + s]
                float bop[2][NT][4];
#pragma unroll
                for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                    for (int s = 0; s < 4; ++s) bop[0][Ti][s] = sb_load1(srs, vyn, yprev + (Ti * J + 0) * TILE + s * 16);
                PINN_JB();
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int c = j & 1, nx = c ^ 1;
                    // the stash loads of component j+1 are in flight while component j is processed
                    if (j + 1 < J) {
#pragma unroll
                        for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                            for (int s = 0; s < 4; ++s) bop[nx][Ti][s] = sb_load1(srs, vyn, yprev + (Ti * J + j + 1) * TILE + s * 16);
                    }
                    // A operand: Zb through the staging image [point][neuron] -> lane (m = p, kk = q) reads
                    // Zb[16To + m][point 4kk + s]; STG tiles per pass (all NT unless LDS is short)
#pragma unroll
                    for (int g0 = 0; g0 < NT; g0 += STG) {
#pragma unroll
                        for (int t = 0; t < STG; ++t)
                            if (g0 + t < NT) *(f32x4 *)&stage[t * STAGE_TILE + p * STAGE_ROW + 4 * q] = A[g0 + t][j];
                        PINN_LDS_FENCE();
                        float aop[STG][4];
#pragma unroll
                        for (int t = 0; t < STG; ++t)
#pragma unroll
                            for (int s = 0; s < 4; ++s) aop[t][s] = stage[t * STAGE_TILE + (4 * q + s) * STAGE_ROW + p];
#pragma unroll
                        for (int t = 0; t < STG; ++t)
#pragma unroll
                            for (int s = 0; s < 4; ++s) pin(aop[t][s]);
                        PINN_SB();
#pragma unroll
                        for (int t = 0; t < STG; ++t)
                            if (g0 + t < NT) {
#pragma unroll
                                for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                                    for (int s = 0; s < 4; ++s)
                                        wacc[g0 + t][Ti] = mfma4(aop[t][s], bop[c][Ti][s], wacc[g0 + t][Ti]);
                            }
                        if (g0 + STG < NT) PINN_JB();
                    }
                    PINN_JB();
                }
                // accumulator (To,Ti): lane (i = p, q) register r holds Wb[16To + 4q + r][16Ti + p]
                if (WGLOBAL || L.wglobal) {
                    {
                        // buffer atomics: one per-lane offset VGPR, everything else scalar
                        const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(slabp, 0, (d.P + MAXF) * 4, 0x00020000);
                        const int gvoff = (4 * q * W + p) * 4;
                        const int gsoff = d.woff[l] * 4;
#pragma unroll
                        for (int To = 0; To < NT; ++To)
#pragma unroll
                            for (int Ti = 0; Ti < NT; ++Ti)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (TAIL || (16 * To + 4 * q + r < W && 16 * Ti + p < W))
                                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wacc[To][Ti][r], grs, gvoff,
                                                                                        gsoff + ((16 * To + r) * W + 16 * Ti) * 4, 0);
                    }
                } else {
                    int *lk = (int *)(lds + L.locks) + (l - 1);
                    lds_lock(lk, lane);
#pragma unroll
                    for (int To = 0; To < NT; ++To)
#pragma unroll
                        for (int Ti = 0; Ti < NT; ++Ti) {
                            f32x4 *dst = (f32x4 *)&wim[(To * NT + Ti) * TILE + lane * 4];
                            *dst = *dst + wacc[To][Ti];
                        }
                    lds_unlock(lk, lane);
                }
                }
                PINN_SB();
                if (TAIL) {
                    // tiled rows x tail columns: Wb[o][NM+e] = sum_{p,k} Zb_k[o][p] * ytail_k[e][p],
                    // and the tail x tail corner
#pragma unroll
                    for (int e = 0; e < R; ++e) {
                        float yp[J];
#pragma unroll
                        for (int j = 0; j < J; ++j) yp[j] = sb_load1(srs, vp, yprev + ZY + TS + (e * J + j) * 16);
#pragma unroll
                        for (int e1 = 0; e1 < R; ++e1) {
                            float v = 0.0f;
#pragma unroll
                            for (int j = 0; j < J; ++j) v = fmaf(At[e1][j], yp[j], v);
                            v = row_sum16(v);
                            if (lane == 0) PINN_ADD_S(&gtr[e1 * W + NM + e], v);
                        }
#pragma unroll
                        for (int To = 0; To < NT; ++To)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float v = 0.0f;
#pragma unroll
                                for (int j = 0; j < J; ++j) v = fmaf(A[To][j][r], yp[j], v);
                                v = row_sum16(v);
                                if (p == 0) PINN_ADD_S(&gtc[(16 * To + 4 * q + r) * TCS + e], v);
                            }
                    }
                }
            }

            PINN_SB();
            PINN_REFRESH();

            // Phase B: Yb^T[i][p] = sum_o W[o][i] Zb^T[o][p], one jet component at a time, in place.
            {
                float Ht[RR][J];
#pragma unroll
                for (int e = 0; e < RR; ++e)
#pragma unroll
                    for (int j = 0; j < J; ++j) Ht[e][j] = 0.0f;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    f32x4 acc[NT];
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) {
                        acc[Ti] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                        if (TAIL) {
                            // tail outputs' contribution: part of the accumulator's initial value
#pragma unroll
                            for (int e = 0; e < R; ++e) {
                                const f32x4 wr = *(const f32x4 *)&img[C::O_WTR + e * NM + 16 * Ti + 4 * q];
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[Ti][r] = fmaf(wr[r], At[e][j], acc[Ti][r]);
                            }
                        }
                        // A fragments: lane (m = p, q) holds W[16To + 4q + r][16Ti + m].  All 4*NT reads of this
                        // output tile are issued together and waited for once.
                        float wf[NT][4];
#pragma unroll
                        for (int To = 0; To < NT; ++To)
#pragma unroll
                            for (int r = 0; r < 4; ++r) wf[To][r] = img[(16 * To + 4 * q + r) * S + 16 * Ti + p];
#pragma unroll
                        for (int To = 0; To < NT; ++To)
#pragma unroll
                            for (int r = 0; r < 4; ++r) pin(wf[To][r]);
#pragma unroll
                        for (int To = 0; To < NT; ++To)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (16 * To + r < W) acc[Ti] = mfma4(wf[To][r], A[To][j][r], acc[Ti]);
                        PINN_SB();
                    }
                    PINN_SB();
                    if (TAIL) {
                        // adjoint of the tail inputs: sum over tiled outputs (cross-quad) + tail outputs
#pragma unroll
                        for (int e2 = 0; e2 < R; ++e2) {
                            float h = 0.0f;
#pragma unroll
                            for (int To = 0; To < NT; ++To) {
                                const f32x4 wc = *(const f32x4 *)&img[C::O_WTC + e2 * NM + 16 * To + 4 * q];
#pragma unroll
                                for (int r = 0; r < 4; ++r) h = fmaf(wc[r], A[To][j][r], h);
                            }
                            h = quad_rows_sum(h);
#pragma unroll
                            for (int e = 0; e < R; ++e) h = fmaf(img[C::O_WTT + e * 4 + e2], At[e][j], h);
                            Ht[e2][j] = h;
                            PINN_SB();
                        }
                    }
#pragma unroll
                    for (int Ti = 0; Ti < NT; ++Ti) A[Ti][j] = acc[Ti];
                    PINN_JB();
                }
#pragma unroll
                for (int e = 0; e < RR; ++e)
#pragma unroll
                    for (int j = 0; j < J; ++j) At[e][j] = Ht[e][j];
            }


            // tanh layer hl = l-1: A/At = adjoint of y_hl -> adjoint of z_hl, in place
            PINN_REFRESH();
            const int hl = l - 1;
            const int zs = hl * LAYER_STASH;
            float *gbias = p_b(hl);
            if (hl == 0) PINN_XT_FETCH();
            // z comes back from the stash neuron by neuron, PINN_ZPF neurons ahead of its use (one wave has
            // only one other wave on its SIMD to hide the load latency behind); y and w are recomputed.
            // The first tanh layer recomputes z instead.  Two instantiations of the same body, so that
            // neither carries the other's registers (prefetch ring vs. first-layer weights).
            auto rev_main = [&](auto first_tag) __attribute__((always_inline)) {
                constexpr bool FIRST = decltype(first_tag)::value;
                constexpr int NQ = 4 * NT;
                constexpr int RING = FIRST ? 1 : PINN_ZPF + 1;
                float zr[RING][J];
                auto zfetch = [&](int nidx, float (&z)[J]) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < J; ++j) z[j] = sb_load1(srs, vyn, zs + ((nidx >> 2) * J + j) * TILE + (nidx & 3) * 16);
                };
                if (!FIRST) {
#pragma unroll
                    for (int k = 0; k < PINN_ZPF; ++k)
                        if (k < NQ) zfetch(k, zr[k % RING]);
                }
                PINN_SB();
#pragma unroll
                for (int T = 0; T < NT; ++T) {
                    W0 w0;
                    if (FIRST) w0 = w0_tile(T);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float z[J], y[J], w[K + 1], yb[J], zb[J];
                        const int nidx = 4 * T + r;
                        const int n = 16 * T + 4 * q + r;
                        if (!FIRST) {
                            if (nidx + PINN_ZPF < NQ) zfetch(nidx + PINN_ZPF, zr[(nidx + PINN_ZPF) % RING]);
#pragma unroll
                            for (int j = 0; j < J; ++j) z[j] = zr[nidx % RING][j];
                        } else {
                            z0_main(w0, r, z);
                        }
#pragma unroll
                        for (int j = 0; j < J; ++j) yb[j] = A[T][j][r];
                        tanh_jet_fwd<K, KT, FIRST ? ZN0 : K, FIRST ? ZTN0 : KT>(z, y, w);
                        if (TAIL) {
                            // tail rows of Wb_l against this neuron of y_hl (Zbtail of layer l: parked in LDS)
#pragma unroll
                            for (int e = 0; e < R; ++e) {
                                float v = 0.0f;
#pragma unroll
                                for (int j = 0; j < J; ++j) v = fmaf(ztl[(e * ZTL_J + j) * 16 + p], y[j], v);
                                v = row_sum16(v);
                                if (p == 0) PINN_ADD_S(&gtr[e * W + n], v);
                            }
                        }
                        tanh_jet_bwd<K, KT, FIRST ? ZN0 : K, FIRST ? ZTN0 : KT>(z, y, w, yb, zb);
#pragma unroll
                        for (int j = 0; j < J; ++j) pin(zb[j]);
#pragma unroll
                        for (int j = 0; j < J; ++j) A[T][j][r] = zb[j];
                        const float bs = row_sum16(zb[0]);
                        if (p == 0 && n < W) PINN_ADD_S(&gbias[n], bs);
                        PINN_SB();
                    }
                }
            };
            PINN_AT_PARK();
            PINN_SB();
            if (hl > 0) rev_main(std::false_type{});
            else rev_main(std::true_type{});
            PINN_AT_FETCH();
            PINN_REFRESH();
            if (TAIL) {
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    float z[J], y[J], w[K + 1], zb[J];
                    if (hl > 0) {
#pragma unroll
                        for (int j = 0; j < J; ++j) z[j] = sb_load1(srs, vp, zs + 2 * ZY + (e * J + j) * 16);
                        tanh_jet_fwd<K, KT>(z, y, w);
                        tanh_jet_bwd<K, KT>(z, y, w, At[e], zb);
                    } else {
                        z0_tail(e, z);
                        tanh_jet_fwd<K, KT, ZN0, ZTN0>(z, y, w);
                        tanh_jet_bwd<K, KT, ZN0, ZTN0>(z, y, w, At[e], zb);
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) At[e][j] = zb[j];
                    const float bs = row_sum16(zb[0]);
                    if (lane == 0) PINN_ADD_S(&gbias[NM + e], bs);
                }
            }
        }


        // ---- first layer gradients from A/At = adjoint of z_0 ----
        PINN_REFRESH();
        PINN_XT_FETCH();
        {
            float *gw0 = p_w0;
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                // (coordinates re-read from their LDS parking slot per tile of neurons: kept in registers over
                // the whole phase they are spilled to scratch and reloaded neuron by neuron)
                if (C::PARK) { asm volatile("" ::: "memory"); hx = xtp[lane]; ht = xtp[64 + lane]; }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g0 = A[T][0][r];
                    float gx = g0 * hx, gt = g0 * ht;
                    if (K >= 1) gx = fmaf(A[T][K >= 1 ? 1 : 0][r], sx, gx);
                    if (KT >= 1) gt = fmaf(A[T][KT >= 1 ? K + 1 : 0][r], stt, gt);
                    gx = row_sum16(gx);
                    gt = row_sum16(gt);
                    const int n = 16 * T + 4 * q + r;
                    if (p == 0 && n < W) {
                        PINN_ADD_S(&gw0[2 * n], gx);
                        PINN_ADD_S(&gw0[2 * n + 1], gt);
                    }
                }
            }
            if (TAIL) {
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    const float g0 = At[e][0];
                    float gx = g0 * hx, gt = g0 * ht;
                    if (K >= 1) gx = fmaf(At[e][K >= 1 ? 1 : 0], sx, gx);
                    if (KT >= 1) gt = fmaf(At[e][KT >= 1 ? K + 1 : 0], stt, gt);
                    gx = row_sum16(gx);
                    gt = row_sum16(gt);
                    if (lane == 0) {
                        PINN_ADD_S(&gw0[2 * (NM + e)], gx);
                        PINN_ADD_S(&gw0[2 * (NM + e) + 1], gt);
                    }
                }
            }
        }
    }

    if (dyn && lane0 == 0) {
        // every wave's last request has returned before it counts itself out: after the last one nothing touches
        // the counters any more
        if (atomicAdd(&a.tile_ctr[1], 1) == (int)nwaves - 1) {
            atomicExch(&a.tile_ctr[0], 0);
            atomicExch(&a.tile_ctr[1], 0);
        }
    }
    PINN_REFRESH();
    if (MODE != MODE_FWD) {
        if (MODE == MODE_LOSS) {
            for (int c = 0; c < nf; ++c) {
                const float s = wave_sum(lsum[c]);
                if (lane == 0) atomicAdd(&gs[L.gs_loss + c], s);
            }
        }
        __syncthreads();
        {
            // parameter order out of the compact block and the accumulator images
            float *slab = a.slab + (long long)blockIdx.x * (P + MAXF);
            for (int i = tid; i < P + MAXF; i += MFMA_BLOCK) {
                float v;
                if (i >= P) {
                    v = gs[L.gs_loss + (i - P)];
                } else {
                    int l = 0;
                    while (l + 1 < n_lin && i >= d.woff[l + 1]) ++l;
                    const int fi = d.widths[l], fo = d.widths[l + 1];
                    const int rel = i - d.woff[l];
                    if (rel >= fi * fo) {                       // bias
                        const int n = rel - fi * fo;
                        v = (l == n_lin - 1) ? gs[L.gs_out + nout * W + n] : gblock(l)[n];
                    } else {
                        const int o = rel / fi, c = rel - o * fi;
                        if (l == 0) v = gs[rel];
                        else if (l == n_lin - 1) v = gs[L.gs_out + rel];
                        else if (o >= NM) v = gblock(l)[W + (o - NM) * W + c];
                        else if (c >= NM) v = gblock(l)[W + R * W + o * R + (c - NM)];
                        else if (WGLOBAL || L.wglobal) continue;   // went to the (pre-zeroed) slab directly, by hardware atomics
                        else {
                            // accumulator (To,Ti): lane (p, q) register r holds Wb[16To + 4q + r][16Ti + p]
                            const int To = o >> 4, q4 = (o & 15) >> 2, r = o & 3, Ti = c >> 4, pp = c & 15;
                            v = lds[L.wimg + (((l - 1) * NT + To) * NT + Ti) * TILE + (q4 * 16 + pp) * 4 + r];
                        }
                    }
                }
                slab[i] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ host side
// hidden widths with an MFMA instantiation; anything else runs on the VALU path
#if defined(PINN_DEV_KDV7_ONLY)
#define PINN_MFMA_WIDTHS(X) X(50)
#elif defined(PINN_DEV_GNSS100_ONLY)
#define PINN_MFMA_WIDTHS(X) X(100)
#elif defined(PINN_DEV_CH_ONLY)
#define PINN_MFMA_WIDTHS(X) X(50) X(64)
#elif defined(PINN_DEV_BURGERS_ONLY)
#define PINN_MFMA_WIDTHS(X) X(20) X(50)
#else
#define PINN_MFMA_WIDTHS(X) X(20) X(50) X(64) X(100)
#endif

template <int W> size_t mfma_lds_bytes(const KDesc &kd, int J) { return (size_t)mfma_layout<W>(kd, J).total * sizeof(float); }

// waves per workgroup / whether the kernel accumulates straight into the (pre-zeroed) slab
inline int mfma_waves(const KDesc &kd, int J)
{
    switch (kd.widths[1]) {
#define X(w) case w: return WidthCfg<w>::waves(J);
        PINN_MFMA_WIDTHS(X)
#undef X
    default: return MFMA_MAX_WAVES;
    }
}
inline bool mfma_wglobal(const KDesc &kd, int J)
{
    switch (kd.widths[1]) {
#define X(w) case w: return mfma_layout<w>(kd, J).wglobal;
        PINN_MFMA_WIDTHS(X)
#undef X
    default: return false;
    }
}

inline bool mfma_supported(const KDesc &kd, int J)
{
    if (kd.n_lin < 2) return false;
    const int W = kd.widths[1];
    for (int l = 1; l < kd.n_lin; ++l)
        if (kd.widths[l] != W) return false;
    size_t lds = 0;
    switch (W) {
#define X(w) case w: lds = mfma_lds_bytes<w>(kd, J); break;
        PINN_MFMA_WIDTHS(X)
#undef X
    default: return false;
    }
    return lds <= (size_t)MFMA_LDS_LIMIT;
}

// floats of stash one wave needs for equation jets J
template <int W> size_t mfma_stash_per_wave(const KDesc &kd, int J)
{
    typedef WidthCfg<W> C;
    const size_t ZY = (size_t)C::NT * J * 256;
    const size_t TS = (size_t)((C::RR * J * 16 + 63) / 64) * 64;
    return (size_t)(kd.n_lin - 1) * (2 * ZY + 2 * TS) + 128;
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

template <int EQ, int W, int MODE>
int launch_mfma_t(float *stash, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    const size_t ldsb = mfma_lds_bytes<W>(kd, EqTraits<EQ>::K + 1 + EqTraits<EQ>::KT);
    auto kern = k_mfma<EQ, W, MODE>;
    if (ldsb > 48 * 1024) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) {
            (void)hipGetLastError();
            return PINN_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WidthCfg<W>::block(EqTraits<EQ>::K + 1 + EqTraits<EQ>::KT)), ldsb, st, kd, a, stash);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

template <int EQ, int MODE>
int launch_mfma_w(float *stash, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    switch (kd.widths[1]) {
#define X(w) case w: return launch_mfma_t<EQ, w, MODE>(stash, kd, a, grid, st);
        PINN_MFMA_WIDTHS(X)
#undef X
    default: return PINN_ERR_UNSUPPORTED;
    }
}

template <int MODE>
int launch_mfma_eq(float *stash, int eq, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    switch (eq) {
#if defined(PINN_DEV_GNSS100_ONLY)
    case EQ_GNSS:    return launch_mfma_w<EQ_GNSS, MODE>(stash, kd, a, grid, st);
#elif defined(PINN_DEV_CH_ONLY)
    case EQ_CH:      return launch_mfma_w<EQ_CH, MODE>(stash, kd, a, grid, st);
#elif defined(PINN_DEV_BURGERS_ONLY)
    case EQ_BURGERS: return launch_mfma_w<EQ_BURGERS, MODE>(stash, kd, a, grid, st);
#else
#ifndef PINN_DEV_KDV7_ONLY
    case EQ_NONE:    return launch_mfma_w<EQ_NONE, MODE>(stash, kd, a, grid, st);
    case EQ_BURGERS: return launch_mfma_w<EQ_BURGERS, MODE>(stash, kd, a, grid, st);
    case EQ_KDV5:    return launch_mfma_w<EQ_KDV5, MODE>(stash, kd, a, grid, st);
    case EQ_CH:      return launch_mfma_w<EQ_CH, MODE>(stash, kd, a, grid, st);
    case EQ_GNSS:    return launch_mfma_w<EQ_GNSS, MODE>(stash, kd, a, grid, st);
    case EQ_LPD:     return launch_mfma_w<EQ_LPD, MODE>(stash, kd, a, grid, st);
    case EQ_NLSREG:  return launch_mfma_w<EQ_NLSREG, MODE>(stash, kd, a, grid, st);
#endif
    case EQ_KDV7:    return launch_mfma_w<EQ_KDV7, MODE>(stash, kd, a, grid, st);
#endif
    default: return PINN_ERR_INVALID_DESC;
    }
}

}  // namespace pinn
