// C-ABI of the PINN collocation engine (include/pinnjet.h) and kernel dispatch.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>

#include "../../include/pinnjet.h"
#include "jet_math.hpp"
#include "kernel_valu.hpp"
#include "pinn_common.hpp"
#ifdef PINN_WITH_MFMA
#include "kernel_mfma.hpp"
#endif

using namespace pinn;

struct pinn_handle {
    pinn_desc desc;
    KDesc kd;
    int K, KT, J, nf;
    int kernel;          // resolved PINN_KERNEL_*
    int device;
    int n_cu;
    // workspace (sized by pinn_reserve or grown on demand, never shrunk)
    float *jets = nullptr;       // VALU: per-thread jet scratch; MFMA: per-wave stash
    size_t jets_bytes = 0;
    float *slab = nullptr;       // per-workgroup partial [grid][P+2]
    size_t slab_bytes = 0;
    int *tile_ctr = nullptr;     // MFMA kernels: the two counters tiles are handed out from (zero between launches)
    int mfma_grid = 0;           // persistent grid of the MFMA kernels (<= n_cu)
    int spare_cus = 32;          // CUs a small launch leaves free for launches on other streams: one per shader
                                 // engine of every XCD (with 8 or 16, a 31-workgroup fill still queued behind the launch)
};

// ---- the scratch of pinn_forward_vjp_static -------------------------------------------------------
// Device memory that belongs to the code object, one copy per GPU the library runs on, held until the
// process ends.  The NAME is a host-side stand-in: host code only ever uses the per-device address
// that hipGetSymbolAddress returned (g_static_ptr), as kernel argument and as memset target.
static __device__ __align__(256) unsigned char scratch[PINN_STATIC_SCRATCH_BYTES];

namespace {

constexpr int MAX_DEVICES = 64;
void *g_static_ptr[MAX_DEVICES];      // per device: address of `scratch` on that device (0 = not looked up yet)
std::mutex g_static_mu;

// Look the static scratch array up on the CURRENT device (once per device).  Called from pinn_create,
// i.e. at set-up, never from the no-allocation entry point's steady state.
int static_scratch_lookup(int device)
{
    if (device < 0 || device >= MAX_DEVICES) return PINN_ERR_INVALID_DESC;
    std::lock_guard<std::mutex> lk(g_static_mu);
    if (g_static_ptr[device]) return PINN_OK;
    void *ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) { (void)hipGetLastError(); return PINN_ERR_HIP; }
    g_static_ptr[device] = ptr;
    return PINN_OK;
}

// where a call's scratch memory comes from: the handle's own buffers (grown on demand) or a fixed
// region (the static array) that the grid is sized to fit
struct FixedWs {
    unsigned char *base;
    size_t bytes;
};

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int eq_orders(int eq, int nout, int *K, int *KT, int *nf)
{
    switch (eq) {
    case PINN_EQ_NONE:    *K = 0; *KT = 0; *nf = nout; return 0;
    case PINN_EQ_BURGERS: *K = 2; *KT = 1; *nf = 1; return nout == 1 ? 0 : -1;
    case PINN_EQ_KDV5:    *K = 5; *KT = 1; *nf = 1; return nout == 1 ? 0 : -1;
    case PINN_EQ_KDV7:    *K = 7; *KT = 1; *nf = 1; return nout == 1 ? 0 : -1;
    case PINN_EQ_CH:      *K = 3; *KT = 3; *nf = 1; return nout == 1 ? 0 : -1;
    case PINN_EQ_GNSS:    *K = 3; *KT = 1; *nf = 2; return nout == 2 ? 0 : -1;
    case PINN_EQ_LPD:     *K = 4; *KT = 1; *nf = 2; return nout == 2 ? 0 : -1;
    case PINN_EQ_NLSREG:  *K = 2; *KT = 1; *nf = 3; return nout == 2 ? 0 : -1;
    default: return -1;
    }
}

int ensure(float **buf, size_t *have, size_t need)
{
    if (need <= *have) return PINN_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    if (hipMalloc((void **)buf, need) != hipSuccess) { (void)hipGetLastError(); return PINN_ERR_ALLOC; }
    *have = need;
    return PINN_OK;
}

// Sum the per-workgroup partial slabs in a fixed order (deterministic for a
// given grid) and accumulate into the caller's gradient / loss buffers.
// zero the per-workgroup slabs of the wide shapes (they are accumulated into with atomics).  A kernel, not
// hipMemsetAsync: as a memset NODE of a captured HIP graph the clear was not reliably ordered before the
// kernel on replays (the main blocks of the hidden-layer gradients came back accumulated / non-finite from
// the second replay on); a kernel node in the same stream is.
__global__ __launch_bounds__(256) void k_zero_floats(float *p, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.0f;
}

// 16 waves per workgroup, each over a 16th of the slabs for 64 consecutive entries (the loads of a wave's loop are
// independent and issued ahead of the float64 adds; the order of the adds is fixed)
constexpr int REDUCE_WAVES = 16;
__global__ __launch_bounds__(64 * REDUCE_WAVES) void k_reduce_slabs(const float *slab, int nblk, int P, float *grad_accum,
                                                                    float *loss_accum, int nf)
{
    __shared__ double part[REDUCE_WAVES][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (i < P + MAXF) {
        const int per = (nblk + REDUCE_WAVES - 1) / REDUCE_WAVES;
        const int b0 = wv * per, b1 = min(nblk, b0 + per);
        const float *src = slab + i;
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            const float v0 = src[(size_t)b * (P + MAXF)], v1 = src[(size_t)(b + 1) * (P + MAXF)];
            const float v2 = src[(size_t)(b + 2) * (P + MAXF)], v3 = src[(size_t)(b + 3) * (P + MAXF)];
            s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
        }
        for (; b < b1; ++b) s += (double)src[(size_t)b * (P + MAXF)];
    }
    part[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && i < P + MAXF) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < REDUCE_WAVES; ++w) tot += part[w][lane];
        if (i < P) {
            if (grad_accum) grad_accum[i] += (float)tot;
        } else if (loss_accum && (i - P) < nf) {
            loss_accum[i - P] += (float)tot;
        }
    }
}

// the residual polynomial alone (Residual<EQ>::eval, the `return u_t + ...` line of net_f) on
// caller-supplied true derivatives [n][nout][J]: one point per thread
template <int EQ>
__global__ __launch_bounds__(256) void k_residual_from_derivs(KDesc d, const float *derivs, long long n, float *f_out)
{
    constexpr int K = EqTraits<EQ>::K, KT = EqTraits<EQ>::KT, J = K + 1 + KT;
    const int nout = d.nout;
    const int nf = EqTraits<EQ>::NF ? EqTraits<EQ>::NF : (nout < MAXOUT ? nout : MAXOUT);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        float D[MAXOUT][J], T[MAXOUT][J], f[MAXF] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float v = (o < nout) ? derivs[(p * nout + o) * J + j] : 0.0f;
                if (j <= K) { D[o][j] = v; T[o][j] = 0.0f; }
                else { T[o][j - K - 1] = v; }
            }
#pragma unroll
        for (int o = 0; o < MAXOUT; ++o)
#pragma unroll
            for (int j = KT; j < J; ++j) T[o][j] = 0.0f;
        Residual<EQ>::eval(D, T, d.coef, nout, f);
        for (int c = 0; c < nf; ++c) f_out[p * nf + c] = f[c];
    }
}

template <int EQ, int MODE>
int launch_valu_t(float *jets, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    const size_t lds = (MODE == MODE_FWD) ? 0 : (size_t)(kd.P + MAXF) * sizeof(float);
    auto kern = k_valu<EQ, MODE>;
    if (lds > 48 * 1024) {
        if (lds > 160 * 1024) return PINN_ERR_UNSUPPORTED;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return PINN_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(VALU_BLOCK), lds, st, kd, a, jets);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

template <int MODE>
int launch_valu_eq(float *jets, int eq, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    switch (eq) {
    case PINN_EQ_NONE:    return launch_valu_t<EQ_NONE, MODE>(jets, kd, a, grid, st);
    case PINN_EQ_BURGERS: return launch_valu_t<EQ_BURGERS, MODE>(jets, kd, a, grid, st);
    case PINN_EQ_KDV5:    return launch_valu_t<EQ_KDV5, MODE>(jets, kd, a, grid, st);
    case PINN_EQ_KDV7:    return launch_valu_t<EQ_KDV7, MODE>(jets, kd, a, grid, st);
    case PINN_EQ_CH:      return launch_valu_t<EQ_CH, MODE>(jets, kd, a, grid, st);
    case PINN_EQ_GNSS:    return launch_valu_t<EQ_GNSS, MODE>(jets, kd, a, grid, st);
    case PINN_EQ_LPD:     return launch_valu_t<EQ_LPD, MODE>(jets, kd, a, grid, st);
    case PINN_EQ_NLSREG:  return launch_valu_t<EQ_NLSREG, MODE>(jets, kd, a, grid, st);
    default: return PINN_ERR_INVALID_DESC;
    }
}

// What one call needs: the grid it would like and, per workgroup of it, the bytes of jet scratch
// (VALU: [slot][thread] jets; MFMA: the per-wave stash of the reverse sweep) and of partial slab.
struct Need {
    int grid;
    size_t jets_per_wg, slab_per_wg;
    int nf;
    bool zero_slab;     // wide MFMA layers add into the slab with memory-side atomics: it has to start at zero
};


int call_need(const pinn_handle *h, int eq, int mode, long long n, Need *nd)
{
    int K, KT, nf;
    if (eq_orders(eq, h->kd.nout, &K, &KT, &nf)) return PINN_ERR_INVALID_DESC;
    const int J = K + 1 + KT;
    nd->nf = nf;
    nd->zero_slab = false;
    nd->slab_per_wg = (mode != MODE_FWD) ? (size_t)(h->kd.P + MAXF) * sizeof(float) : 0;
#ifdef PINN_WITH_MFMA
    if (h->kernel == PINN_KERNEL_MFMA) {
        const long long ntiles = (n + 15) / 16;
        const int waves = mfma_waves(h->kd, J);
        // one workgroup per CU at most; with fewer tiles than that, one tile per workgroup (the kernel deals
        // tiles to workgroups first, to the waves of a workgroup second).  A launch whose tiles all get a wave
        // of their own anyway (the scripts' N_f <= 20 000) leaves four CUs per XCD free: a training closure runs
        // its small data-term launches on another stream beside this one, and a workgroup of theirs cannot
        // share a CU with one of these (LDS) - without a free CU they would queue up behind the whole launch.
        const int SPARE_CUS = h->spare_cus;
        int cap = h->mfma_grid;
        if (cap > 4 * SPARE_CUS && ntiles <= (long long)(cap - SPARE_CUS) * waves) cap -= SPARE_CUS;
        nd->grid = (int)(ntiles < cap ? ntiles : cap);
        // ... and the small launches themselves (the IC / boundary batch: a few hundred points) pack four tiles
        // per workgroup (one per SIMD), so that they fit into those free CUs instead of asking for one CU per tile
        if (cap > 4 * SPARE_CUS && ntiles <= 8 * SPARE_CUS) {
            const long long packed = (ntiles + 3) / 4;
            nd->grid = (int)(packed < SPARE_CUS ? packed : SPARE_CUS);
        }
        nd->jets_per_wg = (mode != MODE_FWD) ? mfma_stash_floats(h->kd, J) * sizeof(float) * waves : 0;
        nd->zero_slab = (mode != MODE_FWD) && mfma_wglobal(h->kd, J);
        return PINN_OK;
    }
#endif
    const long long need_blocks = (n + VALU_BLOCK - 1) / VALU_BLOCK;
    nd->grid = (int)(need_blocks < h->n_cu ? need_blocks : h->n_cu);
    nd->jets_per_wg = (size_t)valu_slots(h->kd, J).total * VALU_BLOCK * sizeof(float);
    return PINN_OK;
}

// one call of any kind on either kernel family
int run(pinn_handle *h, int eq, int mode, CallArgs a, float *grad_accum, float *loss_accum, void *stream,
        const FixedWs *fixed = nullptr)
{
    if (!h) return PINN_ERR_NULL;
    if (!a.theta || !a.x || !a.t) return PINN_ERR_NULL;
    if (a.n < 0 || a.stride < 1) return PINN_ERR_INVALID_DESC;
    if (a.n == 0) return PINN_OK;
    hipStream_t st = (hipStream_t)stream;
    Need nd;
    int rc = call_need(h, eq, mode, a.n, &nd);
    if (rc) return rc;
    int grid = nd.grid;
    float *jets, *slab;
    if (fixed) {
        // size the (persistent) grid to the fixed region; every kernel strides its tiles over whatever grid it gets
        const size_t per_wg = nd.jets_per_wg + nd.slab_per_wg;
        // (the region's last 256 bytes hold the tile counters: zero when the code object is loaded, put back to
        // zero by every launch that uses them)
        const long long fit = per_wg ? (long long)((fixed->bytes - 512) / per_wg) : grid;
        if (fit < 1) return PINN_ERR_TOO_LARGE;
        if (fit < grid) grid = (int)fit;
        jets = (float *)fixed->base;
        slab = (float *)(fixed->base + align256(nd.jets_per_wg * grid));
        a.tile_ctr = (int *)(fixed->base + fixed->bytes - 256);
    } else {
        rc = ensure(&h->jets, &h->jets_bytes, nd.jets_per_wg * grid);
        if (rc) return rc;
        rc = ensure(&h->slab, &h->slab_bytes, nd.slab_per_wg * grid);
        if (rc) return rc;
        jets = h->jets;
        slab = h->slab;
        a.tile_ctr = h->tile_ctr;
    }
    if (mode != MODE_FWD) a.slab = slab;
    if (nd.zero_slab) {
        const long long nfl = (long long)(nd.slab_per_wg / sizeof(float)) * grid;
        const long long zb = (nfl + 256 * 8 - 1) / (256 * 8);
        hipLaunchKernelGGL(k_zero_floats, dim3(zb < 1 ? 1 : (zb < 2048 ? (int)zb : 2048)), dim3(256), 0, st, slab, nfl);
    }
#ifdef PINN_WITH_MFMA
    if (h->kernel == PINN_KERNEL_MFMA) {
        switch (mode) {
        case MODE_FWD:  rc = launch_mfma_eq<MODE_FWD>(jets, eq, h->kd, a, grid, st); break;
        case MODE_VJP:  rc = launch_mfma_eq<MODE_VJP>(jets, eq, h->kd, a, grid, st); break;
        default:        rc = launch_mfma_eq<MODE_LOSS>(jets, eq, h->kd, a, grid, st); break;
        }
    } else
#endif
    {
        switch (mode) {
        case MODE_FWD:  rc = launch_valu_eq<MODE_FWD>(jets, eq, h->kd, a, grid, st); break;
        case MODE_VJP:  rc = launch_valu_eq<MODE_VJP>(jets, eq, h->kd, a, grid, st); break;
        default:        rc = launch_valu_eq<MODE_LOSS>(jets, eq, h->kd, a, grid, st); break;
        }
    }
    if (rc) return rc;
    if (mode != MODE_FWD) {
        // fixed-order sum of the per-workgroup partials into the caller's buffers
        const int rb = (h->kd.P + MAXF + 63) / 64;
        hipLaunchKernelGGL(k_reduce_slabs, dim3(rb), dim3(64 * REDUCE_WAVES), 0, st, slab, grid, h->kd.P, grad_accum, loss_accum, nd.nf);
        if (hipGetLastError() != hipSuccess) return PINN_ERR_HIP;
    }
    return PINN_OK;
}

}  // namespace

extern "C" {

int pinn_create(const pinn_desc *desc, pinn_handle **out)
{
    if (!desc || !out) return PINN_ERR_NULL;
    *out = nullptr;
    if (desc->n_lin < 2 || desc->n_lin > PINN_MAX_LAYERS) return PINN_ERR_INVALID_DESC;
    if (desc->widths[0] != 2) return PINN_ERR_INVALID_DESC;
    for (int l = 1; l <= desc->n_lin; ++l)
        if (desc->widths[l] < 1 || desc->widths[l] > PINN_MAX_WIDTH) return PINN_ERR_INVALID_DESC;
    const int nout = desc->widths[desc->n_lin];
    if (nout > PINN_MAX_OUT) return PINN_ERR_INVALID_DESC;
    if (!(desc->ub[0] > desc->lb[0]) || !(desc->ub[1] > desc->lb[1])) return PINN_ERR_INVALID_DESC;
    int K, KT, nf;
    if (eq_orders(desc->eq, nout, &K, &KT, &nf)) return PINN_ERR_INVALID_DESC;

    pinn_handle *h = new (std::nothrow) pinn_handle();
    if (!h) return PINN_ERR_ALLOC;
    h->desc = *desc;
    h->K = K; h->KT = KT; h->J = K + 1 + KT; h->nf = nf;
    KDesc &kd = h->kd;
    memset(&kd, 0, sizeof(kd));
    kd.n_lin = desc->n_lin;
    int off = 0;
    for (int l = 0; l <= desc->n_lin; ++l) kd.widths[l] = desc->widths[l];
    for (int l = 0; l < desc->n_lin; ++l) {
        kd.woff[l] = off; off += desc->widths[l] * desc->widths[l + 1];
        kd.boff[l] = off; off += desc->widths[l + 1];
    }
    kd.P = off;
    kd.nout = nout;
    kd.lbx = desc->lb[0]; kd.lbt = desc->lb[1];
    kd.ubx = desc->ub[0]; kd.ubt = desc->ub[1];
    kd.sx = 2.0f / (kd.ubx - kd.lbx);
    kd.st = 2.0f / (kd.ubt - kd.lbt);
    for (int i = 0; i < 8; ++i) kd.coef[i] = desc->coef[i];

    if (hipGetDevice(&h->device) != hipSuccess) { (void)hipGetLastError(); delete h; return PINN_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) { (void)hipGetLastError(); delete h; return PINN_ERR_HIP; }
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    // kernel family: MFMA whenever an instantiation covers the (uniform) hidden width and fits LDS
    bool uniform = desc->n_lin >= 2;
    const int W = desc->widths[1];
    for (int l = 1; l < desc->n_lin; ++l) uniform = uniform && desc->widths[l] == W;
    int kernel = desc->kernel;
#ifdef PINN_WITH_MFMA
    const bool mfma_ok = uniform && mfma_supported(kd, K + 1 + KT);
#else
    const bool mfma_ok = false;
#endif
    if (kernel == PINN_KERNEL_AUTO) kernel = mfma_ok ? PINN_KERNEL_MFMA : PINN_KERNEL_VALU;
    if (kernel == PINN_KERNEL_MFMA && !mfma_ok) { delete h; return PINN_ERR_UNSUPPORTED; }
    if (kernel != PINN_KERNEL_MFMA && kernel != PINN_KERNEL_VALU) { delete h; return PINN_ERR_INVALID_DESC; }
    if (kernel == PINN_KERNEL_VALU && (size_t)(kd.P + MAXF) * sizeof(float) > 160 * 1024) { delete h; return PINN_ERR_UNSUPPORTED; }
    h->kernel = kernel;
    h->mfma_grid = h->n_cu;      // persistent kernels: at most one workgroup per CU
#ifdef PINN_DEV_KNOBS                                        // development builds only (tools/dev_build.sh -DPINN_DEV_KNOBS)
    if (const char *e = getenv("PINN_SPARE_CUS")) {
        const int v = atoi(e);
        if (v >= 0 && v <= h->n_cu / 4) h->spare_cus = v;
    }
#endif
    // set-up work of the no-allocation entry point: this device's address of the static scratch array
    { const int rc = static_scratch_lookup(h->device); if (rc) { delete h; return rc; } }
    // the counters the MFMA kernels hand their tiles out from: zeroed once, here; every launch leaves them at zero
    if (kernel == PINN_KERNEL_MFMA) {
        if (hipMalloc((void **)&h->tile_ctr, 256) != hipSuccess) { (void)hipGetLastError(); delete h; return PINN_ERR_ALLOC; }
        if (hipMemset(h->tile_ctr, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(h->tile_ctr); delete h; return PINN_ERR_HIP;
        }
    }
    *out = h;
    return PINN_OK;
}

int pinn_destroy(pinn_handle *h)
{
    if (!h) return PINN_OK;
    if (h->jets) (void)hipFree(h->jets);
    if (h->slab) (void)hipFree(h->slab);
    if (h->tile_ctr) (void)hipFree(h->tile_ctr);
    delete h;
    return PINN_OK;
}

int pinn_param_count(const pinn_handle *h) { return h ? h->kd.P : PINN_ERR_NULL; }
int pinn_jet_count(const pinn_handle *h) { return h ? h->J : PINN_ERR_NULL; }
int pinn_nf(const pinn_handle *h) { return h ? h->nf : PINN_ERR_NULL; }
int pinn_nout(const pinn_handle *h) { return h ? h->kd.nout : PINN_ERR_NULL; }
int pinn_kernel_kind(const pinn_handle *h) { return h ? h->kernel : PINN_ERR_NULL; }

int pinn_forward(pinn_handle *h, const float *theta, const float *x, const float *t, int64_t stride, int64_t n,
                 float *out, void *stream)
{
    if (!out) return PINN_ERR_NULL;
    CallArgs a{theta, x, t, stride, n, nullptr, 0.0f, out, nullptr, nullptr};
    return run(h, PINN_EQ_NONE, MODE_FWD, a, nullptr, nullptr, stream);
}

int pinn_forward_vjp(pinn_handle *h, const float *theta, const float *x, const float *t, int64_t stride,
                     int64_t n, const float *gbar, float *grad_theta_accum, void *stream)
{
    if (!gbar || !grad_theta_accum) return PINN_ERR_NULL;
    CallArgs a{theta, x, t, stride, n, gbar, 0.0f, nullptr, nullptr, nullptr};
    return run(h, PINN_EQ_NONE, MODE_VJP, a, grad_theta_accum, nullptr, stream);
}

int pinn_forward_vjp_static(pinn_handle *h, const float *theta, const float *x, const float *t, int64_t stride,
                            int64_t n, const float *gbar, float *grad_theta_accum, void *stream)
{
    if (!h) return PINN_ERR_NULL;
    if (!gbar || !grad_theta_accum) return PINN_ERR_NULL;
    if (n > PINN_STATIC_MAX_POINTS) return PINN_ERR_TOO_LARGE;
    // the array lives on the handle's device, at the address looked up when the handle was created
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return PINN_ERR_HIP; }
    if (cur != h->device) return PINN_ERR_INVALID_DESC;
    void *ptr = g_static_ptr[h->device];
    if (!ptr) return PINN_ERR_HIP;
    const FixedWs ws{(unsigned char *)ptr, (size_t)PINN_STATIC_SCRATCH_BYTES};
    CallArgs a{theta, x, t, stride, n, gbar, 0.0f, nullptr, nullptr, nullptr};
    return run(h, PINN_EQ_NONE, MODE_VJP, a, grad_theta_accum, nullptr, stream, &ws);
}

int pinn_reserve(pinn_handle *h, int64_t n_max)
{
    if (!h) return PINN_ERR_NULL;
    if (n_max < 0) return PINN_ERR_INVALID_DESC;
    if (n_max == 0) return PINN_OK;
    size_t jets = 0, slab = 0;
    const int eqs[2] = {h->desc.eq, PINN_EQ_NONE};
    for (int e = 0; e < 2; ++e)
        for (int mode = MODE_FWD; mode <= MODE_LOSS; ++mode) {
            Need nd;
            const int rc = call_need(h, eqs[e], mode, n_max, &nd);
            if (rc) return rc;
            if (nd.jets_per_wg * nd.grid > jets) jets = nd.jets_per_wg * nd.grid;
            if (nd.slab_per_wg * nd.grid > slab) slab = nd.slab_per_wg * nd.grid;
        }
    int rc = ensure(&h->jets, &h->jets_bytes, jets);
    if (rc) return rc;
    return ensure(&h->slab, &h->slab_bytes, slab);
}

int pinn_residual_fwd(pinn_handle *h, const float *theta, const float *x, const float *t, int64_t stride,
                      int64_t n, float *f_out, float *derivs_out, void *stream)
{
    if (!h) return PINN_ERR_NULL;
    if (!f_out && !derivs_out) return PINN_ERR_NULL;
    CallArgs a{theta, x, t, stride, n, nullptr, 0.0f, f_out, derivs_out, nullptr};
    return run(h, h->desc.eq, MODE_FWD, a, nullptr, nullptr, stream);
}

int pinn_residual_from_derivs(pinn_handle *h, const float *derivs, int64_t n, float *f_out, void *stream)
{
    if (!h) return PINN_ERR_NULL;
    if (!derivs || !f_out) return PINN_ERR_NULL;
    if (n < 0) return PINN_ERR_INVALID_DESC;
    if (n == 0) return PINN_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long nb = (n + 255) / 256;
    const dim3 grid((unsigned)(nb < 4096 ? nb : 4096)), block(256);
    switch (h->desc.eq) {
    case PINN_EQ_NONE:    hipLaunchKernelGGL(k_residual_from_derivs<EQ_NONE>, grid, block, 0, st, h->kd, derivs, (long long)n, f_out); break;
    case PINN_EQ_BURGERS: hipLaunchKernelGGL(k_residual_from_derivs<EQ_BURGERS>, grid, block, 0, st, h->kd, derivs, (long long)n, f_out); break;
    case PINN_EQ_KDV5:    hipLaunchKernelGGL(k_residual_from_derivs<EQ_KDV5>, grid, block, 0, st, h->kd, derivs, (long long)n, f_out); break;
    case PINN_EQ_KDV7:    hipLaunchKernelGGL(k_residual_from_derivs<EQ_KDV7>, grid, block, 0, st, h->kd, derivs, (long long)n, f_out); break;
    case PINN_EQ_CH:      hipLaunchKernelGGL(k_residual_from_derivs<EQ_CH>, grid, block, 0, st, h->kd, derivs, (long long)n, f_out); break;
    case PINN_EQ_GNSS:    hipLaunchKernelGGL(k_residual_from_derivs<EQ_GNSS>, grid, block, 0, st, h->kd, derivs, (long long)n, f_out); break;
    case PINN_EQ_LPD:     hipLaunchKernelGGL(k_residual_from_derivs<EQ_LPD>, grid, block, 0, st, h->kd, derivs, (long long)n, f_out); break;
    case PINN_EQ_NLSREG:  hipLaunchKernelGGL(k_residual_from_derivs<EQ_NLSREG>, grid, block, 0, st, h->kd, derivs, (long long)n, f_out); break;
    default: return PINN_ERR_INVALID_DESC;
    }
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

int pinn_residual_vjp(pinn_handle *h, const float *theta, const float *x, const float *t, int64_t stride,
                      int64_t n, const float *gbar, float *grad_theta_accum, void *stream)
{
    if (!h) return PINN_ERR_NULL;
    if (!gbar || !grad_theta_accum) return PINN_ERR_NULL;
    CallArgs a{theta, x, t, stride, n, gbar, 0.0f, nullptr, nullptr, nullptr};
    return run(h, h->desc.eq, MODE_VJP, a, grad_theta_accum, nullptr, stream);
}

int pinn_loss_grad(pinn_handle *h, const float *theta, const float *x, const float *t, int64_t stride, int64_t n,
                   float scale, float *loss_sums_accum, float *grad_theta_accum, void *stream)
{
    if (!h) return PINN_ERR_NULL;
    if (!loss_sums_accum || !grad_theta_accum) return PINN_ERR_NULL;
    CallArgs a{theta, x, t, stride, n, nullptr, scale, nullptr, nullptr, nullptr};
    return run(h, h->desc.eq, MODE_LOSS, a, grad_theta_accum, loss_sums_accum, stream);
}

const char *pinn_error_string(int code)
{
    switch (code) {
    case PINN_OK: return "ok";
    case PINN_ERR_INVALID_DESC: return "invalid descriptor or argument";
    case PINN_ERR_UNSUPPORTED: return "unsupported network shape for the requested kernel";
    case PINN_ERR_HIP: return "HIP runtime error";
    case PINN_ERR_NULL: return "null pointer";
    case PINN_ERR_ALLOC: return "device allocation failed";
    case PINN_ERR_TOO_LARGE: return "problem too large for the static scratch of the no-allocation entry point";
    default: return "unknown error";
    }
}

}  // extern "C"
