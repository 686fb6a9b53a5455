// C-ABI of the PINN collocation engine (include/pinnjet.h) and kernel dispatch.
#include <hip/hip_runtime.h>
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
    // workspace (grown on demand, never shrunk)
    float *scratch = nullptr;
    size_t scratch_bytes = 0;
    float *slab = nullptr;
    size_t slab_bytes = 0;
#ifdef PINN_WITH_MFMA
    MfmaPlan mfma;
#endif
};

namespace {

int eq_orders(int eq, int nout, int *K, int *KT, int *nf)
{
    switch (eq) {
    case PINN_EQ_NONE:    *K = 0; *KT = 0; *nf = nout; return 0;
    case PINN_EQ_BURGERS: *K = 2; *KT = 1; *nf = 1; return nout == 1 ? 0 : -1;
    case PINN_EQ_KDV5:    *K = 5; *KT = 1; *nf = 1; return nout == 1 ? 0 : -1;
    case PINN_EQ_KDV7:    *K = 7; *KT = 1; *nf = 1; return nout == 1 ? 0 : -1;
    case PINN_EQ_CH:      *K = 3; *KT = 3; *nf = 1; return nout == 1 ? 0 : -1;
    case PINN_EQ_GNSS:    *K = 3; *KT = 1; *nf = 2; return nout == 2 ? 0 : -1;
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

__global__ __launch_bounds__(256) void k_reduce_slabs(const float *slab, int nblk, int P, float *grad_accum,
                                                      float *loss_accum, int nf)
{
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (i < P + 2) {
        const int per = (nblk + 3) / 4;
        const int b0 = wv * per, b1 = min(nblk, b0 + per);
        for (int b = b0; b < b1; ++b) s += (double)slab[(size_t)b * (P + 2) + i];
    }
    part[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && i < P + 2) {
        const double tot = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        if (i < P) {
            if (grad_accum) grad_accum[i] += (float)tot;
        } else if (loss_accum && (i - P) < nf) {
            loss_accum[i - P] += (float)tot;
        }
    }
}

template <int EQ, int MODE>
int launch_valu_t(pinn_handle *h, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    const size_t lds = (MODE == MODE_FWD) ? 0 : (size_t)(kd.P + 2) * sizeof(float);
    auto kern = k_valu<EQ, MODE>;
    if (lds > 48 * 1024) {
        if (lds > 160 * 1024) return PINN_ERR_UNSUPPORTED;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return PINN_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(VALU_BLOCK), lds, st, kd, a, h->scratch);
    return hipGetLastError() == hipSuccess ? PINN_OK : PINN_ERR_HIP;
}

template <int MODE>
int launch_valu_eq(pinn_handle *h, int eq, const KDesc &kd, const CallArgs &a, int grid, hipStream_t st)
{
    switch (eq) {
    case PINN_EQ_NONE:    return launch_valu_t<EQ_NONE, MODE>(h, kd, a, grid, st);
    case PINN_EQ_BURGERS: return launch_valu_t<EQ_BURGERS, MODE>(h, kd, a, grid, st);
    case PINN_EQ_KDV5:    return launch_valu_t<EQ_KDV5, MODE>(h, kd, a, grid, st);
    case PINN_EQ_KDV7:    return launch_valu_t<EQ_KDV7, MODE>(h, kd, a, grid, st);
    case PINN_EQ_CH:      return launch_valu_t<EQ_CH, MODE>(h, kd, a, grid, st);
    case PINN_EQ_GNSS:    return launch_valu_t<EQ_GNSS, MODE>(h, kd, a, grid, st);
    default: return PINN_ERR_INVALID_DESC;
    }
}

// one residual-type call on the VALU path
int run_valu(pinn_handle *h, int eq, int mode, CallArgs a, float *grad_accum, float *loss_accum, hipStream_t st)
{
    int K, KT, nf;
    if (eq_orders(eq, h->kd.nout, &K, &KT, &nf)) return PINN_ERR_INVALID_DESC;
    const int J = K + 1 + KT;
    const long long need_blocks = (a.n + VALU_BLOCK - 1) / VALU_BLOCK;
    const int grid = (int)(need_blocks < h->n_cu ? need_blocks : h->n_cu);
    const ValuSlots sl = valu_slots(h->kd, J);
    int rc = ensure(&h->scratch, &h->scratch_bytes, (size_t)sl.total * grid * VALU_BLOCK * sizeof(float));
    if (rc) return rc;
    if (mode != MODE_FWD) {
        rc = ensure(&h->slab, &h->slab_bytes, (size_t)grid * (h->kd.P + 2) * sizeof(float));
        if (rc) return rc;
        a.slab = h->slab;
    }
    switch (mode) {
    case MODE_FWD:  rc = launch_valu_eq<MODE_FWD>(h, eq, h->kd, a, grid, st); break;
    case MODE_VJP:  rc = launch_valu_eq<MODE_VJP>(h, eq, h->kd, a, grid, st); break;
    default:        rc = launch_valu_eq<MODE_LOSS>(h, eq, h->kd, a, grid, st); break;
    }
    if (rc) return rc;
    if (mode != MODE_FWD) {
        const int rb = (h->kd.P + 2 + 63) / 64;
        hipLaunchKernelGGL(k_reduce_slabs, dim3(rb), dim3(256), 0, st, h->slab, grid, h->kd.P, grad_accum,
                           loss_accum, nf);
        if (hipGetLastError() != hipSuccess) return PINN_ERR_HIP;
    }
    return PINN_OK;
}

#ifdef PINN_WITH_MFMA
// one residual-type call on the MFMA path
int run_mfma(MfmaPlan &pl, const KDesc &kd, int eq, int mode, CallArgs a, float *grad_accum, float *loss_accum,
             hipStream_t st)
{
    int K, KT, nf;
    if (eq_orders(eq, kd.nout, &K, &KT, &nf)) return PINN_ERR_INVALID_DESC;
    const int J = K + 1 + KT;
    const long long ntiles = (a.n + 15) / 16;
    const int waves = mfma_waves(kd, J);
    const long long need_blocks = (ntiles + waves - 1) / waves;
    const int grid = (int)(need_blocks < pl.grid ? need_blocks : pl.grid);
    int rc;
    if (mode != MODE_FWD) {
        const size_t per_wave = mfma_stash_floats(kd, J) * sizeof(float);
        rc = ensure(&pl.stash, &pl.stash_bytes, per_wave * grid * waves);
        if (rc) return rc;
        const size_t slab_bytes = (size_t)grid * (kd.P + 2) * sizeof(float);
        rc = ensure(&pl.slab, &pl.slab_bytes, slab_bytes);
        if (rc) return rc;
        a.slab = pl.slab;
        // wide layers accumulate into their slab with atomics: it has to start at zero
        if (mfma_wglobal(kd)) {
            const long long nfl = (long long)(slab_bytes / sizeof(float));
            const int zb = (int)((nfl + 256 * 8 - 1) / (256 * 8) < 2048 ? (nfl + 256 * 8 - 1) / (256 * 8) : 2048);
            hipLaunchKernelGGL(k_zero_floats, dim3(zb > 0 ? zb : 1), dim3(256), 0, st, pl.slab, nfl);
        }
    }
    switch (mode) {
    case MODE_FWD:  rc = launch_mfma_eq<MODE_FWD>(pl, eq, kd, a, grid, st); break;
    case MODE_VJP:  rc = launch_mfma_eq<MODE_VJP>(pl, eq, kd, a, grid, st); break;
    default:        rc = launch_mfma_eq<MODE_LOSS>(pl, eq, kd, a, grid, st); break;
    }
    if (rc) return rc;
    if (mode != MODE_FWD) {
        const int rb = (kd.P + 2 + 63) / 64;
        hipLaunchKernelGGL(k_reduce_slabs, dim3(rb), dim3(256), 0, st, pl.slab, grid, kd.P, grad_accum, loss_accum, nf);
        if (hipGetLastError() != hipSuccess) return PINN_ERR_HIP;
    }
    return PINN_OK;
}
#endif

int run(pinn_handle *h, int eq, int mode, const CallArgs &a, float *grad_accum, float *loss_accum, void *stream)
{
    if (!h) return PINN_ERR_NULL;
    if (!a.theta || !a.x || !a.t) return PINN_ERR_NULL;
    if (a.n < 0 || a.stride < 1) return PINN_ERR_INVALID_DESC;
    if (a.n == 0) return PINN_OK;
    hipStream_t st = (hipStream_t)stream;
#ifdef PINN_WITH_MFMA
    if (h->kernel == PINN_KERNEL_MFMA) return run_mfma(h->mfma, h->kd, eq, mode, a, grad_accum, loss_accum, st);
#endif
    return run_valu(h, eq, mode, a, grad_accum, loss_accum, st);
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
        if (desc->widths[l] < 1 || desc->widths[l] > 1024) return PINN_ERR_INVALID_DESC;
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
    if (kernel == PINN_KERNEL_VALU && (size_t)(kd.P + 2) * sizeof(float) > 160 * 1024) { delete h; return PINN_ERR_UNSUPPORTED; }
    h->kernel = kernel;
#ifdef PINN_WITH_MFMA
    if (kernel == PINN_KERNEL_MFMA) {
        const int rc = mfma_plan_create(h->mfma, kd, h->n_cu);
        if (rc) { delete h; return rc; }
    }
#endif
    *out = h;
    return PINN_OK;
}

int pinn_destroy(pinn_handle *h)
{
    if (!h) return PINN_OK;
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->slab) (void)hipFree(h->slab);
#ifdef PINN_WITH_MFMA
    if (h->kernel == PINN_KERNEL_MFMA) mfma_plan_destroy(h->mfma);
#endif
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

int pinn_residual_fwd(pinn_handle *h, const float *theta, const float *x, const float *t, int64_t stride,
                      int64_t n, float *f_out, float *derivs_out, void *stream)
{
    if (!h) return PINN_ERR_NULL;
    if (!f_out && !derivs_out) return PINN_ERR_NULL;
    CallArgs a{theta, x, t, stride, n, nullptr, 0.0f, f_out, derivs_out, nullptr};
    return run(h, h->desc.eq, MODE_FWD, a, nullptr, nullptr, stream);
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
    default: return "unknown error";
    }
}

}  // extern "C"
