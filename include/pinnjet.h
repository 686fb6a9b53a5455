/*
 * pinnjet — C-ABI of the MI355X-native PINN collocation engine.
 *
 * The reference (Emily-Qu/PDE) has no FFI/operator interface: its de-facto
 * boundary is the method surface of the per-script class PhysicsInformedNN
 * (`self.dnn(X)`, `net_f`, `backward`, `predict`; 7kdv/PI-7kdv3soliton.py:76-184)
 * and the torch.optim closure protocol.  This header is what a binding for that
 * surface attaches to; each entry point cites the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch storages);
 *    the library never frees or reallocates them;
 *  - `theta` is one contiguous f32 buffer in state-dict order
 *    [W0 (out,in) row-major, b0, W1, b1, ...]  (torch.nn.Linear layout,
 *    7kdv/PI-7kdv3soliton.py:49-56; keys 1.layers.layer_i.{weight,bias});
 *  - point coordinates are given as two base pointers plus one common element
 *    stride: separate (N,1) columns -> stride 1 (xf/tf, :88-89); an (N,2)
 *    row-major X_star -> x = X, t = X + 1, stride 2 (predict, :179-184);
 *  - `*_accum` outputs are ACCUMULATED into (+=), matching `.grad` semantics
 *    (callers zero first: optimizer.zero_grad(), :155,159);
 *  - kernels are enqueued on `stream` (a hipStream_t passed as void*).  An entry
 *    point allocates (hipMalloc / hipFree, which may synchronise the device and
 *    cannot run under stream capture) ONLY when the call needs more workspace than
 *    pinn_reserve() or any earlier call on the handle provided; size the handle
 *    with pinn_reserve() at set-up and no later call allocates or synchronises;
 *  - return value: 0 = ok, negative = error (pinn_error_string()); nothing is
 *    printed and nothing aborts.
 *  - a handle's entry points must not be called concurrently from two host
 *    threads/streams (they share the handle's device workspace: the reverse
 *    sweep's stash, the per-workgroup partial sums and the two counters from
 *    which a launch hands its tiles out, which every launch leaves at zero);
 *    distinct handles are independent;
 *  - which workgroup sums which points is decided while a launch runs (tiles
 *    are handed out first come, first served): two launches on the same inputs
 *    agree to float32 summation order, not bit for bit.
 */
#ifndef PINNJET_H
#define PINNJET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PINN_MAX_LAYERS 16
#define PINN_MAX_WIDTH 1024   /* any layer; the generic kernels also need 4*(P+3) <= 160 KB of LDS */
#define PINN_MAX_OUT 2

/* equation ids: which `net_f` body of the reference is evaluated */
enum {
    PINN_EQ_NONE = 0,    /* plain net_u (self.dnn), no derivatives                               */
    PINN_EQ_BURGERS = 1, /* Burgers/Burgers_equation(PINN).py:126-131   coef[0]=nu               */
    PINN_EQ_KDV5 = 2,    /* 5kdv/PI-51.py:132-140                       coef[0..3]=180,30,30,1   */
    PINN_EQ_KDV7 = 3,    /* 7kdv/PI-7kdv3soliton.py:122-132 / PI-7kdv1soliton.py:132  coef[0..7] */
    PINN_EQ_CH = 4,      /* CH/PI-CH-1soliton.py:131-138 (coef[0]=0), CH/bc-CH-1soliton.py:270-278 (coef[0]=2) */
    PINN_EQ_GNSS = 5,    /* NLS/sw-GNSS-adamW.py:159-180   coef[0..2]=a,b,c (c=0: plain NLS)     */
    PINN_EQ_LPD = 6,     /* NLS/Ibc-test.py:163-164, NLS/testGNSS.py:139-140 (4th order)  coef[0..5]=0.5,1,0.125,0.0625,6,10 */
    PINN_EQ_NLSREG = 7   /* NLS/sw-test.py:159-189: plain NLS + derivative regulariser; coef[0..2]=a,b,lambda(1e-9).
                            THREE residual rows (pinn_nf = 3): fu, fv and f2 = sqrt(lambda * sum of the six squared
                            derivatives :173-180), so sum_c mean(f_c^2) = loss_f + reg_term                        */
};

/* kernel selection.  The FP32-MFMA kernels are instantiated for nets whose hidden layers all have
 * the SAME width, one of 20, 50, 64, 100 (every width the reference's scripts use), with at least
 * one hidden layer, as long as the WEIGHTS fit LDS (the 6 x 64 and 4 x 100 nets of the reference do; five
 * hidden layers of width 100 do not); every other shape runs the generic VALU kernels, which are ~20x
 * slower and need the whole gradient (P + 3 floats) in LDS.  A net neither family covers makes pinn_create
 * return PINN_ERR_UNSUPPORTED. */
enum {
    PINN_KERNEL_AUTO = 0,    /* the MFMA kernels when an instantiation covers the shape, else VALU     */
    PINN_KERNEL_VALU = 1,    /* generic one-point-per-lane VALU kernels, any widths                    */
    PINN_KERNEL_MFMA = 2     /* 16-point-tile FP32-MFMA kernels; PINN_ERR_UNSUPPORTED if not covered   */
};

typedef struct pinn_desc {
    int32_t eq;                           /* PINN_EQ_*                                        */
    int32_t n_lin;                        /* number of Linear layers = len(layers) - 1        */
    int32_t widths[PINN_MAX_LAYERS + 1];  /* the script's `layers` list, widths[0] must be 2  */
    float lb[2], ub[2];                   /* Scale bounds (x, t), float32 like the reference  */
    float coef[8];                        /* residual coefficients, see the enum above        */
    int32_t kernel;                       /* PINN_KERNEL_*                                    */
    int32_t reserved[3];
} pinn_desc;

typedef struct pinn_handle pinn_handle;

/* Create/destroy an engine for one (network shape, equation) on the CURRENT
 * HIP device.  Owns only workspace.  Replaces the construction of
 * Sequential(Scale, DNN) + the choice of net_f body (:77-81, :122-132). */
int pinn_create(const pinn_desc *desc, pinn_handle **out);
int pinn_destroy(pinn_handle *h);

/* Size the handle's workspace for calls of up to n_max points (any entry point, any of them on the
 * handle's equation or on plain net_u), so that none of them allocates afterwards (see Conventions).
 * No counterpart in the reference (torch's caching allocator plays this role there). */
int pinn_reserve(pinn_handle *h, int64_t n_max);

/* shape queries: P = parameter count, J = jet components per output carried by
 * the residual kernels (K+1 x-orders then KT t-tangents), nf = residual
 * components (2 for GNSS: (f_u, f_v)), nout = network outputs. */
int pinn_param_count(const pinn_handle *h);
int pinn_jet_count(const pinn_handle *h);
int pinn_nf(const pinn_handle *h);
int pinn_nout(const pinn_handle *h);
/* which kernel family the handle resolved to (PINN_KERNEL_VALU / _MFMA) */
int pinn_kernel_kind(const pinn_handle *h);

/* net_u: out[n][nout] = dnn([x,t]).  Replaces `self.dnn(torch.concat([x,t],1))`
 * (:99,102-103) and `predict` (:179-184). */
int pinn_forward(pinn_handle *h, const float *theta, const float *x, const float *t,
                 int64_t stride, int64_t n, float *out, void *stream);

/* theta-gradient of sum_p sum_o gbar[p][o] * dnn(x_p,t_p)[o]; the backward of
 * the data-loss terms loss0 / loss_b / loss1 / loss_prev (:99-105,
 * 7kdv/sw-73.py:132-142, 7kdv/bc-7kdv3soliton.py:251-253). */
int pinn_forward_vjp(pinn_handle *h, const float *theta, const float *x, const float *t,
                     int64_t stride, int64_t n, const float *gbar, float *grad_theta_accum,
                     void *stream);

/* pinn_forward_vjp for callers that cannot allocate — code under HIP-graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference.  Same arguments
 * and result as pinn_forward_vjp (the data-loss terms are its use: they have at most a few thousand
 * points), but the kernels' scratch (per-wave stash / per-thread jets and the per-workgroup partial
 * slabs) lives in ONE static device array of PINN_STATIC_SCRATCH_BYTES that every GPU the library
 * runs on holds until the process ends; the call never allocates, frees or synchronises.  The grid is
 * sized to what fits the array.  n > PINN_STATIC_MAX_POINTS, or a net of which not even one
 * workgroup's scratch fits, is rejected with PINN_ERR_TOO_LARGE.  The handle's device must be current.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams,
 * two host threads, two handles) race; run them one after the other on one stream. */
#define PINN_STATIC_SCRATCH_BYTES (8 * 1024 * 1024)
#define PINN_STATIC_MAX_POINTS 65536
int pinn_forward_vjp_static(pinn_handle *h, const float *theta, const float *x, const float *t,
                            int64_t stride, int64_t n, const float *gbar, float *grad_theta_accum,
                            void *stream);

/* net_f: f_out[n][nf]; optional derivs_out[n][nout][J] = true derivatives
 * (d^k u/dx^k, k=0..K, then d/dt of orders 0..KT-1).  Replaces net_f's nested
 * torch.autograd.grad chain (:122-132).  Either output may be NULL. */
int pinn_residual_fwd(pinn_handle *h, const float *theta, const float *x, const float *t,
                      int64_t stride, int64_t n, float *f_out, float *derivs_out, void *stream);

/* The residual polynomial alone — the `return u_t + ...` line of net_f (7kdv/PI-7kdv3soliton.py:132,
 * 5kdv/PI-51.py:140, Burgers/Burgers_equation(PINN).py:131, CH/PI-CH-1soliton.py:138,
 * NLS/sw-GNSS-adamW.py:178-179) — on caller-supplied true derivatives derivs[n][nout][J] (layout of
 * derivs_out): f_out[n][nf].  The same device code (Residual<EQ>) the fused kernels inline; it lets a
 * caller check the formulas and coefficient sets on exact solutions, without any network. */
int pinn_residual_from_derivs(pinn_handle *h, const float *derivs, int64_t n, float *f_out, void *stream);

/* theta-gradient of sum_p sum_c gbar[p][c] * f_c(x_p,t_p): what autograd's
 * backward through net_f's graph produces for an arbitrary loss on f. */
int pinn_residual_vjp(pinn_handle *h, const float *theta, const float *x, const float *t,
                      int64_t stride, int64_t n, const float *gbar, float *grad_theta_accum,
                      void *stream);

/* fused fast path of `backward()` for the collocation term (:107-108,111):
 *   loss_sums_accum[c] += sum_p f_c^2            (unscaled, c < nf)
 *   grad_theta_accum   += scale * d/dtheta sum_c sum_p f_c^2
 * with scale = 1/N_global so that the result is the gradient of
 * mean(square(f)) summed over components. */
int pinn_loss_grad(pinn_handle *h, const float *theta, const float *x, const float *t,
                   int64_t stride, int64_t n, float scale, float *loss_sums_accum,
                   float *grad_theta_accum, void *stream);

/* ---- L-BFGS on the device --------------------------------------------------
 * The reference ends every run with torch.optim.LBFGS.step(closure), strong-Wolfe
 * (7kdv/PI-7kdv3soliton.py:138-161; third-party torch/optim/lbfgs.py).  These two entry points
 * replace that optimizer's per-iteration tensor work with one launch each; the line search's
 * scalar logic stays on the host (pde_amd/lbfgs.py).  No handle: plain device vectors of n floats.
 *
 * workspace: pinn_lbfgs_workspace_floats(n, history) floats, caller-allocated, kept between calls
 *            (history rows S/Y, their Gram matrices in float64, previous gradient, H_diag); 8-byte
 *            aligned; contents are (re)initialised by a call with first != 0.  history <= 256,
 *            n <= 2^24.
 * pinn_lbfgs_direction: with first != 0: d = -g.  Otherwise: y = g - g_prev, s = t*d (d and t of
 *            the previous call/step), pair stored if y.s > 1e-10 (oldest dropped when full),
 *            H = y.s/y.y, d = two-loop recursion applied to -g.  Always: g_prev = g and
 *            stats8 = [g.d, sum|g|, max|d|, max|g|, pairs held, H_diag, pair accepted (0/1), 0].
 *            Three launches (row dot products over the whole chip, the scalar recursion in one
 *            workgroup, the combination, whose last workgroup folds the statistics); deterministic
 *            (fixed-order reductions; the one atomic is a ticket).
 * pinn_lbfgs_direction_advance: the same, and when x != NULL the combination launch also moves the
 *            weights to the line search's first trial point: x0 = x, x = x + t_next * d (x, x0: n floats;
 *            torch.optim.LBFGS does this as a clone and an add per evaluation, torch/optim/lbfgs.py
 *            _directional_evaluate).
 * pinn_lbfgs_eval_stats: out4 = [*loss (0 when NULL), g.d, max|g|, sum|g|] after one closure
 *            evaluation of the line search.
 * pinn_lbfgs_eval_stats_keep: the same, and g_copy (n floats, may be NULL) = g: the gradient the line
 *            search keeps for the point (torch: flat_grad.clone()).  host13 (may be NULL): 13 words of
 *            pinned host memory, as a device pointer (pinn_host_device_pointer), that receive
 *            [stats8 (8 floats, skipped when stats8 is NULL) | out4 | ticket (int32)], the ticket last and
 *            after a system-scope fence: the host polls that word instead of waiting on the stream
 *            (torch.optim.LBFGS blocks on `float(loss)` and on every scalar it compares).
 * pinn_lbfgs_poll_advance: HOST work of one line-search evaluation in the usual case, without an interpreter between
 *            the arrival of the statistics and the next launch: spins (at most max_spins times; returns 2 when they
 *            run out) until host13[12] == ticket, copies the 12 floats to vals12 and, when allow != 0, the first
 *            trial point t = lr satisfies both Wolfe conditions against f_prev and stats8's g.d (c1 = 1e-4, c2 = 0.9)
 *            and none of torch.optim.LBFGS.step's stopping rules fires (direction not descending, max|g| <= tol_g,
 *            max|d| lr <= tol_x, |f - f_prev| < tol_x), calls pinn_lbfgs_direction_advance(g_keep, t = lr, x, x0, lr)
 *            and returns 1; returns 0 when the caller has to decide (nothing launched).  host13 is the HOST address
 *            here.  The same float64 comparisons as pde_amd/lbfgs.py makes, one rounding per operation.
 * pinn_host_device_pointer: the device address of pinned (hipHostMalloc / torch pin_memory) host memory;
 *            PINN_ERR_HIP when the memory is not mapped into the device's address space. */
int64_t pinn_lbfgs_workspace_floats(int64_t n, int32_t history);
int pinn_lbfgs_direction(float *workspace, int64_t n, int32_t history, const float *g, float *d, float t,
                         int32_t first, float *stats8, void *stream);
int pinn_lbfgs_direction_advance(float *workspace, int64_t n, int32_t history, const float *g, float *d, float t,
                                 int32_t first, float *stats8, float *x, float *x0, float t_next, void *stream);
int pinn_lbfgs_eval_stats(const float *g, const float *d, int64_t n, const float *loss, float *out4,
                          void *stream);
int pinn_lbfgs_eval_stats_keep(const float *g, const float *d, int64_t n, const float *loss, float *out4,
                               float *g_copy, const float *stats8, float *host13, int32_t ticket, void *stream);
int pinn_lbfgs_poll_advance(const float *host13, int32_t ticket, int64_t max_spins, double f_prev, double lr,
                            double tol_x, double tol_g, int32_t allow, float *workspace, int64_t n, int32_t history,
                            const float *g_keep, float *d, float *stats8, float *x, float *x0, void *stream,
                            float *vals12);
int pinn_host_device_pointer(void *host, void **device);

/* ---- data terms of backward() on the outputs of one net_u batch ------------------------------------------
 * loss0 / loss_b of the reference's backward() (7kdv/PI-7kdv3soliton.py:99-105; two outputs:
 * NLS/sw-GNSS-adamW.py:114-129; Dirichlet form: Burgers/Burgers_equation(PINN).py:109), which it forms as a chain
 * of ~12 small torch kernels each way.  `out` = what pinn_forward wrote for the batch
 * X = [IC rows (n0) | lower-boundary rows (nb) | upper-boundary rows (nb)], [rows][nout]; tgt0: [n0][nout].
 *   losses[0] = sum (out - tgt0)^2 / n0 over the IC rows and all columns (nb = 0: any plain mean-square term,
 *               e.g. loss1 of the bc classes, loss_prev of the sliding-window classes)
 *   losses[1] = sum (out[lower] - out[upper])^2 / nb (dirichlet == 0) or sum out^2 / nb over both boundary blocks
 * pinn_data_loss_cotangent writes gbar[rows][nout] = g0 * d losses[0]/d out + gb * d losses[1]/d out (g0, gb: device
 * scalars, the upstream gradients of the two terms), the argument of pinn_forward_vjp.  One launch each, float64
 * sums in fixed order; nothing is allocated. */
int pinn_data_loss(const float *out, const float *tgt0, int32_t n0, int32_t nb, int32_t nout, int32_t dirichlet,
                   float *losses, void *stream);
int pinn_data_loss_cotangent(const float *out, const float *tgt0, int32_t n0, int32_t nb, int32_t nout, int32_t dirichlet,
                             const float *g0, const float *gb, float *gbar, void *stream);

/* ---- the tail of backward(): `loss = loss0 + loss_b [+ loss1] [+ loss_prev] [+ loss_s] + loss_f; loss.backward()`
 * (7kdv/PI-7kdv3soliton.py:110-111; bc: 7kdv/bc-7kdv3soliton.py:252-253; sw: 7kdv/sw-73.py:142-143) once every
 * term's value and flat weight gradient exist, in ONE launch instead of a divide and four adds:
 *   loss2[1] = loss_f = (sums[0] + ... + sums[nf-1]) / n_total        (sums: what pinn_loss_grad accumulated)
 *   loss2[0] = loss   = (((terms[0] + terms[1]) + loss_f) + terms[2]) + ...   (float32, in the reference's order:
 *                       terms = loss0, loss_b, then loss1 / loss_prev / loss_s as present)
 *   grad[i]  = ga[i] + gb[i]   (flags & 1: grad[i] += ga[i] + gb[i]; gb may be NULL), i < P
 * flags & 2: gb and sums are set to zero once they have been read - a captured closure keeps ONE collocation buffer
 * for pinn_loss_grad to accumulate into and never launches a fill for it.  n_terms <= 16, nf <= 3. */
int pinn_closure_sum(const float *terms, int32_t n_terms, float *sums, int32_t nf, float n_total, float *loss2,
                     const float *ga, float *gb, float *grad, int64_t P, int32_t flags, void *stream);

/* ---- Adam / AdamW on the device, one launch per step ------------------------------------
 * Replaces torch.optim.Adam / AdamW(weight_decay=1e-2).step() of the reference's Adam phase
 * (7kdv/PI-7kdv3soliton.py:138,154-157; 7kdv/sw73-adamW.py:170) for parameters held as ONE flat vector:
 * the arithmetic of torch's default (non-amsgrad) Adam, decoupled != 0 selects AdamW's decay.  `state` is
 * 32 bytes of device memory (zero it to start: step count and running beta powers in float64 + a ticket
 * word); the step count advances on the device, so the call has no per-step host argument and replays from a
 * captured graph.  exp_avg / exp_avg_sq: n floats each, zero-initialised by the caller. */
int pinn_adam_step(float *theta, const float *grad, float *exp_avg, float *exp_avg_sq, double *state,
                   int64_t n, float lr, double beta1, double beta2, float eps, float weight_decay,
                   int32_t decoupled, void *stream);

const char *pinn_error_string(int code);

#define PINN_OK 0
#define PINN_ERR_INVALID_DESC (-1)
#define PINN_ERR_UNSUPPORTED (-2)
#define PINN_ERR_HIP (-3)
#define PINN_ERR_NULL (-4)
#define PINN_ERR_ALLOC (-5)
#define PINN_ERR_TOO_LARGE (-6)

#ifdef __cplusplus
}
#endif
#endif /* PINNJET_H */
