/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the reference's PINN collocation hot path:
 * Scale -> DNN -> x-derivatives up to order K and t-derivative -> PDE residual
 * -> sum of squares -> gradient to the weights.  Only tests/, __graft_entry__
 * .smoke() and bench.py's cpu_baseline leg may load this.  The product
 * (pde_amd/) never includes, links or calls anything in oracle/.
 *
 * Method: the reference obtains u_x..u_xxxxxxx by nested torch.autograd.grad
 * (7kdv/PI-7kdv3soliton.py:122-132).  The same quantities are evaluated here as
 * per-point truncated Taylor series ("jets") in x, carried through every
 * Linear+Tanh layer, plus a first-order tangent in t.  The gradient to the
 * weights is a hand-written reverse sweep through the same recurrences.
 * Pinned against golden vectors generated from the imported reference itself
 * (oracle/gen_golden.py -> tests/golden/ *.npz): fp64 agreement <= 1e-10.
 *
 * This file is a "template": include it with
 *   #define REAL float|double   and   #define SFX(name) name##_f32|_f64
 */

/* ---- one tanh neuron: jet forward ---------------------------------------
 * z[0..K]   Taylor coefficients (normalised: c_k = d^k/dx^k / k!) of the
 *           pre-activation, zt[0..KT-1] Taylor coefficients of d/dt of it.
 * y = tanh(z) as a series; w = 1 - y*y as a series; yt = w * zt.
 * Recurrence: y' = w z'  =>  k y_k = sum_{j=1..k} j z_j w_{k-j}.
 * Reference semantics: torch.nn.Tanh in DNN (7kdv/PI-7kdv3soliton.py:47-53)
 * differentiated K times by autograd (:124-130). */
static void SFX(tanh_jet_fwd)(int K, int KT, const REAL *z, const REAL *zt,
                              REAL *y, REAL *w, REAL *yt)
{
    y[0] = TANH(z[0]);
    w[0] = (REAL)1 - y[0] * y[0];
    for (int k = 1; k <= K; ++k) {
        REAL s = 0;
        for (int j = 1; j <= k; ++j) s += (REAL)j * z[j] * w[k - j];
        y[k] = s / (REAL)k;
        REAL q = 0;
        for (int i = 0; i <= k; ++i) q += y[i] * y[k - i];
        w[k] = -q;
    }
    for (int k = 0; k < KT; ++k) {
        REAL s = 0;
        for (int j = 0; j <= k; ++j) s += w[j] * zt[k - j];
        yt[k] = s;
    }
}

/* reverse of the above: given adjoints yb[0..K], ytb[0..KT-1] produce
 * zb[0..K], ztb[0..KT-1].  wb is scratch of size K+1. */
static void SFX(tanh_jet_bwd)(int K, int KT, const REAL *z, const REAL *zt,
                              const REAL *y, const REAL *w,
                              REAL *yb /* modified */, const REAL *ytb,
                              REAL *zb, REAL *ztb)
{
    REAL wb[ORC_MAXJ];
    for (int k = 0; k <= K; ++k) { wb[k] = 0; zb[k] = 0; }
    for (int k = 0; k < KT; ++k) ztb[k] = 0;
    for (int k = 0; k < KT; ++k)
        for (int j = 0; j <= k; ++j) {
            wb[j] += ytb[k] * zt[k - j];
            ztb[k - j] += ytb[k] * w[j];
        }
    for (int k = K; k >= 1; --k) {
        /* w_k = -sum_i y_i y_{k-i} */
        for (int m = 0; m <= k; ++m) yb[m] += (REAL)(-2) * wb[k] * y[k - m];
        /* y_k = (1/k) sum_j j z_j w_{k-j} */
        REAL g = yb[k] / (REAL)k;
        for (int j = 1; j <= k; ++j) {
            zb[j] += g * (REAL)j * w[k - j];
            wb[k - j] += g * (REAL)j * z[j];
        }
    }
    yb[0] += (REAL)(-2) * wb[0] * y[0];
    zb[0] = yb[0] * w[0];
}

/* ---- residuals -----------------------------------------------------------
 * D[o][k] = d^k u_o/dx^k (true derivatives, k = 0..K), T[o][k] = d/dt of it
 * (k = 0..KT-1).  Writes f[0..nf-1]; if Db/Tb != NULL also accumulates
 * fb[c] * df_c/d(.) into them. */
static void SFX(residual)(const orc_desc *d, int K, int KT,
                          REAL D[ORC_MAXOUT][ORC_MAXJ], REAL T[ORC_MAXOUT][ORC_MAXJ],
                          REAL *f, const REAL *fb,
                          REAL Db[ORC_MAXOUT][ORC_MAXJ], REAL Tb[ORC_MAXOUT][ORC_MAXJ])
{
    const REAL *u = D[0];
    REAL c[8];
    for (int i = 0; i < 8; ++i) c[i] = (REAL)d->coef[i];
    (void)K; (void)KT;
    switch (d->eq) {
    case ORC_EQ_NONE: {
        int nout = d->widths[d->n_lin];
        for (int o = 0; o < nout; ++o) {
            f[o] = D[o][0];
            if (fb) Db[o][0] += fb[o];
        }
    } break;
    case ORC_EQ_BURGERS: /* Burgers/Burgers_equation(PINN).py:126-131: u_t + u u_x - nu u_xx */
        f[0] = T[0][0] + u[0] * u[1] - c[0] * u[2];
        if (fb) {
            Tb[0][0] += fb[0];
            Db[0][0] += fb[0] * u[1];
            Db[0][1] += fb[0] * u[0];
            Db[0][2] += fb[0] * (-c[0]);
        }
        break;
    case ORC_EQ_KDV5: /* 5kdv/PI-51.py:132-140: u_t + 180 u^2 u_x + 30 u_x u_xx + 30 u u_xxx + u_5x */
        f[0] = T[0][0] + c[0] * u[0] * u[0] * u[1] + c[1] * u[1] * u[2] + c[2] * u[0] * u[3] + c[3] * u[5];
        if (fb) {
            Tb[0][0] += fb[0];
            Db[0][0] += fb[0] * ((REAL)2 * c[0] * u[0] * u[1] + c[2] * u[3]);
            Db[0][1] += fb[0] * (c[0] * u[0] * u[0] + c[1] * u[2]);
            Db[0][2] += fb[0] * (c[1] * u[1]);
            Db[0][3] += fb[0] * (c[2] * u[0]);
            Db[0][5] += fb[0] * c[3];
        }
        break;
    case ORC_EQ_KDV7: /* 7kdv/PI-7kdv3soliton.py:132 and 7kdv/PI-7kdv1soliton.py:132 (coef sets) */
        f[0] = T[0][0] + c[0] * u[0] * u[0] * u[0] * u[1] + c[1] * u[1] * u[1] * u[1]
             + c[2] * u[0] * u[1] * u[2] + c[3] * u[0] * u[0] * u[3] + c[4] * u[2] * u[3]
             + c[5] * u[1] * u[4] + c[6] * u[0] * u[5] + c[7] * u[7];
        if (fb) {
            Tb[0][0] += fb[0];
            Db[0][0] += fb[0] * ((REAL)3 * c[0] * u[0] * u[0] * u[1] + c[2] * u[1] * u[2]
                                 + (REAL)2 * c[3] * u[0] * u[3] + c[6] * u[5]);
            Db[0][1] += fb[0] * (c[0] * u[0] * u[0] * u[0] + (REAL)3 * c[1] * u[1] * u[1]
                                 + c[2] * u[0] * u[2] + c[5] * u[4]);
            Db[0][2] += fb[0] * (c[2] * u[0] * u[1] + c[4] * u[3]);
            Db[0][3] += fb[0] * (c[3] * u[0] * u[0] + c[4] * u[2]);
            Db[0][4] += fb[0] * (c[5] * u[1]);
            Db[0][5] += fb[0] * (c[6] * u[0]);
            Db[0][7] += fb[0] * c[7];
        }
        break;
    case ORC_EQ_CH: /* CH/PI-CH-1soliton.py:131-138 (kappa=0), CH/bc-CH-1soliton.py:270-278 (kappa=2):
                       u_t - u_xxt + 3 u u_x - 2 u_xx u_x + kappa u_x - u_xxx u */
        f[0] = T[0][0] - T[0][2] + (REAL)3 * u[0] * u[1] - (REAL)2 * u[2] * u[1] + c[0] * u[1] - u[3] * u[0];
        if (fb) {
            Tb[0][0] += fb[0];
            Tb[0][2] += -fb[0];
            Db[0][0] += fb[0] * ((REAL)3 * u[1] - u[3]);
            Db[0][1] += fb[0] * ((REAL)3 * u[0] - (REAL)2 * u[2] + c[0]);
            Db[0][2] += fb[0] * ((REAL)(-2) * u[1]);
            Db[0][3] += fb[0] * (-u[0]);
        }
        break;
    case ORC_EQ_GNSS: { /* NLS/sw-GNSS-adamW.py:159-180; plain NLS (NLS/sw-PINN-rogue.py:179-180) is c2=0:
                           fu = u_t + a v_xx + b rho v - c (u_xxx + 6 rho u_x)
                           fv = v_t - a u_xx - b rho u - c (v_xxx + 6 rho v_x) */
        const REAL *v = D[1];
        REAL a = c[0], b = c[1], cc = c[2];
        REAL rho = u[0] * u[0] + v[0] * v[0];
        f[0] = T[0][0] + a * v[2] + b * rho * v[0] - cc * (u[3] + (REAL)6 * rho * u[1]);
        f[1] = T[1][0] - a * u[2] - b * rho * u[0] - cc * (v[3] + (REAL)6 * rho * v[1]);
        if (fb) {
            REAL g0 = fb[0], g1 = fb[1];
            Tb[0][0] += g0;
            Tb[1][0] += g1;
            Db[0][0] += g0 * (b * (REAL)2 * u[0] * v[0] - cc * (REAL)12 * u[0] * u[1])
                      + g1 * (-b * (rho + (REAL)2 * u[0] * u[0]) - cc * (REAL)12 * u[0] * v[1]);
            Db[1][0] += g0 * (b * (rho + (REAL)2 * v[0] * v[0]) - cc * (REAL)12 * v[0] * u[1])
                      + g1 * (-b * (REAL)2 * v[0] * u[0] - cc * (REAL)12 * v[0] * v[1]);
            Db[0][1] += g0 * (-cc * (REAL)6 * rho);
            Db[1][1] += g1 * (-cc * (REAL)6 * rho);
            Db[1][2] += g0 * a;
            Db[0][2] += g1 * (-a);
            Db[0][3] += g0 * (-cc);
            Db[1][3] += g1 * (-cc);
        }
    } break;
    case ORC_EQ_LPD: { /* NLS/Ibc-test.py:163-164, NLS/testGNSS.py:139-140 (as coded: 6 rho v_xx but 10 rho u_xx):
                          fu = u_t + a v_xx + b rho v + c4 Gv - c3 (u_xxx + 6 rho u_x)
                          fv = v_t - a u_xx - b rho u - c4 Gu - c3 (v_xxx + 6 rho v_x)
                          Gv = v_4 + 6v(v_x^2-u_x^2) + 12 u u_x v_x + 4v(u_x^2+v_x^2) + p rho v_xx + 6 rho^2 v, Gu likewise */
        const REAL *v = D[1];
        REAL a = c[0], b = c[1], c3 = c[2], c4 = c[3], pp = c[4], qq = c[5];
        REAL rho = u[0] * u[0] + v[0] * v[0];
        REAL Gv = v[4] + (REAL)6 * v[0] * (v[1] * v[1] - u[1] * u[1]) + (REAL)12 * u[0] * u[1] * v[1]
                + (REAL)4 * v[0] * (u[1] * u[1] + v[1] * v[1]) + pp * rho * v[2] + (REAL)6 * rho * rho * v[0];
        REAL Gu = u[4] + (REAL)6 * u[0] * (u[1] * u[1] - v[1] * v[1]) + (REAL)12 * v[0] * u[1] * v[1]
                + (REAL)4 * u[0] * (u[1] * u[1] + v[1] * v[1]) + qq * rho * u[2] + (REAL)6 * rho * rho * u[0];
        f[0] = T[0][0] + a * v[2] + b * rho * v[0] + c4 * Gv - c3 * (u[3] + (REAL)6 * rho * u[1]);
        f[1] = T[1][0] - a * u[2] - b * rho * u[0] - c4 * Gu - c3 * (v[3] + (REAL)6 * rho * v[1]);
        if (fb) {
            REAL g0 = fb[0], g1 = fb[1];
            REAL Gv_u0 = (REAL)12 * u[1] * v[1] + (REAL)2 * pp * u[0] * v[2] + (REAL)24 * rho * u[0] * v[0];
            REAL Gv_v0 = (REAL)10 * v[1] * v[1] - (REAL)2 * u[1] * u[1] + (REAL)2 * pp * v[0] * v[2]
                       + (REAL)24 * rho * v[0] * v[0] + (REAL)6 * rho * rho;
            REAL Gv_u1 = (REAL)(-4) * v[0] * u[1] + (REAL)12 * u[0] * v[1];
            REAL Gv_v1 = (REAL)20 * v[0] * v[1] + (REAL)12 * u[0] * u[1];
            REAL Gu_u0 = (REAL)10 * u[1] * u[1] - (REAL)2 * v[1] * v[1] + (REAL)2 * qq * u[0] * u[2]
                       + (REAL)24 * rho * u[0] * u[0] + (REAL)6 * rho * rho;
            REAL Gu_v0 = (REAL)12 * u[1] * v[1] + (REAL)2 * qq * v[0] * u[2] + (REAL)24 * rho * u[0] * v[0];
            REAL Gu_u1 = (REAL)20 * u[0] * u[1] + (REAL)12 * v[0] * v[1];
            REAL Gu_v1 = (REAL)(-4) * u[0] * v[1] + (REAL)12 * v[0] * u[1];
            Tb[0][0] += g0;
            Tb[1][0] += g1;
            Db[0][0] += g0 * ((REAL)2 * b * u[0] * v[0] + c4 * Gv_u0 - (REAL)12 * c3 * u[0] * u[1])
                      + g1 * (-b * (rho + (REAL)2 * u[0] * u[0]) - c4 * Gu_u0 - (REAL)12 * c3 * u[0] * v[1]);
            Db[1][0] += g0 * (b * (rho + (REAL)2 * v[0] * v[0]) + c4 * Gv_v0 - (REAL)12 * c3 * v[0] * u[1])
                      + g1 * ((REAL)(-2) * b * v[0] * u[0] - c4 * Gu_v0 - (REAL)12 * c3 * v[0] * v[1]);
            Db[0][1] += g0 * (c4 * Gv_u1 - (REAL)6 * c3 * rho) + g1 * (-c4 * Gu_u1);
            Db[1][1] += g0 * (c4 * Gv_v1) + g1 * (-c4 * Gu_v1 - (REAL)6 * c3 * rho);
            Db[1][2] += g0 * (a + c4 * pp * rho);
            Db[0][2] += g1 * (-a - c4 * qq * rho);
            Db[0][3] += g0 * (-c3);
            Db[1][3] += g1 * (-c3);
            Db[1][4] += g0 * c4;
            Db[0][4] += g1 * (-c4);
        }
    } break;
    case ORC_EQ_NLSREG: { /* NLS/sw-test.py:159-189: fu, fv of the plain NLS and, as a third row,
                             f2 = sqrt(lambda (u_x^2 + u_xx^2 + u_t^2 + v_x^2 + v_xx^2 + v_t^2)): mean(f2^2) is the script's
                             reg_term (:173-180), which it adds to the loss (:147) */
        const REAL *v = D[1];
        REAL rho = u[0] * u[0] + v[0] * v[0];
        REAL S = u[1] * u[1] + u[2] * u[2] + T[0][0] * T[0][0] + v[1] * v[1] + v[2] * v[2] + T[1][0] * T[1][0];
        f[0] = T[0][0] + c[0] * v[2] + c[1] * rho * v[0];
        f[1] = T[1][0] - c[0] * u[2] - c[1] * rho * u[0];
        f[2] = SQRT_(c[2] * S);
        if (fb) {
            REAL g0 = fb[0], g1 = fb[1];
            REAL r = f[2] > 0 ? fb[2] * c[2] / f[2] : (REAL)0;
            Tb[0][0] += g0 + r * T[0][0];
            Tb[1][0] += g1 + r * T[1][0];
            Db[1][2] += g0 * c[0] + r * v[2];
            Db[0][2] += -g1 * c[0] + r * u[2];
            Db[0][1] += r * u[1];
            Db[1][1] += r * v[1];
            Db[0][0] += c[1] * (g0 * (REAL)2 * u[0] * v[0] - g1 * (rho + (REAL)2 * u[0] * u[0]));
            Db[1][0] += c[1] * (g0 * (rho + (REAL)2 * v[0] * v[0]) - g1 * (REAL)2 * u[0] * v[0]);
        }
    } break;
    default: break;
    }
}

/* f from the output derivatives: the plain polynomial, or (precise-top model, float instantiation) the double
 * polynomial on the double-accumulated output layer, rounded to float once */
static void SFX(residual_top)(const orc_desc *d, int K, int KT,
                              REAL D[ORC_MAXOUT][ORC_MAXJ], REAL T[ORC_MAXOUT][ORC_MAXJ],
                              double Dd[ORC_MAXOUT][ORC_MAXJ], double Td[ORC_MAXOUT][ORC_MAXJ], REAL *f)
{
#ifdef ORC_IS_F32
    if (ORC_TOP_ACC && d->n_lin >= 2) {
        double fd[ORC_MAXF] = {0, 0, 0};
        residual_f64(d, K, KT, Dd, Td, fd, NULL, NULL, NULL);
        for (int c = 0; c < ORC_MAXF; ++c) f[c] = (REAL)fd[c];
        return;
    }
#endif
    (void)Dd; (void)Td;
    SFX(residual)(d, K, KT, D, T, f, NULL, NULL, NULL);
}

/* ---- per-point forward through the whole net ------------------------------
 * work layout: for every linear layer l the input jets A_l[i][j] (i<in_l,
 * j<J) and pre-activation jets Z_l[o][j]; J = K+1+KT, components
 * [0..K] = x orders, [K+1..] = t-tangent of x orders 0..KT-1.
 * Scale: H = 2 (X - lb)/(ub - lb) - 1 (7kdv/PI-7kdv3soliton.py:39-41),
 * Linear: y = x W^T + b with W (out,in) row-major (:49-56). */
typedef struct {
    REAL *A[ORC_MAX_LAYERS];   /* inputs of layer l  [in_l * J]  */
    REAL *Z[ORC_MAX_LAYERS];   /* pre-activations    [out_l * J] */
    REAL *Y[ORC_MAX_LAYERS];   /* tanh outputs y     [out_l * J] (l < n_lin-1) */
    REAL *Wj[ORC_MAX_LAYERS];  /* w = 1-y^2 jets     [out_l * (K+1)] */
} SFX(orc_work);

static void SFX(point_forward)(const orc_desc *d, int K, int KT, const REAL *theta,
                               REAL x, REAL t, SFX(orc_work) *wk,
                               REAL D[ORC_MAXOUT][ORC_MAXJ], REAL T[ORC_MAXOUT][ORC_MAXJ],
                               double Dd[ORC_MAXOUT][ORC_MAXJ], double Td[ORC_MAXOUT][ORC_MAXJ])
{
    const int J = K + 1 + KT;
    const REAL lbx = (REAL)d->lb[0], lbt = (REAL)d->lb[1];
    const REAL ubx = (REAL)d->ub[0], ubt = (REAL)d->ub[1];
    REAL *a0 = wk->A[0];
    for (int j = 0; j < 2 * J; ++j) a0[j] = 0;
    a0[0 * J + 0] = (REAL)2 * (x - lbx) / (ubx - lbx) - (REAL)1;
    if (K >= 1) a0[0 * J + 1] = (REAL)2 / (ubx - lbx);
    a0[1 * J + 0] = (REAL)2 * (t - lbt) / (ubt - lbt) - (REAL)1;
    if (KT >= 1) a0[1 * J + K + 1] = (REAL)2 / (ubt - lbt);

    const REAL *p = theta;
    for (int l = 0; l < d->n_lin; ++l) {
        const int nin = d->widths[l], nout = d->widths[l + 1];
        const REAL *W = p, *b = p + (size_t)nin * nout;
        p = b + nout;
        const REAL *a = wk->A[l];
        REAL *z = wk->Z[l];
        /* (precise-top model: the last hidden Linear layer and the output layer accumulate in double) */
        const int top_acc = ORC_TOP_ACC && l >= d->n_lin - 2 && l >= 1;
        for (int o = 0; o < nout; ++o) {
            for (int j = 0; j < J; ++j) {
                /* last hidden layer: only the value, the first x-derivative and the first t-derivative of the
                 * pre-activation (the kernels' choice, kernel_mfma.hpp); output layer: every component */
                if (top_acc && (l == d->n_lin - 1 || j == 0 || j == 1 || j == K + 1)) {
                    double s = (j == 0) ? (double)b[o] : 0.0;
                    for (int i = 0; i < nin; ++i) s += (double)W[(size_t)o * nin + i] * (double)a[i * J + j];
                    z[o * J + j] = (REAL)s;
                    if (l == d->n_lin - 1 && Dd) {
                        double fact = 1;
                        const int k = j <= K ? j : j - K - 1;
                        for (int q = 2; q <= k; ++q) fact *= q;
                        if (j <= K) Dd[o][k] = fact * s; else Td[o][k] = fact * s;
                    }
                } else {
                    REAL s = (j == 0) ? b[o] : (REAL)0;
                    for (int i = 0; i < nin; ++i) s += W[(size_t)o * nin + i] * a[i * J + j];
                    z[o * J + j] = s;
                }
            }
        }
        if (l < d->n_lin - 1) {
            REAL *y = wk->Y[l], *w = wk->Wj[l];
            for (int o = 0; o < nout; ++o)
                SFX(tanh_jet_fwd)(K, KT, z + o * J, z + o * J + K + 1,
                                  y + o * J, w + o * (K + 1), y + o * J + K + 1);
            wk->A[l + 1] = y;
        } else {
            REAL fact = 1;
            for (int k = 0; k <= K; ++k) {
                if (k > 0) fact *= (REAL)k;
                for (int o = 0; o < nout; ++o) {
                    D[o][k] = fact * z[o * J + k];
                    if (k < KT) T[o][k] = fact * z[o * J + K + 1 + k];
                }
            }
        }
    }
}

/* reverse sweep for one point: Db/Tb adjoints of the output derivatives ->
 * accumulate into grad (same layout as theta). */
static void SFX(point_backward)(const orc_desc *d, int K, int KT, const REAL *theta,
                                SFX(orc_work) *wk, REAL Db[ORC_MAXOUT][ORC_MAXJ],
                                REAL Tb[ORC_MAXOUT][ORC_MAXJ], REAL *grad,
                                REAL *zb_buf, REAL *ab_buf)
{
    const int J = K + 1 + KT;
    const int L = d->n_lin;
    size_t off[ORC_MAX_LAYERS + 1];
    off[0] = 0;
    for (int l = 0; l < L; ++l)
        off[l + 1] = off[l] + (size_t)d->widths[l] * d->widths[l + 1] + d->widths[l + 1];

    /* adjoint of the last layer's pre-activation = factorial-scaled Db/Tb */
    REAL *zb = zb_buf;
    {
        const int nout = d->widths[L];
        REAL fact = 1;
        for (int k = 0; k <= K; ++k) {
            if (k > 0) fact *= (REAL)k;
            for (int o = 0; o < nout; ++o) {
                zb[o * J + k] = fact * Db[o][k];
                if (k < KT) zb[o * J + K + 1 + k] = fact * Tb[o][k];
            }
        }
    }
    for (int l = L - 1; l >= 0; --l) {
        const int nin = d->widths[l], nout = d->widths[l + 1];
        const REAL *W = theta + off[l];
        REAL *gW = grad + off[l], *gb = gW + (size_t)nin * nout;
        const REAL *a = wk->A[l];
        REAL *ab = ab_buf;
        for (int i = 0; i < nin * J; ++i) ab[i] = 0;
        for (int o = 0; o < nout; ++o) {
            gb[o] += zb[o * J + 0];
            for (int i = 0; i < nin; ++i) {
                REAL s = 0;
                const REAL w = W[(size_t)o * nin + i];
                for (int j = 0; j < J; ++j) {
                    s += zb[o * J + j] * a[i * J + j];
                    ab[i * J + j] += w * zb[o * J + j];
                }
                gW[(size_t)o * nin + i] += s;
            }
        }
        if (l == 0) break;
        /* through tanh of layer l-1: ab is adjoint of Y[l-1] */
        const int nprev = d->widths[l];
        const REAL *zp = wk->Z[l - 1], *yp = wk->Y[l - 1], *wp = wk->Wj[l - 1];
        for (int o = 0; o < nprev; ++o)
            SFX(tanh_jet_bwd)(K, KT, zp + o * J, zp + o * J + K + 1, yp + o * J, wp + o * (K + 1),
                              ab + o * J, ab + o * J + K + 1, zb + o * J, zb + o * J + K + 1);
    }
}

static int SFX(orc_alloc_work)(const orc_desc *d, int J, SFX(orc_work) *wk, REAL **zb, REAL **ab)
{
    int maxw = 2;
    for (int l = 0; l <= d->n_lin; ++l) if (d->widths[l] > maxw) maxw = d->widths[l];
    for (int l = 0; l < d->n_lin; ++l) {
        wk->A[l] = NULL;
        wk->Z[l] = (REAL *)malloc(sizeof(REAL) * (size_t)d->widths[l + 1] * J);
        wk->Y[l] = (REAL *)malloc(sizeof(REAL) * (size_t)d->widths[l + 1] * J);
        wk->Wj[l] = (REAL *)malloc(sizeof(REAL) * (size_t)d->widths[l + 1] * J);
    }
    wk->A[0] = (REAL *)malloc(sizeof(REAL) * 2 * (size_t)J);
    *zb = (REAL *)malloc(sizeof(REAL) * (size_t)maxw * J);
    *ab = (REAL *)malloc(sizeof(REAL) * (size_t)maxw * J);
    return 0;
}

static void SFX(orc_free_work)(const orc_desc *d, SFX(orc_work) *wk, REAL *zb, REAL *ab)
{
    free(wk->A[0]);
    for (int l = 0; l < d->n_lin; ++l) { free(wk->Z[l]); free(wk->Y[l]); free(wk->Wj[l]); }
    free(zb); free(ab);
}

/* derivs_out: [n][nout][J] true derivatives: D_0..D_K then T_0..T_{KT-1}
 * f_out:      [n][nf]      (either may be NULL) */
int SFX(orc_forward)(const orc_desc *d, const REAL *theta, const REAL *x, const REAL *t,
                     long n, REAL *derivs_out, REAL *f_out)
{
    int K, KT, nf;
    if (orc_orders(d, &K, &KT, &nf)) return -1;
    const int J = K + 1 + KT, nout = d->widths[d->n_lin];
#pragma omp parallel
    {
        SFX(orc_work) wk; REAL *zb, *ab;
        SFX(orc_alloc_work)(d, J, &wk, &zb, &ab);
        REAL D[ORC_MAXOUT][ORC_MAXJ], T[ORC_MAXOUT][ORC_MAXJ], f[ORC_MAXF];
#pragma omp for schedule(static)
        for (long p = 0; p < n; ++p) {
            double Dd[ORC_MAXOUT][ORC_MAXJ], Td[ORC_MAXOUT][ORC_MAXJ];
            SFX(point_forward)(d, K, KT, theta, x[p], t[p], &wk, D, T, Dd, Td);
            if (derivs_out)
                for (int o = 0; o < nout; ++o) {
                    for (int k = 0; k <= K; ++k) derivs_out[((size_t)p * nout + o) * J + k] = D[o][k];
                    for (int k = 0; k < KT; ++k) derivs_out[((size_t)p * nout + o) * J + K + 1 + k] = T[o][k];
                }
            if (f_out) {
                SFX(residual_top)(d, K, KT, D, T, Dd, Td, f);
                for (int c = 0; c < nf; ++c) f_out[(size_t)p * nf + c] = f[c];
            }
        }
        SFX(orc_free_work)(d, &wk, zb, ab);
    }
    return 0;
}

int SFX(orc_residual_derivs)(const orc_desc *d, const REAL *derivs, long n, REAL *f_out)
{
    int K, KT, nf;
    if (orc_orders(d, &K, &KT, &nf)) return -1;
    const int J = K + 1 + KT, nout = d->widths[d->n_lin];
    for (long p = 0; p < n; ++p) {
        REAL D[ORC_MAXOUT][ORC_MAXJ], T[ORC_MAXOUT][ORC_MAXJ], f[ORC_MAXF];
        for (int o = 0; o < nout; ++o) {
            for (int k = 0; k <= K; ++k) D[o][k] = derivs[((size_t)p * nout + o) * J + k];
            for (int k = 0; k < KT; ++k) T[o][k] = derivs[((size_t)p * nout + o) * J + K + 1 + k];
        }
        SFX(residual)(d, K, KT, D, T, f, NULL, NULL, NULL);
        for (int c = 0; c < nf; ++c) f_out[(size_t)p * nf + c] = f[c];
    }
    return 0;
}

/* gbar == NULL: loss mode, cotangent of f_c is 2*scale*f_c and
 *               loss_sums[c] += sum_p f_c^2 (unscaled).
 * gbar != NULL: vjp mode, cotangent given [n][nf].
 * grad (same layout as theta, P entries) is ACCUMULATED into. */
int SFX(orc_loss_grad)(const orc_desc *d, const REAL *theta, const REAL *x, const REAL *t,
                       long n, const REAL *gbar, double scale, double *loss_sums, REAL *grad)
{
    int K, KT, nf;
    if (orc_orders(d, &K, &KT, &nf)) return -1;
    const int J = K + 1 + KT, nout = d->widths[d->n_lin];
    size_t P = 0;
    for (int l = 0; l < d->n_lin; ++l) P += (size_t)d->widths[l] * d->widths[l + 1] + d->widths[l + 1];
#pragma omp parallel
    {
        SFX(orc_work) wk; REAL *zb, *ab;
        SFX(orc_alloc_work)(d, J, &wk, &zb, &ab);
        REAL *g = (REAL *)calloc(P, sizeof(REAL));
        double ls[ORC_MAXF] = {0};
        REAL D[ORC_MAXOUT][ORC_MAXJ], T[ORC_MAXOUT][ORC_MAXJ], f[ORC_MAXF], fb[ORC_MAXF];
        REAL Db[ORC_MAXOUT][ORC_MAXJ], Tb[ORC_MAXOUT][ORC_MAXJ];
#pragma omp for schedule(static)
        for (long p = 0; p < n; ++p) {
            double Dd[ORC_MAXOUT][ORC_MAXJ], Td[ORC_MAXOUT][ORC_MAXJ];
            SFX(point_forward)(d, K, KT, theta, x[p], t[p], &wk, D, T, Dd, Td);
            SFX(residual_top)(d, K, KT, D, T, Dd, Td, f);
            for (int c = 0; c < nf; ++c) {
                if (gbar) fb[c] = gbar[(size_t)p * nf + c];
                else { fb[c] = (REAL)(2.0 * scale) * f[c]; ls[c] += (double)f[c] * (double)f[c]; }
            }
            for (int o = 0; o < nout; ++o)
                for (int j = 0; j < ORC_MAXJ; ++j) { Db[o][j] = 0; Tb[o][j] = 0; }
            SFX(residual)(d, K, KT, D, T, f, fb, Db, Tb);
            SFX(point_backward)(d, K, KT, theta, &wk, Db, Tb, g, zb, ab);
        }
#pragma omp critical
        {
            for (size_t i = 0; i < P; ++i) grad[i] += g[i];
            if (loss_sums) for (int c = 0; c < nf; ++c) loss_sums[c] += ls[c];
        }
        free(g);
        SFX(orc_free_work)(d, &wk, zb, ab);
    }
    return 0;
}
