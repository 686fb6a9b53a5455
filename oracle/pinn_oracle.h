/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  C interface of the CPU oracle. */
#ifndef PINN_ORACLE_H
#define PINN_ORACLE_H

#define ORC_MAX_LAYERS 16
#define ORC_MAXJ 12
#define ORC_MAXOUT 2
#define ORC_MAXF 3     /* residual rows per point: two outputs + the penalty row of ORC_EQ_NLSREG */

enum {
    ORC_EQ_NONE = 0,    /* plain net_u, no derivatives                                  */
    ORC_EQ_BURGERS = 1, /* coef[0] = nu                                                  */
    ORC_EQ_KDV5 = 2,    /* coef[0..3] = 180, 30, 30, 1                                   */
    ORC_EQ_KDV7 = 3,    /* coef[0..7]: 3-soliton 5/98,5/14,10/7,5/14,5,3,1,1;
                                       1-soliton 140,70,280,70,70,42,14,1                */
    ORC_EQ_CH = 4,      /* coef[0] = kappa (0 for PI script, 2 for the bc script)        */
    ORC_EQ_GNSS = 5,    /* coef[0..2] = a(0.5), b(1), c(0.1; 0 gives plain NLS)          */
    ORC_EQ_LPD = 6,     /* coef[0..5] = a, b, c3, c4, p, q = 0.5, 1, 0.125, 0.0625, 6, 10 */
    ORC_EQ_NLSREG = 7   /* coef[0..2] = a, b, lambda (1e-9); three residual rows, NLS/sw-test.py:159-189 */
};

typedef struct {
    int eq;
    int n_lin;                       /* number of Linear layers = len(layers)-1 */
    int widths[ORC_MAX_LAYERS + 1];  /* the script's `layers` list              */
    double lb[2], ub[2];             /* float32-representable domain bounds     */
    double coef[8];
} orc_desc;

int orc_orders(const orc_desc *d, int *K, int *KT, int *nf);
/* float instantiation only: accumulate the last hidden Linear layer and the output layer in double and evaluate
 * the residual polynomial in double (model of the kernels' EqTraits<EQ>::PRECISE arithmetic); 0 = plain float */
void orc_set_f32_precise_top(int on);

int orc_forward_f32(const orc_desc *d, const float *theta, const float *x, const float *t,
                    long n, float *derivs_out, float *f_out);
int orc_forward_f64(const orc_desc *d, const double *theta, const double *x, const double *t,
                    long n, double *derivs_out, double *f_out);
int orc_loss_grad_f32(const orc_desc *d, const float *theta, const float *x, const float *t,
                      long n, const float *gbar, double scale, double *loss_sums, float *grad);
int orc_loss_grad_f64(const orc_desc *d, const double *theta, const double *x, const double *t,
                      long n, const double *gbar, double scale, double *loss_sums, double *grad);
/* the residual polynomial alone, on caller-supplied true derivatives derivs[n][nout][J]
 * (layout of derivs_out): f_out[n][nf].  Used by the net-free known-answer tests. */
int orc_residual_derivs_f32(const orc_desc *d, const float *derivs, long n, float *f_out);
int orc_residual_derivs_f64(const orc_desc *d, const double *derivs, long n, double *f_out);
#endif
