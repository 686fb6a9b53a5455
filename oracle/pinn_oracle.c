/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See pinn_oracle_impl.h.
 *
 * Plain-C CPU restatement of the PINN collocation hot path of Emily-Qu/PDE
 * (net_u / net_f / mean-square loss / weight gradient), instantiated in fp32
 * (the reference's arithmetic type: `.float()` everywhere,
 * 7kdv/PI-7kdv3soliton.py:85-95) and in fp64 (truth for parity tests).
 *
 * Parity pin: tests/test_oracle_golden.py checks every entry point against
 * tests/golden/ *.npz, which oracle/gen_golden.py produced by importing the
 * reference scripts themselves in the build container.
 *
 * Build: make -C oracle   ->  oracle/libpinn_oracle.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pinn_oracle.h"

int orc_orders(const orc_desc *d, int *K, int *KT, int *nf)
{
    if (d->n_lin < 2 || d->n_lin > ORC_MAX_LAYERS) return -1;
    if (d->widths[0] != 2) return -1;
    const int nout = d->widths[d->n_lin];
    if (nout < 1 || nout > ORC_MAXOUT) return -1;
    switch (d->eq) {
    case ORC_EQ_NONE:    *K = 0; *KT = 0; *nf = nout; break;
    case ORC_EQ_BURGERS: *K = 2; *KT = 1; *nf = 1; break;
    case ORC_EQ_KDV5:    *K = 5; *KT = 1; *nf = 1; break;
    case ORC_EQ_KDV7:    *K = 7; *KT = 1; *nf = 1; break;
    case ORC_EQ_CH:      *K = 3; *KT = 3; *nf = 1; break;
    case ORC_EQ_GNSS:    *K = 3; *KT = 1; *nf = 2; if (nout != 2) return -1; break;
    case ORC_EQ_LPD:     *K = 4; *KT = 1; *nf = 2; if (nout != 2) return -1; break;
    case ORC_EQ_NLSREG:  *K = 2; *KT = 1; *nf = 3; if (nout != 2) return -1; break;
    default: return -1;
    }
    return 0;
}

/* Model of the kernels' "precise top" arithmetic (pde_amd/csrc, EqTraits<EQ>::PRECISE), for the float
 * instantiation only: the dot products of the LAST hidden Linear layer (jet components 0, 1 and K+1: the
 * pre-activation's value, x-slope and t-slope) and of the output layer (all components) are accumulated in
 * double (their results are still stored as float) and the residual polynomial is evaluated in double.  Everything
 * else stays float.  Off by default: orc_*_f32 is then the plain float restatement. */
static int orc_f32_precise_top = 0;
void orc_set_f32_precise_top(int on) { orc_f32_precise_top = on; }

#define REAL double
#define SFX(n) n##_f64
#define TANH tanh
#define SQRT_ sqrt
#define ORC_TOP_ACC 0
#include "pinn_oracle_impl.h"
#undef REAL
#undef SFX
#undef TANH
#undef SQRT_
#undef ORC_TOP_ACC

#define REAL float
#define SFX(n) n##_f32
#define TANH tanhf
#define SQRT_ sqrtf
#define ORC_TOP_ACC orc_f32_precise_top
#define ORC_IS_F32 1
#include "pinn_oracle_impl.h"
#undef REAL
#undef SFX
#undef TANH
#undef SQRT_
#undef ORC_TOP_ACC
#undef ORC_IS_F32
