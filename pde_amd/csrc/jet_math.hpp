// Per-lane Taylor-jet arithmetic shared by the VALU and MFMA kernels.
//
// A "jet" of a scalar g(x,t) at one collocation point is the vector
//   [ c_0 .. c_K | d_0 .. d_{KT-1} ],   c_k = (d^k g/dx^k)/k!,  d_k = (d^k (dg/dt)/dx^k)/k!
// i.e. normalised Taylor coefficients in x plus the x-Taylor coefficients of
// the t-derivative.  J = K+1+KT components.  Linear layers act on each
// component separately; tanh acts through the recurrences below.  This
// replaces the reference's nested torch.autograd.grad chain
// (7kdv/PI-7kdv3soliton.py:122-132) by one forward sweep per point.
#pragma once
#include <hip/hip_runtime.h>

namespace pinn {

enum { EQ_NONE = 0, EQ_BURGERS = 1, EQ_KDV5 = 2, EQ_KDV7 = 3, EQ_CH = 4, EQ_GNSS = 5, EQ_LPD = 6, EQ_NLSREG = 7 };

template <int EQ> struct EqTraits;
// NF = residual components per point (0: as many as the net has outputs); NOUT = most network outputs the
// equation's nets can have (pinn_create checks the exact number)
// PRECISE = "precise top": the dot products of the LAST hidden Linear layer and of the output layer are accumulated
// in float64 (f64 MFMA / v_fma_f64; results stored as float32 like everything else) and the residual polynomial is
// evaluated in float64.  For Camassa-Holm: f = (u - u_xx)_t + ... is a cancellation of u_xxt against u u_xxx
// (~1500 against ~1400 at the crest of the trained peakon) whose two sides come out of DIFFERENT jet components,
// i.e. different float32 dot products with independent rounding errors, while the reference's nested-autograd
// graph perturbs both consistently.  Stage by stage (tools/ch_noise_lab.py) the excess noise
// sits in exactly those late dot products: made exact, the residual / loss / gradient noise at CH/model/PINN CH.pth
// falls from 3.4x / 4.8x / 5.7x the reference's own fp32 noise to 1.3x / 1.5x / 1.5x (96 batches of 256 points).
// It is on for every residual: at the other seven shipped checkpoints it takes the noise from 0.7-1.8x the
// reference's to 0.4-1.4x (profiles/r03_noise_report.txt) for 1-2 % of the kernel time (three of the J components of
// one of nine GEMM phases at the f64 rate).  Off for plain net_u (EQ_NONE: nothing cancels) and, in the MFMA
// kernels, for the wide shapes (width 100: no registers to spare, +10 % time; those nets sit at 0.9-1.2x anyway).
template <> struct EqTraits<EQ_NONE>    { static constexpr int K = 0, KT = 0, NF = 0, NOUT = 2; static constexpr bool PRECISE = false; };
template <> struct EqTraits<EQ_BURGERS> { static constexpr int K = 2, KT = 1, NF = 1, NOUT = 1; static constexpr bool PRECISE = true; };
template <> struct EqTraits<EQ_KDV5>    { static constexpr int K = 5, KT = 1, NF = 1, NOUT = 1; static constexpr bool PRECISE = true; };
template <> struct EqTraits<EQ_KDV7>    { static constexpr int K = 7, KT = 1, NF = 1, NOUT = 1; static constexpr bool PRECISE = true; };
template <> struct EqTraits<EQ_CH>      { static constexpr int K = 3, KT = 3, NF = 1, NOUT = 1; static constexpr bool PRECISE = true; };
template <> struct EqTraits<EQ_GNSS>    { static constexpr int K = 3, KT = 1, NF = 2, NOUT = 2; static constexpr bool PRECISE = true; };
template <> struct EqTraits<EQ_LPD>     { static constexpr int K = 4, KT = 1, NF = 2, NOUT = 2; static constexpr bool PRECISE = true; };
template <> struct EqTraits<EQ_NLSREG>  { static constexpr int K = 2, KT = 1, NF = 3, NOUT = 2; static constexpr bool PRECISE = true; };

constexpr int MAXOUT = 2;
constexpr int MAXF = 3;     // residual components per point: two outputs + one penalty row (EQ_NLSREG)

// tanh(x) and w0 = 1 - tanh(x)^2 = sech(x)^2, both to ~1 ulp, branch-free.
//
// Why not exp-based (1 - e)/(1 + e) with the hardware v_exp_f32 (what round 1 shipped): e carries >= 1 ulp
// of ABSOLUTE error near 1, which 1 - e turns into several ulp of tanh for |x| in [0.04, 0.3], and
// w0 = 1 - y^2 then inherits 2|y| ulp(1) of absolute error however small w0 is.  At trained weights the
// PDE residual is a ~1000x cancellation of terms that are all linear in these two numbers, and the extra
// ulps showed up as 2-3x the residual / gradient noise of the reference's own fp32 run
// (oracle variants: a +2 ulp bias in tanh alone gives 2-8x).
//
// Here: a = -2|x| = n ln2 + r, |r| <= ln2/2;  p = expm1(r) by its Taylor polynomial through r^8/8!
// (relative error < 0.3 ulp, r is exact for n = 0);  e = exp(a) = s (1 + p), s = 2^n.  Then
//     1 - e = (1 - s) - s p     (exact for n = 0: no cancellation error at all),
//     1 + e = (1 + s) + s p,    tanh = (1 - e)/(1 + e),    sech^2 = 4 e / (1 + e)^2,
// with one Newton step on the reciprocal and one on the quotient.  ~31 VALU instructions (one
// transcendental).  Accuracy on the GPU: tools/ubench/tanh_accuracy.hip.
__device__ __forceinline__ float tanh_sech2(float x, float &w0)
{
    // |x| > 22: tanh = 1 and sech^2 < 4 e^-44 = 3e-19 in float.  The ARGUMENT is clamped (a select, so that NaN
    // stays NaN), not just the exponent: with only n clamped the remainder r = a - n ln2 of a large |x| is not
    // small, the expm1 polynomial blows up and tanh(500) came out as 0.04 (round-2 regression).
    const float ax0 = fabsf(x);
    const float ax = ax0 > 22.0f ? 22.0f : ax0;
    const float a = -2.0f * ax;
    const float nf = __builtin_rintf(a * 1.44269504f);   // >= -64
    float r = fmaf(nf, -0.693145752f, a);         // ln2 = hi + lo, hi with 9 trailing zero bits: n * hi is exact
    r = fmaf(nf, -1.42860677e-6f, r);
    float q = 1.0f / 40320.0f;
    q = fmaf(q, r, 1.0f / 5040.0f);
    q = fmaf(q, r, 1.0f / 720.0f);
    q = fmaf(q, r, 1.0f / 120.0f);
    q = fmaf(q, r, 1.0f / 24.0f);
    q = fmaf(q, r, 1.0f / 6.0f);
    q = fmaf(q, r, 0.5f);
    const float p = fmaf(r * r, q, r);            // expm1(r)
    const float s = __builtin_amdgcn_ldexpf(1.0f, (int)nf);
    const float e4 = fmaf(4.0f * s, p, 4.0f * s); // 4 e
    const float num = fmaf(-s, p, 1.0f - s);
    const float den = fmaf(s, p, 1.0f + s);
    float rc = __builtin_amdgcn_rcpf(den);
    rc = fmaf(fmaf(-den, rc, 1.0f), rc, rc);
    float y = num * rc;
    y = fmaf(fmaf(-den, y, num), rc, y);
    // (exactly 0 beyond the clamp, like 1 - tanhf(x)^2: a constant 3e-19 there would be multiplied by powers of
    // the pre-activation's slope in the jet recurrences)
    w0 = ax0 > 22.0f ? 0.0f : (e4 * rc) * rc;
    return copysignf(y, x);
}

// ---- tanh on a jet -----------------------------------------------------------
// y' = (1-y^2) z'  =>  k y_k = sum_{j=1..k} j z_j w_{k-j},  w = 1 - y*y.
// z, y: [J] (x orders then t tangents);  w: [K+1].
//
// Operation counts matter here (the kernels are instruction-issue bound): the convolution
// w_k = -sum_i y_i y_{k-i} is taken over HALF its terms with pre-doubled factors
// (ceil((k+1)/2) FMAs instead of k+1), the weights j z_j are formed once, and w_K — which no
// y_k uses unless a t-tangent of that order is carried — is skipped.
template <int K, int KT>
__device__ __forceinline__ void w_order(const float *y, const float (&ym2)[K / 2 + 1], int k, float &wk)
{
    // w_k = sum_{i < k/2 (strict half)} (-2 y_i) y_{k-i}  [ - y_{k/2}^2 when k is even ]
    float q = ym2[0] * y[k];
#pragma unroll
    for (int i = 1; 2 * i < k; ++i) q = fmaf(ym2[i], y[k - i], q);
    if ((k & 1) == 0) q = fmaf(-y[k / 2], y[k / 2], q);
    wk = q;
}

// ZN / ZTN: the input jet is structurally zero above x-order ZN and in t-tangents of order >= ZTN
// (first layer: z = W0 [x,t] + b is affine in x and t, so ZN = 1, ZTN = 1); those terms are not issued.
template <int K, int KT, int ZN = K, int ZTN = KT>
__device__ __forceinline__ void tanh_jet_fwd(const float (&z)[K + 1 + KT], float (&y)[K + 1 + KT],
                                             float (&w)[K + 1])
{
    constexpr int KW0 = KT - 1 > K - 1 ? KT - 1 : K - 1;
    constexpr int KW = KW0 < K ? KW0 : K;   // highest w order anything reads: y_k uses w_0..w_{k-1}, t-tangents w_0..w_{KT-1}
    y[0] = tanh_sech2(z[0], w[0]);   // branch-free: keeps a whole tile's tanh work in one scheduling region
    float zs[K + 1], ym2[K / 2 + 1];
#pragma unroll
    for (int j = 1; j <= ZN; ++j) zs[j] = (float)j * z[j];
    ym2[0] = -2.0f * y[0];
#pragma unroll
    for (int k = 1; k <= K; ++k) {
        float s = ZN >= 1 ? zs[ZN >= 1 ? 1 : 0] * w[k - 1] : 0.0f;
#pragma unroll
        for (int j = 2; j <= (k < ZN ? k : ZN); ++j) s = fmaf(zs[j], w[k - j], s);
        y[k] = s * (1.0f / (float)k);
        if (k <= K / 2) ym2[k] = -2.0f * y[k];
        if (k <= KW) w_order<K, KT>(y, ym2, k, w[k]);
        else w[k] = 0.0f;     // never read
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int j = (k - ZTN + 1 > 0 ? k - ZTN + 1 : 0); j <= k; ++j) s = fmaf(w[j], z[K + 1 + k - j], s);
        y[K + 1 + k] = s;
    }
}

// reverse of tanh_jet_fwd: yb (adjoint of y, modified in place) -> zb.  With ZN < K / ZTN < KT the adjoints
// of the structurally-zero inputs are not formed (returned as 0).
template <int K, int KT, int ZN = K, int ZTN = KT>
__device__ __forceinline__ void tanh_jet_bwd(const float (&z)[K + 1 + KT], const float (&y)[K + 1 + KT],
                                             const float (&w)[K + 1], float (&yb)[K + 1 + KT],
                                             float (&zb)[K + 1 + KT])
{
    // wb: adjoint of w; zsb: adjoint of the weighted coefficient zs_j = j z_j (scaled by j at the end)
    float wb[K + 1], zs[K + 1], zsb[K + 1];
#pragma unroll
    for (int k = 0; k <= K; ++k) { wb[k] = 0.0f; zsb[k] = 0.0f; }
#pragma unroll
    for (int j = 1; j <= ZN; ++j) zs[j] = (float)j * z[j];
#pragma unroll
    for (int k = 0; k < KT; ++k) zb[K + 1 + k] = 0.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
#pragma unroll
        for (int j = (k - ZTN + 1 > 0 ? k - ZTN + 1 : 0); j <= k; ++j) {
            wb[j] = fmaf(yb[K + 1 + k], z[K + 1 + k - j], wb[j]);
            zb[K + 1 + k - j] = fmaf(yb[K + 1 + k], w[j], zb[K + 1 + k - j]);
        }
    }
#pragma unroll
    for (int k = K; k >= 1; --k) {
        // wb[k] is still exactly zero when nothing read w_k: k = K with no t-tangent of that order
        if (k < K || KT > K) {
            const float m2 = -2.0f * wb[k];
#pragma unroll
            for (int m = 0; m <= k; ++m) yb[m] = fmaf(m2, y[k - m], yb[m]);
        }
        const float g = yb[k] * (1.0f / (float)k);
#pragma unroll
        for (int j = 1; j <= (k < ZN ? k : ZN); ++j) {
            zsb[j] = fmaf(g, w[k - j], zsb[j]);
            wb[k - j] = fmaf(g, zs[j], wb[k - j]);
        }
    }
    yb[0] = fmaf(-2.0f * wb[0], y[0], yb[0]);
    zb[0] = yb[0] * w[0];
#pragma unroll
    for (int j = 1; j <= K; ++j) zb[j] = j <= ZN ? (float)j * zsb[j] : 0.0f;
}

__host__ __device__ constexpr float factorial(int k)
{
    float f = 1.0f;
    for (int i = 2; i <= k; ++i) f *= (float)i;
    return f;
}

__device__ __forceinline__ float sqrt_r(float v) { return sqrtf(v); }
__device__ __forceinline__ double sqrt_r(double v) { return sqrt(v); }

// ---- residuals ---------------------------------------------------------------
// D[o][k] = d^k u_o/dx^k, T[o][k] = d/dt d^k u_o/dx^k (true derivatives).
// eval() writes f[0..NF-1] (in the arithmetic type R of its arguments: float, or double for the PRECISE
// equations); adj() accumulates fb[c]*df_c/d(.) into Db/Tb.
template <int EQ> struct Residual;

template <> struct Residual<EQ_NONE> {  // plain net_u: f_o = u_o  (self.dnn(X), :99)
    static constexpr int K = 0, KT = 0;
    template <class A, class R> __device__ static void eval(const A &D, const A &, const float *, int nout, R *f)
    {
        f[0] = D[0][0];
        if (nout > 1) f[1] = D[1][0];
    }
    template <class A> __device__ static void adj(const A &, const A &, const float *, int nout,
                                                  const float *fb, A &Db, A &)
    {
        Db[0][0] += fb[0];
        if (nout > 1) Db[1][0] += fb[1];
    }
};

template <> struct Residual<EQ_BURGERS> {  // Burgers/Burgers_equation(PINN).py:126-131
    static constexpr int K = 2, KT = 1;
    template <class A, class R> __device__ static void eval(const A &D, const A &T, const float *c, int, R *f)
    {
        const R *u = D[0];
        f[0] = T[0][0] + u[0] * u[1] - c[0] * u[2];
    }
    template <class A> __device__ static void adj(const A &D, const A &, const float *c, int,
                                                  const float *fb, A &Db, A &Tb)
    {
        const float *u = D[0];
        const float g = fb[0];
        Tb[0][0] += g;
        Db[0][0] += g * u[1];
        Db[0][1] += g * u[0];
        Db[0][2] += g * (-c[0]);
    }
};

template <> struct Residual<EQ_KDV5> {  // 5kdv/PI-51.py:132-140
    static constexpr int K = 5, KT = 1;
    template <class A, class R> __device__ static void eval(const A &D, const A &T, const float *c, int, R *f)
    {
        const R *u = D[0];
        f[0] = T[0][0] + c[0] * u[0] * u[0] * u[1] + c[1] * u[1] * u[2] + c[2] * u[0] * u[3] + c[3] * u[5];
    }
    template <class A> __device__ static void adj(const A &D, const A &, const float *c, int,
                                                  const float *fb, A &Db, A &Tb)
    {
        const float *u = D[0];
        const float g = fb[0];
        Tb[0][0] += g;
        Db[0][0] += g * (2.0f * c[0] * u[0] * u[1] + c[2] * u[3]);
        Db[0][1] += g * (c[0] * u[0] * u[0] + c[1] * u[2]);
        Db[0][2] += g * (c[1] * u[1]);
        Db[0][3] += g * (c[2] * u[0]);
        Db[0][5] += g * c[3];
    }
};

template <> struct Residual<EQ_KDV7> {  // 7kdv/PI-7kdv3soliton.py:132, 7kdv/PI-7kdv1soliton.py:132
    static constexpr int K = 7, KT = 1;
    template <class A, class R> __device__ static void eval(const A &D, const A &T, const float *c, int, R *f)
    {
        const R *u = D[0];
        f[0] = T[0][0] + c[0] * u[0] * u[0] * u[0] * u[1] + c[1] * u[1] * u[1] * u[1]
             + c[2] * u[0] * u[1] * u[2] + c[3] * u[0] * u[0] * u[3] + c[4] * u[2] * u[3]
             + c[5] * u[1] * u[4] + c[6] * u[0] * u[5] + c[7] * u[7];
    }
    template <class A> __device__ static void adj(const A &D, const A &, const float *c, int,
                                                  const float *fb, A &Db, A &Tb)
    {
        const float *u = D[0];
        const float g = fb[0];
        Tb[0][0] += g;
        Db[0][0] += g * (3.0f * c[0] * u[0] * u[0] * u[1] + c[2] * u[1] * u[2] + 2.0f * c[3] * u[0] * u[3] + c[6] * u[5]);
        Db[0][1] += g * (c[0] * u[0] * u[0] * u[0] + 3.0f * c[1] * u[1] * u[1] + c[2] * u[0] * u[2] + c[5] * u[4]);
        Db[0][2] += g * (c[2] * u[0] * u[1] + c[4] * u[3]);
        Db[0][3] += g * (c[3] * u[0] * u[0] + c[4] * u[2]);
        Db[0][4] += g * (c[5] * u[1]);
        Db[0][5] += g * (c[6] * u[0]);
        Db[0][7] += g * c[7];
    }
};

template <> struct Residual<EQ_CH> {  // CH/PI-CH-1soliton.py:131-138; CH/bc-CH-1soliton.py:270-278 (c[0]=2)
    static constexpr int K = 3, KT = 3;
    template <class A, class R> __device__ static void eval(const A &D, const A &T, const float *c, int, R *f)
    {
        const R *u = D[0];
        f[0] = T[0][0] - T[0][2] + 3.0f * u[0] * u[1] - 2.0f * u[2] * u[1] + c[0] * u[1] - u[3] * u[0];
    }
    template <class A> __device__ static void adj(const A &D, const A &, const float *c, int,
                                                  const float *fb, A &Db, A &Tb)
    {
        const float *u = D[0];
        const float g = fb[0];
        Tb[0][0] += g;
        Tb[0][2] -= g;
        Db[0][0] += g * (3.0f * u[1] - u[3]);
        Db[0][1] += g * (3.0f * u[0] - 2.0f * u[2] + c[0]);
        Db[0][2] += g * (-2.0f * u[1]);
        Db[0][3] += g * (-u[0]);
    }
};

template <> struct Residual<EQ_GNSS> {  // NLS/sw-GNSS-adamW.py:159-180 (c[2]=0: NLS/sw-PINN-rogue.py:179-180)
    static constexpr int K = 3, KT = 1;
    template <class A, class R> __device__ static void eval(const A &D, const A &T, const float *c, int, R *f)
    {
        const R *u = D[0], *v = D[1];
        const R rho = u[0] * u[0] + v[0] * v[0];
        f[0] = T[0][0] + c[0] * v[2] + c[1] * rho * v[0] - c[2] * (u[3] + 6.0f * rho * u[1]);
        f[1] = T[1][0] - c[0] * u[2] - c[1] * rho * u[0] - c[2] * (v[3] + 6.0f * rho * v[1]);
    }
    template <class A> __device__ static void adj(const A &D, const A &, const float *c, int,
                                                  const float *fb, A &Db, A &Tb)
    {
        const float *u = D[0], *v = D[1];
        const float a = c[0], b = c[1], cc = c[2];
        const float rho = u[0] * u[0] + v[0] * v[0];
        const float g0 = fb[0], g1 = fb[1];
        Tb[0][0] += g0;
        Tb[1][0] += g1;
        Db[0][0] += g0 * (2.0f * b * u[0] * v[0] - 12.0f * cc * u[0] * u[1])
                  + g1 * (-b * (rho + 2.0f * u[0] * u[0]) - 12.0f * cc * u[0] * v[1]);
        Db[1][0] += g0 * (b * (rho + 2.0f * v[0] * v[0]) - 12.0f * cc * v[0] * u[1])
                  + g1 * (-2.0f * b * v[0] * u[0] - 12.0f * cc * v[0] * v[1]);
        Db[0][1] += g0 * (-6.0f * cc * rho);
        Db[1][1] += g1 * (-6.0f * cc * rho);
        Db[1][2] += g0 * a;
        Db[0][2] -= g1 * a;
        Db[0][3] -= g0 * cc;
        Db[1][3] -= g1 * cc;
    }
};

// Fourth-order (Lakshmanan-Porsezian-Daniel type) NLS of the scratch scripts NLS/Ibc-test.py:163-164 and
// NLS/testGNSS.py:139-140, as coded there (the 6 rho v_xx / 10 rho u_xx asymmetry included):
//   fu = u_t + a v_xx + b rho v + c4 Gv - c3 (u_xxx + 6 rho u_x)
//   fv = v_t - a u_xx - b rho u - c4 Gu - c3 (v_xxx + 6 rho v_x)
//   Gv = v_4x + 6v(v_x^2 - u_x^2) + 12 u u_x v_x + 4v(u_x^2 + v_x^2) + p rho v_xx + 6 rho^2 v
//   Gu = u_4x + 6u(u_x^2 - v_x^2) + 12 v u_x v_x + 4u(u_x^2 + v_x^2) + q rho u_xx + 6 rho^2 u
// coef = [a, b, c3, c4, p, q] = [0.5, 1, 0.125, 0.0625, 6, 10].
template <> struct Residual<EQ_LPD> {
    static constexpr int K = 4, KT = 1;
    template <class A, class R> __device__ static void eval(const A &D, const A &T, const float *c, int, R *f)
    {
        const R *u = D[0], *v = D[1];
        const float a = c[0], b = c[1], c3 = c[2], c4 = c[3], pp = c[4], qq = c[5];
        const R rho = u[0] * u[0] + v[0] * v[0];
        const R Gv = v[4] + 10.0f * v[0] * v[1] * v[1] - 2.0f * v[0] * u[1] * u[1] + 12.0f * u[0] * u[1] * v[1]
                       + pp * rho * v[2] + 6.0f * rho * rho * v[0];
        const R Gu = u[4] + 10.0f * u[0] * u[1] * u[1] - 2.0f * u[0] * v[1] * v[1] + 12.0f * v[0] * u[1] * v[1]
                       + qq * rho * u[2] + 6.0f * rho * rho * u[0];
        f[0] = T[0][0] + a * v[2] + b * rho * v[0] + c4 * Gv - c3 * (u[3] + 6.0f * rho * u[1]);
        f[1] = T[1][0] - a * u[2] - b * rho * u[0] - c4 * Gu - c3 * (v[3] + 6.0f * rho * v[1]);
    }
    template <class A> __device__ static void adj(const A &D, const A &, const float *c, int,
                                                  const float *fb, A &Db, A &Tb)
    {
        const float *u = D[0], *v = D[1];
        const float a = c[0], b = c[1], c3 = c[2], c4 = c[3], pp = c[4], qq = c[5];
        const float rho = u[0] * u[0] + v[0] * v[0];
        const float g0 = fb[0], g1 = fb[1];
        // partial derivatives of Gv and Gu
        const float Gv_u0 = 12.0f * u[1] * v[1] + 2.0f * pp * u[0] * v[2] + 24.0f * rho * u[0] * v[0];
        const float Gv_v0 = 10.0f * v[1] * v[1] - 2.0f * u[1] * u[1] + 2.0f * pp * v[0] * v[2] + 24.0f * rho * v[0] * v[0] + 6.0f * rho * rho;
        const float Gv_u1 = -4.0f * v[0] * u[1] + 12.0f * u[0] * v[1];
        const float Gv_v1 = 20.0f * v[0] * v[1] + 12.0f * u[0] * u[1];
        const float Gu_u0 = 10.0f * u[1] * u[1] - 2.0f * v[1] * v[1] + 2.0f * qq * u[0] * u[2] + 24.0f * rho * u[0] * u[0] + 6.0f * rho * rho;
        const float Gu_v0 = 12.0f * u[1] * v[1] + 2.0f * qq * v[0] * u[2] + 24.0f * rho * u[0] * v[0];
        const float Gu_u1 = 20.0f * u[0] * u[1] + 12.0f * v[0] * v[1];
        const float Gu_v1 = -4.0f * u[0] * v[1] + 12.0f * v[0] * u[1];
        Tb[0][0] += g0;
        Tb[1][0] += g1;
        Db[0][0] += g0 * (2.0f * b * u[0] * v[0] + c4 * Gv_u0 - 12.0f * c3 * u[0] * u[1])
                  + g1 * (-b * (rho + 2.0f * u[0] * u[0]) - c4 * Gu_u0 - 12.0f * c3 * u[0] * v[1]);
        Db[1][0] += g0 * (b * (rho + 2.0f * v[0] * v[0]) + c4 * Gv_v0 - 12.0f * c3 * v[0] * u[1])
                  + g1 * (-2.0f * b * v[0] * u[0] - c4 * Gu_v0 - 12.0f * c3 * v[0] * v[1]);
        Db[0][1] += g0 * (c4 * Gv_u1 - 6.0f * c3 * rho) + g1 * (-c4 * Gu_u1);
        Db[1][1] += g0 * (c4 * Gv_v1) + g1 * (-c4 * Gu_v1 - 6.0f * c3 * rho);
        Db[1][2] += g0 * (a + c4 * pp * rho);
        Db[0][2] += g1 * (-a - c4 * qq * rho);
        Db[0][3] -= g0 * c3;
        Db[1][3] -= g1 * c3;
        Db[1][4] += g0 * c4;
        Db[0][4] -= g1 * c4;
    }
};

// Plain NLS with the derivative regulariser of NLS/sw-test.py:159-189:
//   fu = u_t + a v_xx + b rho v,  fv = v_t - a u_xx - b rho u,
//   reg_term = lambda (mean u_x^2 + mean u_xx^2 + mean u_t^2 + mean v_x^2 + mean v_xx^2 + mean v_t^2),
// which the script adds to the loss (:147).  The regulariser rides as a THIRD residual row
//   f2 = sqrt(lambda (u_x^2 + u_xx^2 + u_t^2 + v_x^2 + v_xx^2 + v_t^2)),   mean(f2^2) = reg_term,
// so that sum_c mean(f_c^2) is loss_f + reg_term and the same fused kernel serves it.  coef = [a, b, lambda].
template <> struct Residual<EQ_NLSREG> {
    static constexpr int K = 2, KT = 1;
    template <class A, class R> __device__ static void eval(const A &D, const A &T, const float *c, int, R *f)
    {
        const R *u = D[0], *v = D[1];
        const R rho = u[0] * u[0] + v[0] * v[0];
        f[0] = T[0][0] + c[0] * v[2] + c[1] * rho * v[0];
        f[1] = T[1][0] - c[0] * u[2] - c[1] * rho * u[0];
        const R S = u[1] * u[1] + u[2] * u[2] + T[0][0] * T[0][0] + v[1] * v[1] + v[2] * v[2] + T[1][0] * T[1][0];
        f[2] = sqrt_r(c[2] * S);
    }
    template <class A> __device__ static void adj(const A &D, const A &T, const float *c, int,
                                                  const float *fb, A &Db, A &Tb)
    {
        const float *u = D[0], *v = D[1];
        const float rho = u[0] * u[0] + v[0] * v[0];
        const float g0 = fb[0], g1 = fb[1];
        Tb[0][0] += g0;
        Tb[1][0] += g1;
        Db[1][2] += g0 * c[0];
        Db[0][2] -= g1 * c[0];
        Db[0][0] += c[1] * (g0 * 2.0f * u[0] * v[0] - g1 * (rho + 2.0f * u[0] * u[0]));
        Db[1][0] += c[1] * (g0 * (rho + 2.0f * v[0] * v[0]) - g1 * 2.0f * u[0] * v[0]);
        // d f2 / d(derivative) = lambda * derivative / f2   (0 where every derivative vanishes)
        const float S = u[1] * u[1] + u[2] * u[2] + T[0][0] * T[0][0] + v[1] * v[1] + v[2] * v[2] + T[1][0] * T[1][0];
        const float f2 = sqrtf(c[2] * S);
        const float r = f2 > 0.0f ? fb[2] * c[2] / f2 : 0.0f;
        Db[0][1] += r * u[1];
        Db[0][2] += r * u[2];
        Tb[0][0] += r * T[0][0];
        Db[1][1] += r * v[1];
        Db[1][2] += r * v[2];
        Tb[1][0] += r * T[1][0];
    }
};

}  // namespace pinn
