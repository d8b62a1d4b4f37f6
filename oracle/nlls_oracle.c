/*
 * oracle/nlls_oracle.c -- TEST INFRASTRUCTURE: plain-C CPU restatement of the NLLSsolver.jl
 * Gauss-Newton / Levenberg-Marquardt inner loop.  Never linked into the product library.
 *
 * Citations are file:line under /root/reference (NLLSsolver.jl v4.0.3).  Parity status: PINNED by
 * the reference's RNG-free golden vectors (tests/test_oracle_pins.py); the Julia
 * reference itself cannot be executed here (no julia toolchain), so nothing is pinned against live
 * reference outputs.  Third-party arithmetic restated from published algorithms:
 *   - ForwardDiff (compat 0.10/1): dual numbers, oracle/jet.h
 *   - LDLFactorizations 0.10 (a Julia port of T. Davis' LDL, ACM TOMS Alg. 849): ldl_symbolic /
 *     ldl_numeric / solves below.  The fill-reducing ordering (AMD in the reference) is replaced by
 *     a block minimum-degree ordering; the solution x of a nonsingular system does not depend on it.
 *   - LAPACK potrf / geqrf (src/linearsolver.jl:21-25): unblocked Cholesky / Householder QR.
 * Two NEW kinds with no reference counterpart ("parity unpinned" for them):
 *   NLLS_VAR_POSE_SO3 and NLLS_RES_BA_SO3(_ADAPTIVE).
 */
#include "nlls_oracle.h"
#include "jet.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

/* ============================================================================================ */
/* kind tables                                                                                  */
/* ============================================================================================ */
typedef struct { int ndeps, nres, ndata, adaptive; int slot_kind[4]; int slot_dim[4]; } res_desc;
static const res_desc RES[NLLS_RES_KIND_COUNT] = {
    {0, 0, 0, 0, {0}, {0}},
    /* BA_AFFINE       */ {2, 2, 2, 0, {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN}, {6, 3}},
    /* ROSENBROCK_A    */ {1, 1, 1, 0, {NLLS_VAR_EUCLIDEAN}, {1}},
    /* ROSENBROCK_B    */ {2, 1, 1, 0, {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN}, {1, 1}},
    /* ROSENBROCK_2D   */ {1, 2, 2, 0, {NLLS_VAR_EUCLIDEAN}, {2}},
    /* CURVE_EXP4      */ {4, 1, 2, 0, {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN}, {1, 1, 1, 1}},
    /* ADAPTIVE_MEAN   */ {2, 1, 1, 1, {NLLS_VAR_CONTAMINATED_GAUSSIAN, NLLS_VAR_EUCLIDEAN}, {3, 1}},
    /* BA_SO3          */ {2, 2, 2, 0, {NLLS_VAR_POSE_SO3, NLLS_VAR_EUCLIDEAN}, {6, 3}},
    /* BA_SO3_ADAPTIVE */ {3, 2, 2, 1, {NLLS_VAR_CONTAMINATED_GAUSSIAN, NLLS_VAR_POSE_SO3, NLLS_VAR_EUCLIDEAN}, {3, 6, 3}},
    /* LINEAR3         */ {1, 3, 12, 0, {NLLS_VAR_EUCLIDEAN}, {3}},
    /* COST_LINEAR3    */ {1, 0, 3, 0, {NLLS_VAR_EUCLIDEAN}, {3}},   /* nres = 0: an AbstractCost, not a residual */
    /* DYN_LINEAR      */ {1, 1, -1, 0, {NLLS_VAR_DYNAMIC}, {0}},    /* ndata = 1 + n: see ogroup.ndata */
    /* DYN_NORM        */ {1, -1, 0, 0, {NLLS_VAR_DYNAMIC}, {0}},    /* nres = n */
    /* DYN_LINEARSQ    */ {1, -1, -1, 0, {NLLS_VAR_DYNAMIC}, {0}},   /* nres = n, ndata = n + n*n */
    /* COST_DYN_LINEAR */ {1, 0, -1, 0, {NLLS_VAR_DYNAMIC}, {0}},    /* an AbstractCost; ndata = n */
    /* SCALE_MIX       */ {2, 1, 3, 0, {NLLS_VAR_ZERO_TO_INF, NLLS_VAR_ZERO_TO_ONE}, {1, 1}},
};
#define IS_DYN_KIND(k) ((k) >= NLLS_RES_DYN_LINEAR && (k) <= NLLS_COST_DYN_LINEAR)
#define IS_COST_KIND(k) ((k) == NLLS_COST_LINEAR3)

static int var_storage(int kind, int dim) {
    switch (kind) {
    case NLLS_VAR_EUCLIDEAN: case NLLS_VAR_DYNAMIC: return dim;
    case NLLS_VAR_ZERO_TO_INF: case NLLS_VAR_ZERO_TO_ONE: return 1;
    case NLLS_VAR_CONTAMINATED_GAUSSIAN: return 3;
    case NLLS_VAR_POSE_SO3: return 12;
    }
    return -1;
}
static int var_dof(int kind, int dim) { /* nvars(): src/variable.jl:4,9,14,21,28; robustadaptive.jl:21 */
    switch (kind) {
    case NLLS_VAR_EUCLIDEAN: case NLLS_VAR_DYNAMIC: return dim;
    case NLLS_VAR_ZERO_TO_INF: case NLLS_VAR_ZERO_TO_ONE: return 1;
    case NLLS_VAR_CONTAMINATED_GAUSSIAN: return 3;
    case NLLS_VAR_POSE_SO3: return 6;
    }
    return -1;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* ============================================================================================ */
/* variables: update()  src/variable.jl:5,10,22,29-32; src/robustadaptive.jl:12-22                */
/* ============================================================================================ */
static double zti_update(double v, double d) { return (v > 0 ? v : DBL_MIN) * exp(d); } /* variable.jl:22 */
static double zto_update(double v, double d) {                                          /* variable.jl:29-32 */
    double val = (v > 0 ? v : DBL_MIN) * exp(d);
    return val < INFINITY ? val / (1 + (val - v)) : 1.0;
}
static void so3_exp(const double w[3], double E[9]) { /* Rodrigues, col-major; NEW kind */
    double th2 = w[0]*w[0] + w[1]*w[1] + w[2]*w[2], A, B;
    if (th2 < 1e-12) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; }
    else { double th = sqrt(th2); A = sin(th) / th; B = (1.0 - cos(th)) / th2; }
    double K[9] = {0, w[2], -w[1], -w[2], 0, w[0], w[1], -w[0], 0}; /* col-major [w]x */
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) {
        double k2 = 0; for (int k = 0; k < 3; ++k) k2 += K[r + 3*k] * K[k + 3*c];
        E[r + 3*c] = (r == c) + A * K[r + 3*c] + B * k2;
    }
}
void oracle_contaminated_gaussian(double s1, double s2, double w, double st[3]) { /* robustadaptive.jl:12-20 */
    double a = 1.0 / s1, b = 1.0 / s2;
    if (!(a >= b)) { double t = a; a = b; b = t; }
    st[0] = a; st[1] = b; st[2] = w;
}
void oracle_var_update(int32_t kind, int32_t dim, const double* in, const double* d, double* out) {
    switch (kind) {
    case NLLS_VAR_EUCLIDEAN: case NLLS_VAR_DYNAMIC: for (int i = 0; i < dim; ++i) out[i] = in[i] + d[i]; break;
    case NLLS_VAR_ZERO_TO_INF: out[0] = zti_update(in[0], d[0]); break;
    case NLLS_VAR_ZERO_TO_ONE: out[0] = zto_update(in[0], d[0]); break;
    case NLLS_VAR_CONTAMINATED_GAUSSIAN: { /* robustadaptive.jl:22 then the ordering of :13-15 */
        double a = zti_update(in[0], d[0]), b = zti_update(in[1], d[1]), w = zto_update(in[2], d[2]);
        if (!(a >= b)) { double t = a; a = b; b = t; }
        out[0] = a; out[1] = b; out[2] = w; break; }
    case NLLS_VAR_POSE_SO3: {
        double E[9]; so3_exp(d, E);
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) {
            double s = 0; for (int k = 0; k < 3; ++k) s += in[r + 3*k] * E[k + 3*c];
            out[r + 3*c] = s;
        }
        for (int i = 0; i < 3; ++i) out[9 + i] = in[9 + i] + d[3 + i];
        break; }
    }
}
/* update(var, dualzeros): storage as jets; start < 0 => fixed variable (plain values)
 * src/autodiff.jl:57-61 */
static void var_update_jet(int kind, int dim, const double* v, int start, int n, jet* out) {
    int st = var_storage(kind, dim);
    if (start < 0) { for (int i = 0; i < st; ++i) out[i] = jet_const(v[i], n); return; }
    switch (kind) {
    case NLLS_VAR_EUCLIDEAN: for (int i = 0; i < dim; ++i) out[i] = jet_seed(v[i], start + i, n); break;
    case NLLS_VAR_ZERO_TO_INF: { double b = v[0] > 0 ? v[0] : DBL_MIN; out[0] = jet_const(b, n); out[0].d[start] = b; break; }
    case NLLS_VAR_ZERO_TO_ONE: {
        double b = v[0] > 0 ? v[0] : DBL_MIN; jet val = jet_const(b, n); val.d[start] = b;
        jet den = jet_addc(val, 1.0 - v[0], n);
        out[0] = jet_div(val, den, n); break; }
    case NLLS_VAR_CONTAMINATED_GAUSSIAN:
        var_update_jet(NLLS_VAR_ZERO_TO_INF, 1, v + 0, start + 0, n, out + 0);
        var_update_jet(NLLS_VAR_ZERO_TO_INF, 1, v + 1, start + 1, n, out + 1);
        var_update_jet(NLLS_VAR_ZERO_TO_ONE, 1, v + 2, start + 2, n, out + 2);
        break;
    case NLLS_VAR_POSE_SO3: {
        /* R*(I + [d]x): exact first-order behaviour of R*expm([d]x) at d = 0 */
        for (int r = 0; r < 3; ++r) {
            double R0 = v[r], R1 = v[r + 3], R2 = v[r + 6];
            jet c0 = jet_const(R0, n), c1 = jet_const(R1, n), c2 = jet_const(R2, n);
            /* col0 = R0 + R1*d2 - R2*d1 ; col1 = R1 - R0*d2 + R2*d0 ; col2 = R2 + R0*d1 - R1*d0 */
            c0.d[start + 2] += R1; c0.d[start + 1] -= R2;
            c1.d[start + 2] -= R0; c1.d[start + 0] += R2;
            c2.d[start + 1] += R0; c2.d[start + 0] -= R1;
            out[r] = c0; out[r + 3] = c1; out[r + 6] = c2;
        }
        for (int i = 0; i < 3; ++i) out[9 + i] = jet_seed(v[9 + i], start + 3 + i, n);
        break; }
    }
}

/* ============================================================================================ */
/* residuals: computeresidual()                                                                 */
/* ============================================================================================ */
static void res_eval(int kind, const double* data, jet* const* sv, jet* r, int n) {
    switch (kind) {
    case NLLS_RES_BA_AFFINE: { /* test/optimizeba.jl:4 + src/residual.jl:13 */
        const jet *c = sv[0], *X = sv[1];
        jet u = jet_add(jet_add(jet_mul(c[0], X[0], n), jet_mul(c[1], X[1], n), n), jet_mul(c[2], X[2], n), n);
        jet w = jet_add(jet_add(jet_mul(c[3], X[0], n), jet_mul(c[4], X[1], n), n), jet_mul(c[5], X[2], n), n);
        r[0] = jet_addc(u, -data[0], n); r[1] = jet_addc(w, -data[1], n); break; }
    case NLLS_RES_ROSENBROCK_A: /* test/functional.jl:12: a*(1-x) */
        r[0] = jet_scale(jet_addc(jet_neg(sv[0][0], n), 1.0, n), data[0], n); break;
    case NLLS_RES_ROSENBROCK_B: /* test/functional.jl:24: b*(x^2-y) */
        r[0] = jet_scale(jet_sub(jet_mul(sv[0][0], sv[0][0], n), sv[1][0], n), data[0], n); break;
    case NLLS_RES_ROSENBROCK_2D: { /* examples/rosenbrock.jl:19 */
        const jet* x = sv[0];
        r[0] = jet_scale(jet_addc(jet_neg(x[0], n), 1.0, n), data[0], n);
        r[1] = jet_scale(jet_sub(jet_mul(x[0], x[0], n), x[1], n), data[1], n); break; }
    case NLLS_RES_CURVE_EXP4: { /* BASELINE config 2: a*exp(b*t) + c*t + d - y */
        double t = data[0], y = data[1];
        jet e = jet_exp(jet_scale(sv[1][0], t, n), n);
        jet m = jet_add(jet_add(jet_mul(sv[0][0], e, n), jet_scale(sv[2][0], t, n), n), sv[3][0], n);
        r[0] = jet_addc(m, -y, n); break; }
    case NLLS_RES_ADAPTIVE_MEAN: /* test/adaptivecost.jl:11: mean - data (sv excludes the kernel) */
        r[0] = jet_addc(sv[0][0], -data[0], n); break;
    case NLLS_RES_BA_SO3: case NLLS_RES_BA_SO3_ADAPTIVE: { /* NEW: pinhole, normalised image plane */
        const jet *P = sv[0], *X = sv[1]; jet Y[3];
        for (int i = 0; i < 3; ++i)
            Y[i] = jet_add(jet_add(jet_add(jet_mul(P[i], X[0], n), jet_mul(P[i + 3], X[1], n), n), jet_mul(P[i + 6], X[2], n), n), P[9 + i], n);
        r[0] = jet_addc(jet_div(Y[0], Y[2], n), -data[0], n);
        r[1] = jet_addc(jet_div(Y[1], Y[2], n), -data[1], n); break; }
    case NLLS_RES_SCALE_MIX: { /* s * (w a + (1 - w) b) - y over a ZeroToInfScalar and a ZeroToOneScalar (src/variable.jl:18-32) */
        jet s = sv[0][0], w = sv[1][0];
        jet mix = jet_add(jet_scale(w, data[0], n), jet_scale(jet_addc(jet_neg(w, n), 1.0, n), data[1], n), n);
        r[0] = jet_addc(jet_mul(s, mix, n), -data[2], n); break; }
    case NLLS_RES_LINEAR3: { /* test/nonsquaredcost.jl:13: X * w - y, data = (y[3], X[9] column-major) */
        const jet* w = sv[0];
        for (int i = 0; i < 3; ++i)
            r[i] = jet_addc(jet_add(jet_add(jet_scale(w[0], data[3 + i], n), jet_scale(w[1], data[6 + i], n), n), jet_scale(w[2], data[9 + i], n), n), -data[i], n);
        break; }
    }
}
/* computecost() of the AbstractCost kinds with second-order jets (computehessian, src/autodiff.jl:123-128,144-159); the
 * variables are Euclidean (linear retraction), seeded as jet2 variable `start + i` */
static jet2 cost_eval2(int kind, const double* data, const double* const* st, const int* start) {
    switch (kind) {
    case NLLS_COST_LINEAR3: { /* test/nonsquaredcost.jl:36: y' * w */
        jet2 c = j2_const(0.0);
        for (int i = 0; i < 3; ++i) { jet2 w = start[0] >= 0 ? j2_seed(st[0][i], start[0] + i) : j2_const(st[0][i]); c = j2_add(c, j2_scale(w, data[i])); }
        return c; }
    }
    return j2_const(0.0);
}

/* ============================================================================================ */
/* robust kernels  src/robust.jl, src/robustadaptive.jl                                          */
/* ============================================================================================ */
typedef struct { double is1, is2, w, s1sq, s2sq, halfs2sqminuss1sq, halfs2sq; } cgauss;
static cgauss cg_make(const double st[3]) { /* robustadaptive.jl:12-20 (ordering already applied to storage) */
    cgauss k; k.is1 = st[0]; k.is2 = st[1]; k.w = st[2];
    k.s1sq = k.is1 * k.is1; k.s2sq = k.is2 * k.is2;
    k.halfs2sqminuss1sq = 0.5 * (k.s2sq - k.s1sq); k.halfs2sq = 0.5 * k.s2sq; return k;
}
static double cg_robustify(const cgauss* k, double cost) { /* robustadaptive.jl:25 */
    return cost * k->halfs2sq - log(k->w * k->is1 * exp(cost * k->halfs2sqminuss1sq) + (1 - k->w) * k->is2);
}
static void cg_robustifydcost(const cgauss* k, double cost, double out[3]) { /* robustadaptive.jl:26-33 */
    double c = cost * k->halfs2sq;
    double s = k->w * k->is1 * exp(cost * k->halfs2sqminuss1sq);
    double t = (1 - k->w) * k->is2;
    double den = 1 / (s + t);
    s *= k->halfs2sqminuss1sq;
    out[0] = c + log(den); out[1] = k->halfs2sq - s * den; out[2] = -s * k->halfs2sqminuss1sq * t * den * den;
}
static double fixed_robustify(int rk, const double* p, double cost) {
    int base = rk & 0xF; double c;
    switch (base) {
    case NLLS_ROBUST_HUBER: case NLLS_ROBUST_HUBER2O: { /* robust.jl:47 */
        double w = p[0], w2 = w * w; c = cost < w2 ? cost : sqrt(cost) * (w * 2) - w2; break; }
    case NLLS_ROBUST_GEMAN_MCCLURE: { double w2 = p[0] * p[0]; c = cost * w2 / (cost + w2); break; } /* robust.jl:71 */
    default: c = cost; /* robust.jl:11 */
    }
    if (rk & NLLS_ROBUST_SCALED) c *= p[1]; /* robust.jl:26 */
    return c;
}
static void fixed_robustifydcost(int rk, const double* p, double cost, double out[3]) {
    int base = rk & 0xF;
    switch (base) {
    case NLLS_ROBUST_HUBER: case NLLS_ROBUST_HUBER2O: { /* robust.jl:48-55 */
        double w = p[0], w2 = w * w;
        if (cost < w2) { out[0] = cost; out[1] = 1; out[2] = 0; }
        else { double sq = sqrt(cost);
               out[0] = sq * (w * 2) - w2; out[1] = w / sq;
               out[2] = base == NLLS_ROBUST_HUBER2O ? (-0.5 * w) / (cost * sq) : 0.0; }
        break; }
    case NLLS_ROBUST_GEMAN_MCCLURE: { /* robust.jl:72-77 */
        double w2 = p[0] * p[0], r = 1.0 / (cost + w2), w = w2 * r, ww = w * w;
        out[0] = cost * w; out[1] = ww; out[2] = -2 * ww * r; break; }
    default: out[0] = cost; out[1] = 1; out[2] = 0; /* robust.jl:12 */
    }
    if (rk & NLLS_ROBUST_SCALED) { out[0] *= p[1]; out[1] *= p[1]; out[2] *= p[1]; } /* robust.jl:28-31 */
}
double oracle_robustify(int32_t rk, const double* kp, double cost) {
    if (rk < 0) { cgauss k = cg_make(kp); return cg_robustify(&k, cost); }
    return fixed_robustify(rk, kp, cost);
}
void oracle_robustifydcost(int32_t rk, const double* kp, double cost, double out[3]) {
    if (rk < 0) { cgauss k = cg_make(kp); cg_robustifydcost(&k, cost, out); return; }
    fixed_robustifydcost(rk, kp, cost, out);
}
/* robustify evaluated on second-order jets (generic, used for the AD cross-check and dkernel) */
static jet2 j2_robustify_fixed(int rk, const double* p, jet2 cost) {
    int base = rk & 0xF; jet2 c;
    switch (base) {
    case NLLS_ROBUST_HUBER: case NLLS_ROBUST_HUBER2O: {
        double w = p[0], w2 = w * w;
        c = cost.v < w2 ? cost : j2_addc(j2_scale(j2_sqrt(cost), w * 2), -w2); break; }
    case NLLS_ROBUST_GEMAN_MCCLURE: { double w2 = p[0] * p[0]; c = j2_div(j2_scale(cost, w2), j2_addc(cost, w2)); break; }
    default: c = cost;
    }
    if (rk & NLLS_ROBUST_SCALED) c = j2_scale(c, p[1]);
    return c;
}
static jet2 j2_robustify_cg(jet2 is1, jet2 is2, jet2 w, jet2 cost) { /* robustadaptive.jl:12-25 on duals (no swap :13) */
    jet2 s1sq = j2_mul(is1, is1), s2sq = j2_mul(is2, is2);
    jet2 hdiff = j2_scale(j2_sub(s2sq, s1sq), 0.5), hs2 = j2_scale(s2sq, 0.5);
    jet2 a = j2_mul(j2_mul(w, is1), j2_exp(j2_mul(cost, hdiff)));
    jet2 b = j2_mul(j2_addc(j2_scale(w, -1.0), 1.0), is2);
    return j2_sub(j2_mul(cost, hs2), j2_log(j2_add(a, b)));
}
void oracle_autorobustifydcost(int32_t rk, const double* kp, double cost, double out[3]) { /* autodiff.jl:163 */
    jet2 c = j2_seed(cost, 3), r;
    if (rk < 0) r = j2_robustify_cg(j2_const(kp[0]), j2_const(kp[1]), j2_const(kp[2]), c);
    else r = j2_robustify_fixed(rk, kp, c);
    out[0] = r.v; out[1] = r.g[3]; out[2] = r.h[3][3];
}
void oracle_robustifydkernel(const double* st, double cost, double* value, double grad[4], double hess[16]) {
    /* autodiff.jl:164-165: hessian of x -> robustify(update(kernel, x), cost + x[end]) at x = 0 */
    double b0 = st[0] > 0 ? st[0] : DBL_MIN, b1 = st[1] > 0 ? st[1] : DBL_MIN, b2 = st[2] > 0 ? st[2] : DBL_MIN;
    /* ZeroToInf: b*exp(x)  (variable.jl:22) -> value b, d/dx = b, d2/dx2 = b */
    jet2 is1 = j2_const(b0); is1.g[0] = b0; is1.h[0][0] = b0;
    jet2 is2 = j2_const(b1); is2.g[1] = b1; is2.h[1][1] = b1;
    /* ZeroToOne (variable.jl:29-32): val = b*exp(x); val/(1+(val-v)) */
    jet2 val = j2_const(b2); val.g[2] = b2; val.h[2][2] = b2;
    jet2 w = j2_div(val, j2_addc(val, 1.0 - st[2]));
    jet2 c = j2_seed(cost, 3);
    jet2 r = j2_robustify_cg(is1, is2, w, c);
    *value = r.v;
    for (int i = 0; i < 4; ++i) { grad[i] = r.g[i]; for (int j = 0; j < 4; ++j) hess[i + 4 * j] = r.h[i][j]; }
}

/* ============================================================================================ */
/* utils  src/utils.jl                                                                          */
/* ============================================================================================ */
int64_t oracle_runlengthencodesortedints(const int64_t* s, int64_t n, int64_t* out) { /* utils.jl:38-52 */
    int64_t ind = 1, currval = 1; /* 1-based like Julia: runindices[currval] */
    out[currval - 1] = ind;
    for (int64_t k = 0; k < n; ++k) {
        int64_t val = s[k];
        while (val >= currval) { currval += 1; out[currval - 1] = ind; }
        ind += 1;
    }
    out[currval] = ind;
    return s[n - 1] + 2;
}
double oracle_fast_bAb_dense(const double* A, const double* b, int64_t n) { /* utils.jl:71-81 */
    double total = 0;
    for (int64_t i = 0; i < n; ++i) { double sub = 0; for (int64_t j = 0; j < n; ++j) sub += A[j + n * i] * b[j]; total += b[i] * sub; }
    return total;
}
double oracle_fast_bAb_csc(const int64_t* cp, const int64_t* rv, const double* nz, const double* b, int64_t n) { /* utils.jl:95-106 */
    double total = 0;
    for (int64_t i = 0; i < n; ++i) { double c = 0; for (int64_t j = cp[i] - 1; j < cp[i + 1] - 1; ++j) c += nz[j] * b[rv[j] - 1]; total += c * b[i]; }
    return total;
}
/* Julia's pairwise mapreduce (Base.mapreduce_impl, blksize 1024) used by sum(f, vector)
 * (src/VectorRepo.jl:64-69, src/cost.jl:11,54) */
typedef double (*sumfun)(void* ctx, int64_t i);
static double pairwise_sum(sumfun f, void* ctx, int64_t lo, int64_t hi) { /* [lo, hi) */
    if (hi - lo < 1024) { double s = 0; for (int64_t i = lo; i < hi; ++i) s += f(ctx, i); return s; }
    int64_t mid = lo + ((hi - lo) >> 1);
    double a = pairwise_sum(f, ctx, lo, mid); return a + pairwise_sum(f, ctx, mid, hi);
}

/* ============================================================================================ */
/* dense solves  src/linearsolver.jl:20-26                                                       */
/* ============================================================================================ */
static int dense_cholesky(double* L, int64_t n) { /* lower, in place, col-major; 0 ok */
    for (int64_t j = 0; j < n; ++j) {
        double d = L[j + n * j];
        for (int64_t k = 0; k < j; ++k) d -= L[j + n * k] * L[j + n * k];
        if (!(d > 0)) return -1;
        d = sqrt(d); L[j + n * j] = d;
        for (int64_t i = j + 1; i < n; ++i) {
            double s = L[i + n * j];
            for (int64_t k = 0; k < j; ++k) s -= L[i + n * k] * L[j + n * k];
            L[i + n * j] = s / d;
        }
    }
    return 0;
}
static void dense_chol_solve(const double* L, double* x, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { double s = x[i]; for (int64_t k = 0; k < i; ++k) s -= L[i + n * k] * x[k]; x[i] = s / L[i + n * i]; }
    for (int64_t i = n - 1; i >= 0; --i) { double s = x[i]; for (int64_t k = i + 1; k < n; ++k) s -= L[k + n * i] * x[k]; x[i] = s / L[i + n * i]; }
}
static void dense_qr_solve(double* A, double* x, int64_t n) { /* Householder QR, A col-major overwritten, x = rhs -> solution */
    for (int64_t k = 0; k < n; ++k) {
        double nrm = 0; for (int64_t i = k; i < n; ++i) nrm += A[i + n * k] * A[i + n * k];
        nrm = sqrt(nrm); if (nrm == 0) continue;
        double alpha = A[k + n * k] > 0 ? -nrm : nrm;
        double v0 = A[k + n * k] - alpha; A[k + n * k] = alpha;
        double vn = v0 * v0; for (int64_t i = k + 1; i < n; ++i) vn += A[i + n * k] * A[i + n * k];
        if (vn == 0) continue;
        for (int64_t j = k + 1; j <= n; ++j) { /* column n == rhs */
            double* col = j < n ? A + n * j : x;
            double dot = v0 * col[k]; for (int64_t i = k + 1; i < n; ++i) dot += A[i + n * k] * col[i];
            double f = 2 * dot / vn; col[k] -= f * v0; for (int64_t i = k + 1; i < n; ++i) col[i] -= f * A[i + n * k];
        }
    }
    for (int64_t i = n - 1; i >= 0; --i) { double s = x[i]; for (int64_t k = i + 1; k < n; ++k) s -= A[i + n * k] * x[k]; x[i] = s / A[i + n * i]; }
}
int oracle_solve_dense(double* x, const double* A, const double* b, int64_t n) {
    double* W = (double*)malloc(sizeof(double) * n * n);
    memcpy(W, A, sizeof(double) * n * n); memcpy(x, b, sizeof(double) * n);
    int how = 0, herm = 1;
    /* LinearAlgebra.cholesky(A; check=false) reports failure for a non-Hermitian A (ishermitian test) */
    for (int64_t j = 0; j < n && herm; ++j) for (int64_t i = 0; i < j; ++i) if (A[i + n * j] != A[j + n * i]) { herm = 0; break; }
    if (herm && dense_cholesky(W, n) == 0) dense_chol_solve(W, x, n);
    else { memcpy(W, A, sizeof(double) * n * n); memcpy(x, b, sizeof(double) * n); dense_qr_solve(W, x, n); how = 1; }
    free(W); return how;
}

/* ============================================================================================ */
/* sparse LDL' (T. Davis, Alg. 849) -- what LDLFactorizations.jl ports  (src/linearsolver.jl:29,32)*/
/* ============================================================================================ */
typedef struct {
    int64_t n; int64_t *P, *Pinv, *Parent, *Lp, *Lnz, *Li, *Flag, *Pattern; double *Lx, *D, *Y;
} ldl_fac;
static void ldl_free(ldl_fac* f) {
    free(f->P); free(f->Pinv); free(f->Parent); free(f->Lp); free(f->Lnz); free(f->Li); free(f->Flag); free(f->Pattern);
    free(f->Lx); free(f->D); free(f->Y); memset(f, 0, sizeof *f);
}
/* Ap/Ai 0-based, any symmetric storage (only entries with permuted row <= col are used) */
static void ldl_symbolic(ldl_fac* f, int64_t n, const int64_t* Ap, const int64_t* Ai, const int64_t* P) {
    f->n = n;
    f->P = (int64_t*)malloc(sizeof(int64_t) * n); f->Pinv = (int64_t*)malloc(sizeof(int64_t) * n);
    f->Parent = (int64_t*)malloc(sizeof(int64_t) * n); f->Lp = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
    f->Lnz = (int64_t*)malloc(sizeof(int64_t) * n); f->Flag = (int64_t*)malloc(sizeof(int64_t) * n);
    f->Pattern = (int64_t*)malloc(sizeof(int64_t) * n);
    f->D = (double*)malloc(sizeof(double) * n); f->Y = (double*)malloc(sizeof(double) * n);
    for (int64_t k = 0; k < n; ++k) { f->P[k] = P ? P[k] : k; }
    for (int64_t k = 0; k < n; ++k) f->Pinv[f->P[k]] = k;
    for (int64_t k = 0; k < n; ++k) {
        f->Parent[k] = -1; f->Flag[k] = k; f->Lnz[k] = 0;
        int64_t kk = f->P[k];
        for (int64_t p = Ap[kk]; p < Ap[kk + 1]; ++p) {
            int64_t i = f->Pinv[Ai[p]];
            if (i < k) for (; f->Flag[i] != k; i = f->Parent[i]) {
                if (f->Parent[i] == -1) f->Parent[i] = k;
                f->Lnz[i]++; f->Flag[i] = k;
            }
        }
    }
    f->Lp[0] = 0; for (int64_t k = 0; k < n; ++k) f->Lp[k + 1] = f->Lp[k] + f->Lnz[k];
    f->Li = (int64_t*)malloc(sizeof(int64_t) * (f->Lp[n] > 0 ? f->Lp[n] : 1));
    f->Lx = (double*)malloc(sizeof(double) * (f->Lp[n] > 0 ? f->Lp[n] : 1));
}
static int64_t ldl_numeric(ldl_fac* f, const int64_t* Ap, const int64_t* Ai, const double* Ax) {
    int64_t n = f->n;
    for (int64_t k = 0; k < n; ++k) {
        f->Y[k] = 0; int64_t top = n; f->Flag[k] = k; f->Lnz[k] = 0;
        int64_t kk = f->P[k];
        for (int64_t p = Ap[kk]; p < Ap[kk + 1]; ++p) {
            int64_t i = f->Pinv[Ai[p]];
            if (i <= k) {
                f->Y[i] += Ax[p]; int64_t len;
                for (len = 0; f->Flag[i] != k; i = f->Parent[i]) { f->Pattern[len++] = i; f->Flag[i] = k; }
                while (len > 0) f->Pattern[--top] = f->Pattern[--len];
            }
        }
        f->D[k] = f->Y[k]; f->Y[k] = 0;
        for (; top < n; ++top) {
            int64_t i = f->Pattern[top]; double yi = f->Y[i]; f->Y[i] = 0;
            int64_t p, p2 = f->Lp[i] + f->Lnz[i];
            for (p = f->Lp[i]; p < p2; ++p) f->Y[f->Li[p]] -= f->Lx[p] * yi;
            double lki = yi / f->D[i]; f->D[k] -= lki * yi;
            f->Li[p] = k; f->Lx[p] = lki; f->Lnz[i]++;
        }
        if (f->D[k] == 0) return k;
    }
    return n;
}
static void ldl_solve(const ldl_fac* f, const double* b, double* x) {
    int64_t n = f->n; double* y = (double*)malloc(sizeof(double) * n);
    for (int64_t k = 0; k < n; ++k) y[k] = b[f->P[k]];
    for (int64_t j = 0; j < n; ++j) { int64_t p2 = f->Lp[j] + f->Lnz[j]; for (int64_t p = f->Lp[j]; p < p2; ++p) y[f->Li[p]] -= f->Lx[p] * y[j]; }
    for (int64_t j = 0; j < n; ++j) y[j] /= f->D[j];
    for (int64_t j = n - 1; j >= 0; --j) { int64_t p2 = f->Lp[j] + f->Lnz[j]; for (int64_t p = f->Lp[j]; p < p2; ++p) y[j] -= f->Lx[p] * y[f->Li[p]]; }
    for (int64_t k = 0; k < n; ++k) x[f->P[k]] = y[k];
    free(y);
}
/* minimum-degree ordering on a graph given as symmetric adjacency (CSC, 0-based, diag ignored) with
 * node weights; stands in for AMD (ordering does not change x). */
typedef struct { int64_t *a; int64_t n, cap; } ivec;
static void iv_push(ivec* v, int64_t x) { if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 8; v->a = (int64_t*)realloc(v->a, sizeof(int64_t) * v->cap); } v->a[v->n++] = x; }
static int cmp_i64(const void* a, const void* b) { int64_t x = *(const int64_t*)a, y = *(const int64_t*)b; return (x > y) - (x < y); }
static void min_degree_order(int64_t n, const int64_t* Ap, const int64_t* Ai, int64_t* perm) {
    ivec* adj = (ivec*)calloc(n, sizeof(ivec));
    char* done = (char*)calloc(n, 1);
    for (int64_t j = 0; j < n; ++j) for (int64_t p = Ap[j]; p < Ap[j + 1]; ++p) if (Ai[p] != j) iv_push(&adj[j], Ai[p]);
    /* lazy binary heap of (degree, node) */
    int64_t hcap = 4 * n + 16, hn = 0; int64_t* hk = (int64_t*)malloc(sizeof(int64_t) * hcap * 2);
#define HPUSH(deg, node) do { if (hn == hcap) { hcap *= 2; hk = (int64_t*)realloc(hk, sizeof(int64_t) * hcap * 2); } \
        int64_t c_ = hn++; hk[2*c_] = (deg); hk[2*c_+1] = (node); \
        while (c_ > 0) { int64_t p_ = (c_ - 1) / 2; if (hk[2*p_] < hk[2*c_] || (hk[2*p_] == hk[2*c_] && hk[2*p_+1] <= hk[2*c_+1])) break; \
            int64_t t0 = hk[2*p_], t1 = hk[2*p_+1]; hk[2*p_] = hk[2*c_]; hk[2*p_+1] = hk[2*c_+1]; hk[2*c_] = t0; hk[2*c_+1] = t1; c_ = p_; } } while (0)
    for (int64_t j = 0; j < n; ++j) HPUSH(adj[j].n, j);
    int64_t* mark = (int64_t*)malloc(sizeof(int64_t) * n); for (int64_t j = 0; j < n; ++j) mark[j] = -1;
    int64_t out = 0;
    while (out < n) {
        /* pop min */
        int64_t deg = hk[0], v = hk[1]; hn--;
        if (hn > 0) { hk[0] = hk[2*hn]; hk[1] = hk[2*hn+1]; int64_t c = 0;
            for (;;) { int64_t l = 2*c+1, r = l+1, m = c;
                if (l < hn && (hk[2*l] < hk[2*m] || (hk[2*l] == hk[2*m] && hk[2*l+1] < hk[2*m+1]))) m = l;
                if (r < hn && (hk[2*r] < hk[2*m] || (hk[2*r] == hk[2*m] && hk[2*r+1] < hk[2*m+1]))) m = r;
                if (m == c) break;
                int64_t t0 = hk[2*m], t1 = hk[2*m+1]; hk[2*m] = hk[2*c]; hk[2*m+1] = hk[2*c+1]; hk[2*c] = t0; hk[2*c+1] = t1; c = m; } }
        if (done[v] || deg != adj[v].n) continue; /* stale */
        done[v] = 1; perm[out++] = v;
        /* neighbours become a clique */
        ivec nb = adj[v];
        for (int64_t a = 0; a < nb.n; ++a) {
            int64_t u = nb.a[a];
            /* rebuild adj[u] = (adj[u] U nb) \ {u, v, done} */
            ivec nu = {0, 0, 0};
            for (int64_t k = 0; k < adj[u].n; ++k) { int64_t w = adj[u].a[k]; if (w != v && !done[w] && mark[w] != u) { mark[w] = u; iv_push(&nu, w); } }
            for (int64_t k = 0; k < nb.n; ++k) { int64_t w = nb.a[k]; if (w != u && !done[w] && mark[w] != u) { mark[w] = u; iv_push(&nu, w); } }
            for (int64_t k = 0; k < nu.n; ++k) mark[nu.a[k]] = -1;
            free(adj[u].a); adj[u] = nu; HPUSH(nu.n, u);
        }
        free(nb.a); adj[v].a = 0; adj[v].n = 0;
    }
#undef HPUSH
    for (int64_t j = 0; j < n; ++j) free(adj[j].a);
    free(adj); free(done); free(hk); free(mark);
}
int oracle_solve_sparse(double* x, const int64_t* cp1, const int64_t* rv1, const double* nz, const double* b, int64_t n) {
    int64_t nnz = cp1[n] - 1; int64_t* Ap = (int64_t*)malloc(sizeof(int64_t) * (n + 1)); int64_t* Ai = (int64_t*)malloc(sizeof(int64_t) * (nnz > 0 ? nnz : 1));
    for (int64_t j = 0; j <= n; ++j) Ap[j] = cp1[j] - 1;
    for (int64_t p = 0; p < nnz; ++p) Ai[p] = rv1[p] - 1;
    int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * n); min_degree_order(n, Ap, Ai, perm);
    ldl_fac f; memset(&f, 0, sizeof f); ldl_symbolic(&f, n, Ap, Ai, perm);
    int64_t d = ldl_numeric(&f, Ap, Ai, nz); int rc = 0;
    if (d == n) ldl_solve(&f, b, x); else rc = -1;
    ldl_free(&f); free(Ap); free(Ai); free(perm); return rc;
}

/* ============================================================================================ */
/* BlockSparseMatrix  src/BlockSparseMatrix.jl                                                    */
/* ============================================================================================ */
int64_t oracle_bsm_build(int64_t nrb, int64_t ncb, const int64_t* cp, const int64_t* rv, const int32_t* rs, const int32_t* cs, int64_t* nz) {
    (void)ncb; int64_t start = 1, ind = 0; /* :34-45 */
    for (int64_t row = 0; row < nrb; ++row) {
        int64_t rw = rs[row];
        for (int64_t p = cp[row] - 1; p < cp[row + 1] - 1; ++p) { int64_t col = rv[p] - 1; nz[ind++] = start; start += rw * cs[col]; }
    }
    return start - 1;
}
void oracle_bsm_to_dense(int64_t nrb, int64_t ncb, const int64_t* cp, const int64_t* rv, const int64_t* nz,
                         const int32_t* rs, const int32_t* cs, const double* data, double* out) { /* :245-264 */
    int64_t* r0 = (int64_t*)calloc(nrb + 1, sizeof(int64_t)); int64_t* c0 = (int64_t*)calloc(ncb + 1, sizeof(int64_t));
    for (int64_t i = 0; i < nrb; ++i) r0[i + 1] = r0[i] + rs[i];
    for (int64_t i = 0; i < ncb; ++i) c0[i + 1] = c0[i] + cs[i];
    int64_t m = r0[nrb], n = c0[ncb]; memset(out, 0, sizeof(double) * m * n);
    for (int64_t row = 0; row < nrb; ++row) for (int64_t p = cp[row] - 1; p < cp[row + 1] - 1; ++p) {
        int64_t col = rv[p] - 1; const double* blk = data + nz[p] - 1;
        for (int64_t c = 0; c < cs[col]; ++c) for (int64_t r = 0; r < rs[row]; ++r) out[(r0[row] + r) + m * (c0[col] + c)] = blk[r + rs[row] * c];
    }
    free(r0); free(c0);
}
void oracle_bsm_symmetrify_full(int64_t nb, const int64_t* cp, const int64_t* rv, const int64_t* nz, const int32_t* sz, const double* data, double* out) { /* :210-243 */
    int64_t* s0 = (int64_t*)calloc(nb + 1, sizeof(int64_t)); for (int64_t i = 0; i < nb; ++i) s0[i + 1] = s0[i] + sz[i];
    int64_t m = s0[nb]; memset(out, 0, sizeof(double) * m * m);
    for (int64_t row = 0; row < nb; ++row) for (int64_t p = cp[row] - 1; p < cp[row + 1] - 1; ++p) {
        int64_t col = rv[p] - 1; const double* blk = data + nz[p] - 1;
        for (int64_t c = 0; c < sz[col]; ++c) for (int64_t r = 0; r < sz[row]; ++r) {
            double v = blk[r + sz[row] * c]; out[(s0[row] + r) + m * (s0[col] + c)] = v;
            if (row != col) out[(s0[col] + c) + m * (s0[row] + r)] = v;
        }
    }
    free(s0);
}
/* transpose of indicestransposed (cacheindices :55-63): CSC with column c listing rows r */
static void bsm_transpose(int64_t nrb, int64_t ncb, const int64_t* cp, const int64_t* rv, const int64_t* nz,
                          int64_t** tcp, int64_t** trv, int64_t** tnz) {
    int64_t nnz = cp[nrb] - 1;
    *tcp = (int64_t*)calloc(ncb + 1, sizeof(int64_t)); *trv = (int64_t*)malloc(sizeof(int64_t) * (nnz + 1)); *tnz = (int64_t*)malloc(sizeof(int64_t) * (nnz + 1));
    int64_t* cnt = (int64_t*)calloc(ncb + 1, sizeof(int64_t));
    for (int64_t p = 0; p < nnz; ++p) cnt[rv[p] - 1]++;
    (*tcp)[0] = 1; for (int64_t c = 0; c < ncb; ++c) (*tcp)[c + 1] = (*tcp)[c] + cnt[c];
    for (int64_t c = 0; c < ncb; ++c) cnt[c] = (*tcp)[c] - 1; /* cursors */
    for (int64_t row = 1; row <= nrb; ++row) for (int64_t p = cp[row - 1] - 1; p < cp[row] - 1; ++p) {
        int64_t c = rv[p] - 1, q = cnt[c]++; (*trv)[q] = row; (*tnz)[q] = nz[p]; }
    free(cnt);
}
int64_t oracle_bsm_sparse_indices(int64_t nrb, int64_t ncb, const int64_t* cp, const int64_t* rv, const int64_t* nz,
                                  const int32_t* rs, const int32_t* cs, int64_t datalen, int symmetrify,
                                  int64_t* colptr, int64_t* rows, int64_t* indices) { /* :141-191 */
    int64_t *icp, *irv, *inz; bsm_transpose(nrb, ncb, cp, rv, nz, &icp, &irv, &inz); /* bsm.indices */
    /* diagonalblockspace :127-139 */
    int64_t diag = 0;
    if (symmetrify) for (int64_t col = 0; col < ncb; ++col) { int64_t ind = icp[col]; if (icp[col + 1] > ind) { int64_t row = irv[ind - 1]; if (row == col + 1) diag += (int64_t)rs[row - 1] * rs[row - 1]; } }
    int64_t nzvals = symmetrify ? datalen * 2 - diag : datalen;
    if (!colptr) { free(icp); free(irv); free(inz); return nzvals; }
    int64_t* startrow = (int64_t*)malloc(sizeof(int64_t) * (nrb + 1)); startrow[0] = 1; for (int64_t i = 0; i < nrb; ++i) startrow[i + 1] = startrow[i] + rs[i];
    int64_t ind = 0, col = 0; colptr[0] = 1;
    for (int64_t c_ = 0; c_ < ncb; ++c_) {
        int64_t cbs = cs[c_];
        int64_t lo0 = icp[c_] - 1, lo1 = icp[c_ + 1] - 1;         /* lower_rows */
        int64_t up0 = 0, up1 = 0;
        if (symmetrify) { up0 = cp[c_] - 1; up1 = cp[c_ + 1] - 1 - ((lo1 > lo0) && irv[lo0] == c_ + 1); }
        for (int64_t inner = 0; inner < cbs; ++inner) {
            for (int64_t r = up0; r < up1; ++r) { int64_t row = rv[r] - 1, s = startrow[row], c = rs[row], v = nz[r] + inner;
                for (int64_t i = 0; i < c; ++i) { rows[ind] = s + i; indices[ind] = v; ind++; v += cbs; } }
            for (int64_t r = lo0; r < lo1; ++r) { int64_t row = irv[r] - 1, s = startrow[row], c = rs[row], v = inz[r] + inner * c;
                for (int64_t i = 0; i < c; ++i) { rows[ind] = s + i; indices[ind] = v + i; ind++; } }
            col++; colptr[col] = ind + 1;
        }
    }
    free(startrow); free(icp); free(irv); free(inz);
    return ind;
}

/* ============================================================================================ */
/* problem                                                                                      */
/* ============================================================================================ */
typedef struct { int32_t res_kind, robust_kind; double rp[4]; int64_t ncost; int64_t* varind; double* data; int64_t ndata; int32_t dyn_n; } ogroup;   /* ndata: doubles per block (dynamic kinds: 1 + n or 0) */
struct oracle_problem {
    int64_t nvar, nstorage; int32_t *kind, *dim; int64_t* voff; double* vars[3];
    int32_t ngroups; ogroup* g; int64_t ncost_total;
};
oracle_problem* oracle_problem_create(int64_t nvar, const int32_t* vk, const int32_t* vd, int32_t ng, const nlls_cost_group* groups) {
    oracle_problem* p = (oracle_problem*)calloc(1, sizeof *p);
    p->nvar = nvar; p->kind = (int32_t*)malloc(sizeof(int32_t) * nvar); p->dim = (int32_t*)malloc(sizeof(int32_t) * nvar); p->voff = (int64_t*)malloc(sizeof(int64_t) * (nvar + 1));
    int64_t off = 0; for (int64_t i = 0; i < nvar; ++i) { p->kind[i] = vk[i]; p->dim[i] = vd[i]; p->voff[i] = off; off += var_storage(vk[i], vd[i]); }
    p->voff[nvar] = off; p->nstorage = off;
    for (int k = 0; k < 3; ++k) p->vars[k] = (double*)calloc(off > 0 ? off : 1, sizeof(double));
    p->ngroups = ng; p->g = (ogroup*)calloc(ng > 0 ? ng : 1, sizeof(ogroup));
    for (int gi = 0; gi < ng; ++gi) {
        ogroup* g = &p->g[gi]; const res_desc* d = &RES[groups[gi].res_kind];
        g->res_kind = groups[gi].res_kind; g->robust_kind = groups[gi].robust_kind; memcpy(g->rp, groups[gi].robust_params, sizeof g->rp);
        g->ncost = groups[gi].ncost; p->ncost_total += g->ncost;
        g->varind = (int64_t*)malloc(sizeof(int64_t) * (g->ncost * d->ndeps + 1)); memcpy(g->varind, groups[gi].varind, sizeof(int64_t) * g->ncost * d->ndeps);
        g->ndata = d->ndata; g->dyn_n = 0;
        if (IS_DYN_KIND(g->res_kind)) { /* dynamic-size blocks (src/autodiff.jl:96-121): n = the length of the block's variable, the same for the whole group */
            g->dyn_n = g->ncost > 0 ? vd[g->varind[0] - 1] : 0; g->ndata = g->res_kind == NLLS_RES_DYN_LINEAR ? 1 + g->dyn_n : g->res_kind == NLLS_RES_DYN_LINEARSQ ? g->dyn_n * (1 + (int64_t)g->dyn_n) : g->res_kind == NLLS_COST_DYN_LINEAR ? g->dyn_n : 0; }
        g->data = (double*)malloc(sizeof(double) * (g->ncost * g->ndata + 1)); if (g->ndata > 0) memcpy(g->data, groups[gi].data, sizeof(double) * g->ncost * g->ndata);
    }
    return p;
}
void oracle_problem_destroy(oracle_problem* p) {
    if (!p) return;
    for (int gi = 0; gi < p->ngroups; ++gi) { free(p->g[gi].varind); free(p->g[gi].data); }
    free(p->g); for (int k = 0; k < 3; ++k) free(p->vars[k]); free(p->kind); free(p->dim); free(p->voff); free(p);
}
int64_t oracle_problem_storage(const oracle_problem* p) { return p->nstorage; }
void oracle_set_variables(oracle_problem* p, int32_t w, const double* x) { memcpy(p->vars[w], x, sizeof(double) * p->nstorage); }
void oracle_get_variables(const oracle_problem* p, int32_t w, double* x) { memcpy(x, p->vars[w], sizeof(double) * p->nstorage); }

/* ---- per-block maths ------------------------------------------------------------------------ */
#define MAXP 16
/* residual value only (computeresidual with plain numbers) */
static void block_residual(const oracle_problem* p, const double* vars, const ogroup* g, int64_t ci, double* r) {
    const res_desc* d = &RES[g->res_kind]; jet sv[4][12]; jet* svp[4]; jet rj[4]; int ns = 0;
    for (int s = d->adaptive; s < d->ndeps; ++s) {
        int64_t vi = g->varind[ci * d->ndeps + s] - 1;
        var_update_jet(p->kind[vi], p->dim[vi], vars + p->voff[vi], -1, 0, sv[ns]); svp[ns] = sv[ns]; ns++;
    }
    res_eval(g->res_kind, g->data + ci * d->ndata, svp, rj, 0);
    for (int m = 0; m < d->nres; ++m) r[m] = rj[m].v;
}
/* computerescost  src/residual.jl:49-55 */
/* dynamic-size residual blocks: computeresjacdynamic (src/autodiff.jl:96-121) of the two registered residuals is known in closed form --
 * LinearResidual X'w - y (test/dynamicvars.jl:3-11): J = X';  NormResidual w (test/dynamicvars.jl:13-21): J = I.
 * cost = 0.5 r'r; with H / gv (n x n col-major, n) also J'J and J'r (src/residual.jl:72-74). */
static double dyn_block_plain(const oracle_problem* p, const double* vars, const ogroup* g, int64_t ci, double* gv, double* H);
/* ... under a robust kernel like any other residual (src/residual.jl:76-101): cost = rho(r'r) / 2, g = rho' J'r, H = rho' J'J + 2 rho'' (J'r)(J'r)' */
static double dyn_block(const oracle_problem* p, const double* vars, const ogroup* g, int64_t ci, double* gv, double* H) {
    const double c = 2.0 * dyn_block_plain(p, vars, g, ci, gv, H);              /* r'r (the non-squared cost kind takes no kernel: refused at setup) */
    if (g->robust_kind == NLLS_ROBUST_NONE || g->res_kind == NLLS_COST_DYN_LINEAR) return 0.5 * c;
    const int n = g->dyn_n; double o[3]; fixed_robustifydcost(g->robust_kind, g->rp, c, o);
    if (gv) {
        if (o[1] != 1) for (size_t i = 0; i < (size_t)n * n; ++i) H[i] *= o[1];                                                    /* :91-93 */
        if (o[2] != 0) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) H[i + (size_t)n * j] += ((2 * o[2]) * gv[i]) * gv[j];   /* :95-97 */
        if (o[1] != 1) for (int i = 0; i < n; ++i) gv[i] *= o[1];                                                                   /* :99-101 */
    }
    return 0.5 * o[0];
}
static double dyn_block_plain(const oracle_problem* p, const double* vars, const ogroup* g, int64_t ci, double* gv, double* H) {
    const int n = g->dyn_n; const double* w = vars + p->voff[g->varind[ci] - 1];
    if (g->res_kind == NLLS_RES_DYN_LINEAR) {
        const double* dd = g->data + ci * g->ndata; const double* X = dd + 1; double r = -dd[0];
        for (int i = 0; i < n; ++i) r += X[i] * w[i];
        if (gv) for (int i = 0; i < n; ++i) { gv[i] = X[i] * r; for (int j = 0; j < n; ++j) H[i + (size_t)n * j] = X[i] * X[j]; }
        return 0.5 * r * r;
    }
    if (g->res_kind == NLLS_RES_DYN_LINEARSQ) { /* LinearResidualDynamic X*w - y, X square col-major (test/nonsquaredcost.jl:16-26): J = X */
        const double* dd = g->data + ci * g->ndata; const double* X = dd + n; double* r = (double*)malloc(sizeof(double) * n); double c2 = 0;
        for (int i = 0; i < n; ++i) { double t = -dd[i]; for (int j = 0; j < n; ++j) t += X[i + (size_t)n * j] * w[j]; r[i] = t; c2 += t * t; }
        if (gv) for (int j = 0; j < n; ++j) { double t = 0; for (int i = 0; i < n; ++i) t += X[i + (size_t)n * j] * r[i]; gv[j] = t;
            for (int k = 0; k < n; ++k) { double h = 0; for (int i = 0; i < n; ++i) h += X[i + (size_t)n * j] * X[i + (size_t)n * k]; H[j + (size_t)n * k] = h; } }
        free(r); return 0.5 * c2;
    }
    if (g->res_kind == NLLS_COST_DYN_LINEAR) { /* LinearCostDynamic y'w (test/nonsquaredcost.jl:39-46): computecostgradhess src/autodiff.jl:144-159 -> y'w, y, 0 */
        const double* y = g->data + ci * g->ndata; double c2 = 0; for (int i = 0; i < n; ++i) c2 += y[i] * w[i];
        if (gv) for (int i = 0; i < n; ++i) { gv[i] = y[i]; for (int j = 0; j < n; ++j) H[i + (size_t)n * j] = 0.0; }
        return c2;
    }
    double c = 0; for (int i = 0; i < n; ++i) c += w[i] * w[i];
    if (gv) for (int i = 0; i < n; ++i) { gv[i] = w[i]; for (int j = 0; j < n; ++j) H[i + (size_t)n * j] = i == j ? 1.0 : 0.0; }
    return 0.5 * c;
}
static double block_cost(const oracle_problem* p, const double* vars, const ogroup* g, int64_t ci) {
    const res_desc* d = &RES[g->res_kind];
    if (IS_DYN_KIND(g->res_kind)) return dyn_block(p, vars, g, ci, 0, 0);
    if (IS_COST_KIND(g->res_kind)) { /* computecost: the value itself (src/cost.jl:10-13 over an AbstractCost) */
        const double* st[4]; int start[4];
        for (int s = 0; s < d->ndeps; ++s) { st[s] = vars + p->voff[g->varind[ci * d->ndeps + s] - 1]; start[s] = -1; }
        return cost_eval2(g->res_kind, g->data + ci * d->ndata, st, start).v;
    }
    double r[4]; block_residual(p, vars, g, ci, r);
    double s = 0; for (int m = 0; m < d->nres; ++m) s += r[m] * r[m];
    if (d->adaptive) { int64_t kv = g->varind[ci * d->ndeps] - 1; cgauss k = cg_make(vars + p->voff[kv]); return 0.5 * cg_robustify(&k, s); }
    return 0.5 * fixed_robustify(g->robust_kind, g->rp, s);
}
/* computeresjacstatic  src/autodiff.jl:81-93 ; slotfree[s] != 0 => slot s unfixed (kernel slot ignored).
 * J is M x n col-major; returns n */
static int block_resjac(const oracle_problem* p, const double* vars, const ogroup* g, int64_t ci, const int* slotfree, double* r, double* J) {
    const res_desc* d = &RES[g->res_kind]; int n = 0, start[4];
    for (int s = d->adaptive; s < d->ndeps; ++s) { int64_t vi = g->varind[ci * d->ndeps + s] - 1; if (slotfree[s]) { start[s] = n; n += var_dof(p->kind[vi], p->dim[vi]); } else start[s] = -1; }
    jet sv[4][12]; jet* svp[4]; jet rj[4]; int ns = 0;
    for (int s = d->adaptive; s < d->ndeps; ++s) { int64_t vi = g->varind[ci * d->ndeps + s] - 1;
        var_update_jet(p->kind[vi], p->dim[vi], vars + p->voff[vi], start[s], n, sv[ns]); svp[ns] = sv[ns]; ns++; }
    res_eval(g->res_kind, g->data + ci * d->ndata, svp, rj, n);
    for (int m = 0; m < d->nres; ++m) { r[m] = rj[m].v; for (int j = 0; j < n; ++j) J[m + d->nres * j] = rj[m].d[j]; }
    return n;
}
/* computerescostgradhess  src/residual.jl:57-111.  slotfree over ALL getvars slots (kernel = slot 0
 * for adaptive residuals).  Returns P (length of g); *cost = 0.5*rho. H is P x P col-major. */
static int block_costgradhess(const oracle_problem* p, const double* vars, const ogroup* g, int64_t ci, const int* slotfree,
                              double* cost, double* gv, double* H) {
    const res_desc* d = &RES[g->res_kind]; int M = d->nres;
    if (IS_COST_KIND(g->res_kind)) { /* computecostgradhess  src/autodiff.jl:144-159: value, gradient, Hessian of the cost through update() */
        const double* st[4]; int start[4], n = 0;
        for (int s = 0; s < d->ndeps; ++s) { int64_t vi = g->varind[ci * d->ndeps + s] - 1; st[s] = vars + p->voff[vi];
            if (slotfree[s]) { start[s] = n; n += var_dof(p->kind[vi], p->dim[vi]); } else start[s] = -1; }
        jet2 c = cost_eval2(g->res_kind, g->data + ci * d->ndata, st, start);
        *cost = c.v; for (int i = 0; i < n; ++i) { gv[i] = c.g[i]; for (int j = 0; j < n; ++j) H[i + n * j] = c.h[i][j]; }
        return n;
    }
    int kernel_opt = d->adaptive && slotfree[0];
    int anyres = 0; for (int s = d->adaptive; s < d->ndeps; ++s) anyres |= slotfree[s];
    const double* kst = 0; if (d->adaptive) { int64_t kv = g->varind[ci * d->ndeps] - 1; kst = vars + p->voff[kv]; }
    if (!anyres) { /* :60-66 only the kernel is optimised */
        double r[4]; block_residual(p, vars, g, ci, r); double s = 0; for (int m = 0; m < M; ++m) s += r[m] * r[m];
        double val, dc[4], d2c[16]; oracle_robustifydkernel(kst, s, &val, dc, d2c);
        *cost = 0.5 * val; for (int i = 0; i < 3; ++i) { gv[i] = dc[i]; for (int j = 0; j < 3; ++j) H[i + 3 * j] = d2c[i + 4 * j]; }
        return 3;
    }
    double r[4], J[4 * MAXP]; int n = block_resjac(p, vars, g, ci, slotfree, r, J); /* :69 */
    double c = 0; for (int m = 0; m < M; ++m) c += r[m] * r[m];                          /* :72 */
    double gr[MAXP], Hr[MAXP * MAXP];
    for (int j = 0; j < n; ++j) { double s = 0; for (int m = 0; m < M; ++m) s += J[m + M * j] * r[m]; gr[j] = s; } /* :73 */
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) { double s = 0; for (int m = 0; m < M; ++m) s += J[m + M * i] * J[m + M * j]; Hr[i + n * j] = s; } /* :74 */
    double rho, dc, d2c, dck[4], d2ck[16], dkdv[3 * MAXP];
    if (!kernel_opt) { /* :76-78 */
        double o[3]; if (d->adaptive) { cgauss k = cg_make(kst); cg_robustifydcost(&k, c, o); } else fixed_robustifydcost(g->robust_kind, g->rp, c, o);
        rho = o[0]; dc = o[1]; d2c = o[2];
    } else { /* :79-88 */
        oracle_robustifydkernel(kst, c, &rho, dck, d2ck); dc = dck[3]; d2c = d2ck[3 + 4 * 3];
        for (int j = 0; j < n; ++j) for (int k = 0; k < 3; ++k) dkdv[j + n * k] = gr[j] * d2ck[k + 4 * 3];
    }
    if (dc != 1) for (int i = 0; i < n * n; ++i) Hr[i] *= dc;                                          /* :91-93 */
    if (d2c != 0) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) Hr[i + n * j] += ((2 * d2c) * gr[i]) * gr[j]; /* :95-97 */
    if (dc != 1) for (int i = 0; i < n; ++i) gr[i] *= dc;                                               /* :99-101 */
    *cost = 0.5 * rho;
    if (!kernel_opt) { memcpy(gv, gr, sizeof(double) * n); memcpy(H, Hr, sizeof(double) * n * n); return n; }
    int P = n + 3; /* :103-107 */
    for (int k = 0; k < 3; ++k) gv[k] = dck[k];
    for (int i = 0; i < n; ++i) gv[3 + i] = gr[i];
    for (int j = 0; j < P; ++j) for (int i = 0; i < P; ++i) {
        double v;
        if (i < 3 && j < 3) v = d2ck[i + 4 * j];
        else if (i >= 3 && j < 3) v = dkdv[(i - 3) + n * j];
        else if (i < 3 && j >= 3) v = dkdv[(j - 3) + n * i];
        else v = Hr[(i - 3) + n * (j - 3)];
        H[i + P * j] = v;
    }
    return P;
}
int oracle_block_costgradhess(const oracle_problem* p, int32_t which, int32_t gi, int64_t ci, double* cost, double* g, double* H) {
    int sf[4] = {1, 1, 1, 1}; return block_costgradhess(p, p->vars[which], &p->g[gi], ci, sf, cost, g, H);
}
int oracle_block_resjac(const oracle_problem* p, int32_t which, int32_t gi, int64_t ci, double* r, double* J) {
    int sf[4] = {1, 1, 1, 1}; return block_resjac(p, p->vars[which], &p->g[gi], ci, sf, r, J);
}

typedef struct { const oracle_problem* p; const double* vars; const ogroup* g; oracle_ls* ls; } sumctx;
static double cost_fn(void* c, int64_t i) { sumctx* s = (sumctx*)c; return block_cost(s->p, s->vars, s->g, i); }
double oracle_cost(const oracle_problem* p, int32_t which) { /* cost.jl:11 + VectorRepo.jl:64-69 */
    double total = 0;
    for (int gi = 0; gi < p->ngroups; ++gi) { sumctx c = {p, p->vars[which], &p->g[gi], 0}; total += pairwise_sum(cost_fn, &c, 0, p->g[gi].ncost); }
    return total;
}

/* ============================================================================================ */
/* linear system  src/linearsystem.jl                                                            */
/* ============================================================================================ */
struct oracle_ls {
    int is_sparse; int64_t nvar, nblocks, ndof, nnz_data, nstored;
    uint64_t* blockindices; int32_t* blocksizes; int64_t* boffsets; /* 1-based, length nblocks(+1 dense) */
    int64_t *it_cp, *it_rv, *it_nz;            /* indicestransposed */
    double *data, *b, *x;
    int64_t h_nnz, *h_cp, *h_rv, *sparseindices; double* h_nz; /* hessian CSC (1-based) */
    ldl_fac fac; int have_fac; int64_t *Ap0, *Ai0;
};
static int64_t it_lookup(const oracle_ls* ls, int64_t j, int64_t i) { /* indicestransposed[j,i], 1-based; 0 if absent */
    int64_t lo = ls->it_cp[i - 1] - 1, hi = ls->it_cp[i] - 2;
    while (lo <= hi) { int64_t m = (lo + hi) >> 1; if (ls->it_rv[m] == j) return ls->it_nz[m]; if (ls->it_rv[m] < j) lo = m + 1; else hi = m - 1; }
    return 0;
}
oracle_ls* oracle_makesymmvls(const oracle_problem* p, const uint64_t* bi, int32_t flags) { /* :91-124 */
    oracle_ls* ls = (oracle_ls*)calloc(1, sizeof *ls);
    ls->nvar = p->nvar; ls->blockindices = (uint64_t*)malloc(sizeof(uint64_t) * p->nvar); memcpy(ls->blockindices, bi, sizeof(uint64_t) * p->nvar);
    int64_t nb = 0; for (int64_t i = 0; i < p->nvar; ++i) if (bi[i] > (uint64_t)nb) nb = (int64_t)bi[i];
    ls->nblocks = nb; ls->blocksizes = (int32_t*)calloc(nb > 0 ? nb : 1, sizeof(int32_t));
    for (int64_t i = 0; i < p->nvar; ++i) if (bi[i]) ls->blocksizes[bi[i] - 1] = var_dof(p->kind[i], p->dim[i]);
    int64_t len = 0; for (int64_t k = 0; k < nb; ++k) len += ls->blocksizes[k];
    ls->ndof = len;
    int sparse = 0;
    if (len >= 40 || (flags & NLLS_FLAG_FORCE_SPARSE)) { /* :105-121 */
        /* sparsity = triu(V*V' .> 0) over unfixed rows: pairs (a <= b) of blocks sharing a cost */
        int64_t cap = 0; for (int gi = 0; gi < p->ngroups; ++gi) { int nd = RES[p->g[gi].res_kind].ndeps; cap += p->g[gi].ncost * (nd * (nd + 1) / 2); }
        int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (cap + 1)); int64_t nk = 0;
        for (int gi = 0; gi < p->ngroups; ++gi) { const ogroup* g = &p->g[gi]; int nd = RES[g->res_kind].ndeps;
            for (int64_t c = 0; c < g->ncost; ++c) for (int s = 0; s < nd; ++s) { uint64_t a = bi[g->varind[c * nd + s] - 1]; if (!a) continue;
                for (int t = 0; t <= s; ++t) { uint64_t b = bi[g->varind[c * nd + t] - 1]; if (!b) continue;
                    int64_t hi = a > b ? a : b, lo = a > b ? b : a; keys[nk++] = (hi - 1) * nb + (lo - 1); } } }
        qsort(keys, nk, sizeof(int64_t), cmp_i64);
        int64_t nu = 0; for (int64_t k = 0; k < nk; ++k) if (k == 0 || keys[k] != keys[k - 1]) keys[nu++] = keys[k];
        /* block_sparse_nnz utils.jl:110-120 and the decision utils.jl:108 */
        int64_t bnnz = 0; for (int64_t k = 0; k < nu; ++k) bnnz += (int64_t)ls->blocksizes[keys[k] / nb] * ls->blocksizes[keys[k] % nb];
        if ((flags & NLLS_FLAG_FORCE_SPARSE) || (bnnz * 64) < (25 * len * (len - 40))) {
            sparse = 1; ls->nstored = nu;
            ls->it_cp = (int64_t*)calloc(nb + 1, sizeof(int64_t)); ls->it_rv = (int64_t*)malloc(sizeof(int64_t) * (nu + 1)); ls->it_nz = (int64_t*)malloc(sizeof(int64_t) * (nu + 1));
            /* keys sorted by (row=hi, col=lo): column `row` of sparsitytransposed lists cols ascending */
            int64_t q = 0; for (int64_t row = 0; row < nb; ++row) { ls->it_cp[row] = q + 1; while (q < nu && keys[q] / nb == row) { ls->it_rv[q] = keys[q] % nb + 1; q++; } }
            ls->it_cp[nb] = nu + 1;
            ls->nnz_data = oracle_bsm_build(nb, nb, ls->it_cp, ls->it_rv, ls->blocksizes, ls->blocksizes, ls->it_nz);
        }
        free(keys);
    }
    ls->is_sparse = sparse;
    /* computestartindices :36-41 */
    ls->boffsets = (int64_t*)malloc(sizeof(int64_t) * (nb + 1)); { int64_t o = 1; for (int64_t k = 0; k < nb; ++k) { ls->boffsets[k] = o; o += ls->blocksizes[k]; } ls->boffsets[nb] = o; }
    ls->b = (double*)calloc(len > 0 ? len : 1, sizeof(double)); ls->x = (double*)calloc(len > 0 ? len : 1, sizeof(double));
    if (sparse) {
        ls->data = (double*)calloc(ls->nnz_data, sizeof(double));
        /* MultiVariateLSsparse :54-70 */
        ls->h_nnz = oracle_bsm_sparse_indices(nb, nb, ls->it_cp, ls->it_rv, ls->it_nz, ls->blocksizes, ls->blocksizes, ls->nnz_data, 1, 0, 0, 0);
        ls->h_cp = (int64_t*)malloc(sizeof(int64_t) * (len + 1)); ls->h_rv = (int64_t*)malloc(sizeof(int64_t) * ls->h_nnz); ls->sparseindices = (int64_t*)malloc(sizeof(int64_t) * ls->h_nnz);
        ls->h_nz = (double*)malloc(sizeof(double) * ls->h_nnz);
        oracle_bsm_sparse_indices(nb, nb, ls->it_cp, ls->it_rv, ls->it_nz, ls->blocksizes, ls->blocksizes, ls->nnz_data, 1, ls->h_cp, ls->h_rv, ls->sparseindices);
    } else { ls->nnz_data = len * len; ls->data = (double*)calloc(len * len > 0 ? len * len : 1, sizeof(double)); }
    return ls;
}
void oracle_ls_destroy(oracle_ls* ls) {
    if (!ls) return;
    if (ls->have_fac) ldl_free(&ls->fac);
    free(ls->Ap0); free(ls->Ai0);
    free(ls->blockindices); free(ls->blocksizes); free(ls->boffsets); free(ls->it_cp); free(ls->it_rv); free(ls->it_nz);
    free(ls->data); free(ls->b); free(ls->x); free(ls->h_cp); free(ls->h_rv); free(ls->sparseindices); free(ls->h_nz); free(ls);
}
void oracle_ls_info(const oracle_ls* ls, nlls_info* o) {
    memset(o, 0, sizeof *o); o->is_sparse = ls->is_sparse; o->nvar = ls->nvar; o->nblocks = ls->nblocks; o->ndof = ls->ndof;
    o->nnz_data = ls->nnz_data; o->nblocks_stored = ls->nstored;
}
double* oracle_ls_data(oracle_ls* ls) { return ls->data; }
double* oracle_ls_b(oracle_ls* ls) { return ls->b; }
double* oracle_ls_x(oracle_ls* ls) { return ls->x; }
void oracle_ls_bsm_index(const oracle_ls* ls, int64_t* cp, int64_t* rv, int64_t* nz, int64_t* bo) {
    if (ls->is_sparse) { if (cp) memcpy(cp, ls->it_cp, sizeof(int64_t) * (ls->nblocks + 1)); if (rv) memcpy(rv, ls->it_rv, sizeof(int64_t) * ls->nstored); if (nz) memcpy(nz, ls->it_nz, sizeof(int64_t) * ls->nstored); }
    if (bo) memcpy(bo, ls->boffsets, sizeof(int64_t) * ls->nblocks);
}
/* block(A, i, j) += a[rangei, rangej]  (BSM :102-105 / BDM :14-17) ; i,j 1-based block ids */
static void add_block(oracle_ls* ls, int64_t i, int64_t j, const double* H, int P, int ri, int ni, int rj, int nj) {
    if (ls->is_sparse) { double* blk = ls->data + it_lookup(ls, j, i) - 1;
        for (int c = 0; c < nj; ++c) for (int r = 0; r < ni; ++r) blk[r + ni * c] += H[(ri + r) + P * (rj + c)]; }
    else { int64_t n = ls->ndof, r0 = ls->boffsets[i - 1] - 1, c0 = ls->boffsets[j - 1] - 1;
        for (int c = 0; c < nj; ++c) for (int r = 0; r < ni; ++r) ls->data[(r0 + r) + n * (c0 + c)] += H[(ri + r) + P * (rj + c)]; }
}
/* costgradhess! for one block  src/cost.jl:29-52 + updatesymlinearsystem! linearsystem.jl:132-175 */
static double grad_fn(void* cv, int64_t ci) {
    sumctx* s = (sumctx*)cv; const oracle_problem* p = s->p; const ogroup* g = s->g; oracle_ls* ls = s->ls; const res_desc* d = &RES[g->res_kind];
    uint64_t bidx[4]; int sf[4], any = 0, dofs[4];
    for (int k = 0; k < d->ndeps; ++k) { int64_t vi = g->varind[ci * d->ndeps + k] - 1; bidx[k] = ls->blockindices[vi]; sf[k] = bidx[k] != 0; any |= sf[k]; dofs[k] = var_dof(p->kind[vi], p->dim[vi]); }
    if (!any) return block_cost(p, s->vars, g, ci); /* cost.jl:51 */
    if (IS_DYN_KIND(g->res_kind)) { /* one variable of run-time size: heap storage like the reference's dynamic path */
        const int n = g->dyn_n; double* gd = (double*)malloc(sizeof(double) * (size_t)n * (n + 1)); double* Hd = gd + n;
        const double cd = dyn_block(p, s->vars, g, ci, gd, Hd);
        for (int r = 0; r < n; ++r) ls->b[ls->boffsets[bidx[0] - 1] - 1 + r] += gd[r];
        add_block(ls, bidx[0], bidx[0], Hd, n, 0, n, 0, n);
        free(gd); return cd;
    }
    double c, gv[MAXP], H[MAXP * MAXP]; int P = block_costgradhess(p, s->vars, g, ci, sf, &c, gv, H);
    int loff[4], o = 0; for (int k = 0; k < d->ndeps; ++k) { loff[k] = o; if (sf[k]) o += dofs[k]; }
    for (int i = 0; i < d->ndeps; ++i) if (sf[i]) {
        for (int r = 0; r < dofs[i]; ++r) ls->b[ls->boffsets[bidx[i] - 1] - 1 + r] += gv[loff[i] + r];        /* updateb! :159-170 */
        add_block(ls, bidx[i], bidx[i], H, P, loff[i], dofs[i], loff[i], dofs[i]);                            /* :140 */
        for (int j = 0; j < i; ++j) if (sf[j]) {
            if (bidx[i] >= bidx[j]) add_block(ls, bidx[i], bidx[j], H, P, loff[i], dofs[i], loff[j], dofs[j]); /* :148-149 */
            else add_block(ls, bidx[j], bidx[i], H, P, loff[j], dofs[j], loff[i], dofs[i]);                   /* :150-151 */
        }
    }
    return c;
}
double oracle_costgradhess(const oracle_problem* p, int32_t which, oracle_ls* ls) {
    memset(ls->b, 0, sizeof(double) * ls->ndof); memset(ls->data, 0, sizeof(double) * ls->nnz_data); /* zero! :192-195 */
    double total = 0;
    for (int gi = 0; gi < p->ngroups; ++gi) { sumctx c = {p, p->vars[which], &p->g[gi], ls}; total += pairwise_sum(grad_fn, &c, 0, p->g[gi].ncost); }
    return total;
}
static void gethessian(oracle_ls* ls) { /* :182-189 */
    if (ls->is_sparse) for (int64_t i = 0; i < ls->h_nnz; ++i) ls->h_nz[i] = ls->data[ls->sparseindices[i] - 1];
    else { int64_t n = ls->ndof; for (int64_t r = 1; r < n; ++r) for (int64_t c = 0; c < r; ++c) ls->data[c + n * r] = ls->data[r + n * c]; } /* BlockDenseMatrix.jl:24-34 */
}
double oracle_max_abs_diag(const oracle_ls* ls) { /* iterators.jl:131-137 */
    double m = 0; int64_t n = ls->ndof;
    if (!ls->is_sparse) { for (int64_t i = 0; i < n; ++i) m = fmax(m, fabs(ls->data[i + n * i])); return m; }
    for (int64_t k = 1; k <= ls->nblocks; ++k) { const double* blk = ls->data + it_lookup(ls, k, k) - 1; int bs = ls->blocksizes[k - 1]; for (int i = 0; i < bs; ++i) m = fmax(m, fabs(blk[i + bs * i])); }
    return m;
}
int oracle_solve_damped(oracle_ls* ls, double lambda) { /* iterators.jl:149-152, linearsolver.jl:28-32 */
    int64_t n = ls->ndof; gethessian(ls);
    if (!ls->is_sparse) {
        double* A = (double*)malloc(sizeof(double) * n * n); memcpy(A, ls->data, sizeof(double) * n * n);
        for (int64_t i = 0; i < n; ++i) A[i + n * i] += lambda;                       /* uniformscaling! BSM.jl:83-88 */
        oracle_solve_dense(ls->x, A, ls->b, n); free(A);
    } else {
        for (int64_t c = 0; c < n; ++c) for (int64_t q = ls->h_cp[c] - 1; q < ls->h_cp[c + 1] - 1; ++q) if (ls->h_rv[q] - 1 == c) ls->h_nz[q] += lambda;
        if (!ls->have_fac) { /* ldl_analyze once, linearsystem.jl:68 */
            int64_t nnz = ls->h_nnz; ls->Ap0 = (int64_t*)malloc(sizeof(int64_t) * (n + 1)); ls->Ai0 = (int64_t*)malloc(sizeof(int64_t) * nnz);
            for (int64_t j = 0; j <= n; ++j) ls->Ap0[j] = ls->h_cp[j] - 1;
            for (int64_t q = 0; q < nnz; ++q) ls->Ai0[q] = ls->h_rv[q] - 1;
            /* block-level minimum degree, expanded to scalars */
            int64_t nb = ls->nblocks; int64_t* bcp = (int64_t*)calloc(nb + 1, sizeof(int64_t));
            for (int64_t row = 0; row < nb; ++row) for (int64_t q = ls->it_cp[row] - 1; q < ls->it_cp[row + 1] - 1; ++q) { int64_t col = ls->it_rv[q] - 1; if (col != row) { bcp[row + 1]++; bcp[col + 1]++; } }
            for (int64_t k = 0; k < nb; ++k) bcp[k + 1] += bcp[k];
            int64_t* bai = (int64_t*)malloc(sizeof(int64_t) * (bcp[nb] + 1)); int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (nb + 1)); memcpy(cur, bcp, sizeof(int64_t) * (nb + 1));
            for (int64_t row = 0; row < nb; ++row) for (int64_t q = ls->it_cp[row] - 1; q < ls->it_cp[row + 1] - 1; ++q) { int64_t col = ls->it_rv[q] - 1; if (col != row) { bai[cur[row]++] = col; bai[cur[col]++] = row; } }
            int64_t* bperm = (int64_t*)malloc(sizeof(int64_t) * nb); min_degree_order(nb, bcp, bai, bperm);
            int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * n); int64_t o = 0;
            for (int64_t k = 0; k < nb; ++k) { int64_t blk = bperm[k]; for (int r = 0; r < ls->blocksizes[blk]; ++r) perm[o++] = ls->boffsets[blk] - 1 + r; }
            ldl_symbolic(&ls->fac, n, ls->Ap0, ls->Ai0, perm); ls->have_fac = 1;
            free(bcp); free(bai); free(cur); free(bperm); free(perm);
        }
        int64_t d = ldl_numeric(&ls->fac, ls->Ap0, ls->Ai0, ls->h_nz);
        if (d != n) return -1;
        ldl_solve(&ls->fac, ls->b, ls->x);
    }
    for (int64_t i = 0; i < n; ++i) ls->x[i] = -ls->x[i]; /* negate! iterators.jl:3 */
    return 0;
}
double oracle_quadform(const oracle_ls* lsc, const double* x, double lambda) {
    oracle_ls* ls = (oracle_ls*)lsc; gethessian(ls); int64_t n = ls->ndof; double q;
    if (ls->is_sparse) q = oracle_fast_bAb_csc(ls->h_cp, ls->h_rv, ls->h_nz, x, n); else q = oracle_fast_bAb_dense(ls->data, x, n);
    double s = 0; for (int64_t i = 0; i < n; ++i) s += x[i] * x[i];
    return q + lambda * s;
}
void oracle_update(oracle_problem* p, int32_t to, int32_t from, const oracle_ls* ls, const double* step) { /* :206-213 */
    const double* st = step ? step : ls->x;
    for (int64_t i = 0; i < p->nvar; ++i) { uint64_t j = ls->blockindices[i];
        if (j) oracle_var_update(p->kind[i], p->dim[i], p->vars[from] + p->voff[i], st + ls->boffsets[j - 1] - 1, p->vars[to] + p->voff[i]);
        /* fixed variables are left untouched in `to` (the reference only assigns unfixed ones) */
    }
}

/* ============================================================================================ */
/* optimize!  src/optimize.jl, src/iterators.jl                                                  */
/* ============================================================================================ */
void oracle_default_options(oracle_options* o) { /* structs.jl:33 */
    o->reldcost = 1e-15; o->absdcost = 1e-15; o->dstep = 1e-15; o->maxfails = 3; o->maxiters = 100; o->maxtime = 30.0; o->iterator = 1; o->store_costs = 0;
}
typedef struct { double bestcost, startcost, timecost, timegradient, timesolver; int64_t iternum, costcomputations, gradientcomputations, linearsolvers; } idata;
static void swapvars(oracle_problem* p, int a, int b) { double* t = p->vars[a]; p->vars[a] = p->vars[b]; p->vars[b] = t; }
static double maxabs(const double* x, int64_t n) { double m = 0; for (int64_t i = 0; i < n; ++i) { double a = fabs(x[i]); if (a > m || a != a) m = a; } return m; }
static double vnorm(const double* x, int64_t n) { double s = 0; for (int64_t i = 0; i < n; ++i) s += x[i] * x[i]; return sqrt(s); }
static double vdot(const double* a, const double* b, int64_t n) { double s = 0; for (int64_t i = 0; i < n; ++i) s += a[i] * b[i]; return s; }
static double timed_cost(oracle_problem* p, idata* d) { double t0 = now_s(); double c = oracle_cost(p, NLLS_VARS_NEXT); d->timecost += now_s() - t0; d->costcomputations++; return c; }

static double iterate_newton(oracle_problem* p, oracle_ls* ls, idata* d) { /* iterators.jl:15-27 */
    double t0 = now_s(); oracle_solve_damped(ls, 0.0); d->timesolver += now_s() - t0; d->linearsolvers++;
    oracle_update(p, NLLS_VARS_NEXT, NLLS_VARS_CURRENT, ls, 0);
    return timed_cost(p, d);
}
static double iterate_levmar(oracle_problem* p, oracle_ls* ls, idata* d, const oracle_options* opt, double* lambda) { /* iterators.jl:139-172 */
    if (*lambda == 0) *lambda = oracle_max_abs_diag(ls) * 1e-6; /* :142-144, :131-137 */
    double mu = 2.0;
    for (;;) {
        double t0 = now_s(); oracle_solve_damped(ls, *lambda); d->timesolver += now_s() - t0; d->linearsolvers++; /* :149-153 */
        oracle_update(p, NLLS_VARS_NEXT, NLLS_VARS_CURRENT, ls, 0);                                                /* :155 */
        double cost_ = timed_cost(p, d);                                                                            /* :157 */
        if (!(cost_ > d->bestcost) || maxabs(ls->x, ls->ndof) < opt->dstep) {                                       /* :160 */
            double q = (cost_ - d->bestcost) / (0.5 * oracle_quadform(ls, ls->x, 0.0) + vdot(ls->b, ls->x, ls->ndof)); /* :162-163 */
            *lambda *= q < 0.983 ? 1 - pow(2 * q - 1, 3) : 0.1;                                                     /* :164 */
            return cost_;
        }
        *lambda *= mu; mu *= 2.0; /* :169-170 */
    }
}
static double iterate_dogleg(oracle_problem* p, oracle_ls* ls, idata* d, const oracle_options* opt, double* tr, double* cauchy) { /* iterators.jl:47-115 */
    int64_t n = ls->ndof; double* x = ls->x; const double* g = ls->b; double t0 = now_s();
    double gnorm2 = vdot(g, g, n);
    double a = gnorm2 / (oracle_quadform(ls, g, 0.0) + DBL_MIN);
    for (int64_t i = 0; i < n; ++i) cauchy[i] = -a * g[i];
    double alpha2 = a * a * gnorm2, alpha = sqrt(alpha2), beta = 0;
    if (*tr == 0) *tr = alpha;
    if (alpha < *tr) { oracle_solve_damped(ls, 0.0); beta = vnorm(x, n); d->linearsolvers++; }
    d->timesolver += now_s() - t0;
    double cost_ = d->bestcost;
    for (;;) {
        double linear_approx;
        if (!(alpha < *tr)) { for (int64_t i = 0; i < n; ++i) x[i] = (*tr / alpha) * cauchy[i]; linear_approx = *tr * (2 * alpha - *tr) / (2 * a); }
        else if (beta <= *tr) linear_approx = cost_;
        else {
            for (int64_t i = 0; i < n; ++i) x[i] -= cauchy[i];
            double sq_leg = vdot(x, x, n), c = vdot(cauchy, x, n), trsq = *tr * *tr - alpha2, step = sqrt(c * c + sq_leg * trsq);
            if (c <= 0) step = (-c + step) / sq_leg; else step = trsq / (c + step);
            for (int64_t i = 0; i < n; ++i) x[i] = x[i] * step + cauchy[i];
            linear_approx = 0.5 * (a * (1 - step) * (1 - step) * gnorm2) + step * (2 - step) * cost_;
        }
        oracle_update(p, NLLS_VARS_NEXT, NLLS_VARS_CURRENT, ls, 0);
        cost_ = timed_cost(p, d);
        double mu = (d->bestcost - cost_) / linear_approx;
        if (mu > 0.375) *tr = fmax(*tr, 3 * vnorm(x, n)); else if (mu < 0.125) *tr *= 0.5;
        if (!(cost_ > d->bestcost) || maxabs(x, n) < opt->dstep) return cost_;
    }
}
static double iterate_gd(oracle_problem* p, oracle_ls* ls, idata* d, double* stepsize) { /* iterators.jl:187-208 */
    int64_t n = ls->ndof; double* x = ls->x; const double* g = ls->b;
    for (int64_t i = 0; i < n; ++i) x[i] = -g[i] * *stepsize;
    oracle_update(p, NLLS_VARS_NEXT, NLLS_VARS_CURRENT, ls, 0);
    double costc = timed_cost(p, d);
    while (costc > d->bestcost) {
        double coststep = vdot(x, g, n), costdiff = d->bestcost + coststep - costc;
        *stepsize *= 0.5 * coststep / costdiff;
        for (int64_t i = 0; i < n; ++i) x[i] = -g[i] * *stepsize;
        oracle_update(p, NLLS_VARS_NEXT, NLLS_VARS_CURRENT, ls, 0);
        costc = timed_cost(p, d);
    }
    *stepsize *= 2; return costc;
}
int oracle_optimize(oracle_problem* p, const uint64_t* bi, const oracle_options* opt, oracle_result* res) { /* optimize.jl:5-17,109-180 */
    double starttime = now_s(); memset(res, 0, sizeof *res);
    oracle_ls* ls = oracle_makesymmvls(p, bi, 0);
    memcpy(p->vars[NLLS_VARS_NEXT], p->vars[NLLS_VARS_CURRENT], sizeof(double) * p->nstorage); /* setupiterator :80-82 */
    int have_best = 0;
    idata d; memset(&d, 0, sizeof d);
    double lambda = 0, tr = 0, stepsize = 1.0; double* cauchy = (double*)calloc(ls->ndof > 0 ? ls->ndof : 1, sizeof(double));
    d.startcost = -INFINITY; /* preoptimization :7 */
    int64_t fails = 0; double stoptime = starttime + opt->maxtime; double timeinit = now_s() - starttime;
    double t0 = now_s(); double cost = oracle_costgradhess(p, NLLS_VARS_CURRENT, ls); d.timegradient += now_s() - t0; d.gradientcomputations++; /* :118 */
    d.bestcost = cost; d.startcost = fmax(cost, d.startcost);
    int64_t converged = 0;
    for (;;) {
        d.iternum++;
        switch (opt->iterator) {
        case 0: cost = iterate_newton(p, ls, &d); break;
        case 1: cost = iterate_levmar(p, ls, &d, opt, &lambda); break;
        case 2: cost = iterate_dogleg(p, ls, &d, opt, &tr, cauchy); break;
        default: cost = iterate_gd(p, ls, &d, &stepsize);
        }
        if (opt->store_costs && res->ncosts_stored < 512) res->costs[res->ncosts_stored++] = cost; /* callbacks.jl:63-66 */
        double dcost = d.bestcost - cost; /* :130-147 */
        if (dcost >= 0) { d.bestcost = cost; fails = 0; }
        else { dcost = cost; fails++;
            if (fails == 1) { if (have_best) swapvars(p, NLLS_VARS_CURRENT, NLLS_VARS_BEST);
                              else { memcpy(p->vars[NLLS_VARS_BEST], p->vars[NLLS_VARS_CURRENT], sizeof(double) * p->nstorage); have_best = 1; } } }
        swapvars(p, NLLS_VARS_CURRENT, NLLS_VARS_NEXT); /* updatefromnext! :207-209 */
        double maxstep = maxabs(ls->x, ls->ndof);
        converged = 0; /* :151-161 */
        converged |= (int64_t)(isinf(cost) != 0) << 0;
        converged |= (int64_t)(cost != cost) << 1;
        converged |= (int64_t)(dcost < d.bestcost * opt->reldcost) << 2;
        converged |= (int64_t)(dcost < opt->absdcost) << 3;
        converged |= (int64_t)(isinf(maxstep) != 0) << 4;
        converged |= (int64_t)(maxstep != maxstep) << 5;
        converged |= (int64_t)(maxstep < opt->dstep) << 6;
        converged |= (int64_t)(fails > opt->maxfails) << 7;
        converged |= (int64_t)(d.iternum >= opt->maxiters) << 8;
        converged |= (int64_t)(now_s() > stoptime) << 9;
        if (converged) break;
        t0 = now_s(); cost = oracle_costgradhess(p, NLLS_VARS_CURRENT, ls); d.timegradient += now_s() - t0; d.gradientcomputations++; /* :167-171 */
        (void)cost;
    }
    if (!(d.bestcost >= cost)) swapvars(p, NLLS_VARS_CURRENT, NLLS_VARS_BEST); /* :173-176 */
    res->startcost = d.startcost; res->bestcost = d.bestcost; res->timetotal = now_s() - starttime; res->timeinit = timeinit;
    res->timecost = d.timecost; res->timegradient = d.timegradient; res->timesolver = d.timesolver; res->termination = converged;
    res->niterations = d.iternum; res->costcomputations = d.costcomputations; res->gradientcomputations = d.gradientcomputations; res->linearsolvers = d.linearsolvers;
    free(cauchy); oracle_ls_destroy(ls); return 0;
}
