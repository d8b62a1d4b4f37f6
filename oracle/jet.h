/*
 * oracle/jet.h -- TEST INFRASTRUCTURE (CPU oracle), not part of the product path.
 *
 * Forward-mode dual numbers ("jets") restating what ForwardDiff does for the reference:
 *   - first order, N partials: src/autodiff.jl:48-61 (dualzeros/dualvars), :70-74 (extractvaldual)
 *   - second order, 4 variables: src/autodiff.jl:123-128 (computehessian), :163-165
 *     (autorobustifydcost / autorobustifydkernel)
 * ForwardDiff (compat "0.10, 1", Project.toml:24) is a third-party dependency absent from
 * /root/reference; its published algorithm is plain dual-number arithmetic, restated here.
 */
#ifndef ORACLE_JET_H
#define ORACLE_JET_H

#include <math.h>
#include <string.h>

#define JET_MAXN 16 /* >= largest number of free dof of any registered residual (3+6+3 = 12) */

typedef struct { double v; double d[JET_MAXN]; } jet;

#define JET_INLINE static inline __attribute__((always_inline))

JET_INLINE jet jet_const(double v, int n) { jet r; r.v = v; for (int i = 0; i < n; ++i) r.d[i] = 0.0; return r; }
JET_INLINE jet jet_seed(double v, int k, int n) { jet r = jet_const(v, n); if (k >= 0 && k < n) r.d[k] = 1.0; return r; }
JET_INLINE jet jet_add(jet a, jet b, int n) { jet r; r.v = a.v + b.v; for (int i = 0; i < n; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
JET_INLINE jet jet_sub(jet a, jet b, int n) { jet r; r.v = a.v - b.v; for (int i = 0; i < n; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
JET_INLINE jet jet_mul(jet a, jet b, int n) { jet r; r.v = a.v * b.v; for (int i = 0; i < n; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
JET_INLINE jet jet_div(jet a, jet b, int n) { jet r; double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < n; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
JET_INLINE jet jet_scale(jet a, double s, int n) { jet r; r.v = a.v * s; for (int i = 0; i < n; ++i) r.d[i] = a.d[i] * s; return r; }
JET_INLINE jet jet_addc(jet a, double c, int n) { jet r = a; (void)n; r.v = a.v + c; return r; }
JET_INLINE jet jet_neg(jet a, int n) { return jet_scale(a, -1.0, n); }
JET_INLINE jet jet_exp(jet a, int n) { jet r; r.v = exp(a.v); for (int i = 0; i < n; ++i) r.d[i] = a.d[i] * r.v; return r; }
JET_INLINE jet jet_log(jet a, int n) { jet r; r.v = log(a.v); double ia = 1.0 / a.v; for (int i = 0; i < n; ++i) r.d[i] = a.d[i] * ia; return r; }
JET_INLINE jet jet_sqrt(jet a, int n) { jet r; r.v = sqrt(a.v); double h = 0.5 / r.v; for (int i = 0; i < n; ++i) r.d[i] = a.d[i] * h; return r; }

/* ---- second order, fixed 4 variables (kernel dof 3 + the cost) ------------------------------ */
#define J2N 4
typedef struct { double v; double g[J2N]; double h[J2N][J2N]; } jet2;

JET_INLINE jet2 j2_const(double v) { jet2 r; memset(&r, 0, sizeof r); r.v = v; return r; }
JET_INLINE jet2 j2_seed(double v, int k) { jet2 r = j2_const(v); r.g[k] = 1.0; return r; }
JET_INLINE jet2 j2_add(jet2 a, jet2 b) {
    jet2 r; r.v = a.v + b.v;
    for (int i = 0; i < J2N; ++i) { r.g[i] = a.g[i] + b.g[i]; for (int j = 0; j < J2N; ++j) r.h[i][j] = a.h[i][j] + b.h[i][j]; }
    return r;
}
JET_INLINE jet2 j2_scale(jet2 a, double s) {
    jet2 r; r.v = a.v * s;
    for (int i = 0; i < J2N; ++i) { r.g[i] = a.g[i] * s; for (int j = 0; j < J2N; ++j) r.h[i][j] = a.h[i][j] * s; }
    return r;
}
JET_INLINE jet2 j2_sub(jet2 a, jet2 b) { return j2_add(a, j2_scale(b, -1.0)); }
JET_INLINE jet2 j2_addc(jet2 a, double c) { jet2 r = a; r.v += c; return r; }
JET_INLINE jet2 j2_mul(jet2 a, jet2 b) {
    jet2 r; r.v = a.v * b.v;
    for (int i = 0; i < J2N; ++i) {
        r.g[i] = a.g[i] * b.v + a.v * b.g[i];
        for (int j = 0; j < J2N; ++j)
            r.h[i][j] = a.h[i][j] * b.v + a.g[i] * b.g[j] + a.g[j] * b.g[i] + a.v * b.h[i][j];
    }
    return r;
}
/* f(a) with first/second derivatives f1, f2 of the scalar function at a.v */
JET_INLINE jet2 j2_chain(jet2 a, double f0, double f1, double f2) {
    jet2 r; r.v = f0;
    for (int i = 0; i < J2N; ++i) {
        r.g[i] = f1 * a.g[i];
        for (int j = 0; j < J2N; ++j) r.h[i][j] = f1 * a.h[i][j] + f2 * a.g[i] * a.g[j];
    }
    return r;
}
JET_INLINE jet2 j2_exp(jet2 a) { double e = exp(a.v); return j2_chain(a, e, e, e); }
JET_INLINE jet2 j2_log(jet2 a) { double i = 1.0 / a.v; return j2_chain(a, log(a.v), i, -i * i); }
JET_INLINE jet2 j2_recip(jet2 a) { double i = 1.0 / a.v; return j2_chain(a, i, -i * i, 2.0 * i * i * i); }
JET_INLINE jet2 j2_div(jet2 a, jet2 b) { return j2_mul(a, j2_recip(b)); }
JET_INLINE jet2 j2_sqrt(jet2 a) { double s = sqrt(a.v); return j2_chain(a, s, 0.5 / s, -0.25 / (s * a.v)); }

#endif
