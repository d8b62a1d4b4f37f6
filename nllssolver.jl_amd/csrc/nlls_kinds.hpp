// nlls_kinds.hpp -- device-side block maths for the registered kinds (gfx950 only).
//
// What the reference does per cost block, restated for one GPU lane:
//   * forward-mode AD through update()   src/autodiff.jl:57-61,81-93  -> Dual<N>
//   * variable retractions               src/variable.jl:5,10,22,29-32; src/robustadaptive.jl:12-22
//   * robust kernels                     src/robust.jl:7-77, src/robustadaptive.jl:25-33
//   * adaptive-kernel derivatives        src/autodiff.jl:163-165       -> Dual2 (second order, 4 vars)
// Everything is fully unrolled over compile-time sizes so that the structural zeros / ones of the
// dual seeds are constant-folded by the compiler (the GPU analogue of the reference's StaticInt /
// SVector specialisation).
#pragma once

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>

#include "../../include/nlls_amd.h"

#define NLLS_DEV __device__ __forceinline__
#define NLLS_HD __host__ __device__ __forceinline__

namespace nlls {

// ------------------------------------------------------------------------------------------------
// first-order dual numbers
// ------------------------------------------------------------------------------------------------
template <int N>
struct Dual {
    double v;
    double d[N];
};

template <int N> NLLS_DEV Dual<N> dconst(double v) { Dual<N> r; r.v = v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = 0.0; return r; }
template <int N> NLLS_DEV Dual<N> dseed(double v, int k) { Dual<N> r; r.v = v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (i == k) ? 1.0 : 0.0; return r; }
template <int N> NLLS_DEV Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> NLLS_DEV Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> NLLS_DEV Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> NLLS_DEV Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; double ib = 1.0 / b.v; r.v = a.v * ib;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> NLLS_DEV Dual<N> operator*(const Dual<N>& a, double s) { Dual<N> r; r.v = a.v * s;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> NLLS_DEV Dual<N> operator*(double s, const Dual<N>& a) { return a * s; }
template <int N> NLLS_DEV Dual<N> operator+(const Dual<N>& a, double c) { Dual<N> r = a; r.v = a.v + c; return r; }
template <int N> NLLS_DEV Dual<N> operator-(const Dual<N>& a, double c) { Dual<N> r = a; r.v = a.v - c; return r; }
template <int N> NLLS_DEV Dual<N> operator-(double c, const Dual<N>& a) { Dual<N> r; r.v = c - a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> NLLS_DEV Dual<N> dexp(const Dual<N>& a) { Dual<N> r; r.v = exp(a.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v; return r; }
NLLS_DEV double dexp(double a) { return exp(a); }
NLLS_DEV double dval(double a) { return a; }
template <int N> NLLS_DEV double dval(const Dual<N>& a) { return a.v; }
NLLS_DEV double dpart(double, int) { return 0.0; }
template <int N> NLLS_DEV double dpart(const Dual<N>& a, int i) { return a.d[i]; }

template <class T> struct Lift;                // constant / seeded construction generic in T
template <> struct Lift<double> {
    static NLLS_DEV double c(double v) { return v; }
    static NLLS_DEV double seed(double v, int) { return v; }
    static NLLS_DEV double seedw(double v, int, double) { return v; }
};
template <int N> struct Lift<Dual<N>> {
    static NLLS_DEV Dual<N> c(double v) { return dconst<N>(v); }
    static NLLS_DEV Dual<N> seed(double v, int k) { return dseed<N>(v, k); }
    static NLLS_DEV Dual<N> seedw(double v, int k, double w) { Dual<N> r = dconst<N>(v);
#pragma unroll
        for (int i = 0; i < N; ++i) if (i == k) r.d[i] = w; return r; }
};

// ------------------------------------------------------------------------------------------------
// second-order duals over 4 variables (3 kernel dof + the cost): src/autodiff.jl:123-128,164-165
// ------------------------------------------------------------------------------------------------
struct Dual2 {
    double v, g[4], h[4][4];
};
NLLS_DEV Dual2 d2const(double v) { Dual2 r; r.v = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.g[i] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[i][j] = 0; } return r; }
NLLS_DEV Dual2 d2add(const Dual2& a, const Dual2& b) { Dual2 r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.g[i] = a.g[i] + b.g[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[i][j] = a.h[i][j] + b.h[i][j]; } return r; }
NLLS_DEV Dual2 d2scale(const Dual2& a, double s) { Dual2 r; r.v = a.v * s;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.g[i] = a.g[i] * s;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[i][j] = a.h[i][j] * s; } return r; }
NLLS_DEV Dual2 d2mul(const Dual2& a, const Dual2& b) { Dual2 r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.g[i] = a.g[i] * b.v + a.v * b.g[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[i][j] = a.h[i][j] * b.v + a.g[i] * b.g[j] + a.g[j] * b.g[i] + a.v * b.h[i][j]; } return r; }
NLLS_DEV Dual2 d2chain(const Dual2& a, double f0, double f1, double f2) { Dual2 r; r.v = f0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.g[i] = f1 * a.g[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[i][j] = f1 * a.h[i][j] + f2 * a.g[i] * a.g[j]; } return r; }
NLLS_DEV Dual2 d2exp(const Dual2& a) { double e = exp(a.v); return d2chain(a, e, e, e); }
NLLS_DEV Dual2 d2log(const Dual2& a) { double i = 1.0 / a.v; return d2chain(a, log(a.v), i, -i * i); }
NLLS_DEV Dual2 d2recip(const Dual2& a) { double i = 1.0 / a.v; return d2chain(a, i, -i * i, 2.0 * i * i * i); }

// ------------------------------------------------------------------------------------------------
// second-order duals over N variables: non-squared AbstractCost blocks, computehessian  src/autodiff.jl:123-128,144-159
// ------------------------------------------------------------------------------------------------
template <int N> struct Dual2N { double v, g[N], h[N][N]; };
template <int N> NLLS_DEV Dual2N<N> d2nconst(double v) { Dual2N<N> r; r.v = v;
#pragma unroll
    for (int i = 0; i < N; ++i) { r.g[i] = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) r.h[i][j] = 0.0; } return r; }
template <int N> NLLS_DEV Dual2N<N> operator+(const Dual2N<N>& a, const Dual2N<N>& b) { Dual2N<N> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) { r.g[i] = a.g[i] + b.g[i];
#pragma unroll
        for (int j = 0; j < N; ++j) r.h[i][j] = a.h[i][j] + b.h[i][j]; } return r; }
template <int N> NLLS_DEV Dual2N<N> operator-(const Dual2N<N>& a, const Dual2N<N>& b) { Dual2N<N> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) { r.g[i] = a.g[i] - b.g[i];
#pragma unroll
        for (int j = 0; j < N; ++j) r.h[i][j] = a.h[i][j] - b.h[i][j]; } return r; }
template <int N> NLLS_DEV Dual2N<N> operator*(const Dual2N<N>& a, const Dual2N<N>& b) { Dual2N<N> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) { r.g[i] = a.g[i] * b.v + a.v * b.g[i];
#pragma unroll
        for (int j = 0; j < N; ++j) r.h[i][j] = a.h[i][j] * b.v + a.g[i] * b.g[j] + a.g[j] * b.g[i] + a.v * b.h[i][j]; } return r; }
template <int N> NLLS_DEV Dual2N<N> operator*(const Dual2N<N>& a, double s) { Dual2N<N> r; r.v = a.v * s;
#pragma unroll
    for (int i = 0; i < N; ++i) { r.g[i] = a.g[i] * s;
#pragma unroll
        for (int j = 0; j < N; ++j) r.h[i][j] = a.h[i][j] * s; } return r; }
template <int N> NLLS_DEV Dual2N<N> operator*(double s, const Dual2N<N>& a) { return a * s; }
template <int N> NLLS_DEV Dual2N<N> operator-(const Dual2N<N>& a, double c) { Dual2N<N> r = a; r.v = a.v - c; return r; }
template <int N> NLLS_DEV Dual2N<N> operator+(const Dual2N<N>& a, double c) { Dual2N<N> r = a; r.v = a.v + c; return r; }
template <int N> struct Lift<Dual2N<N>> {     // (seeding through a LINEAR retraction only: cost kinds take Euclidean variables)
    static NLLS_DEV Dual2N<N> c(double v) { return d2nconst<N>(v); }
    static NLLS_DEV Dual2N<N> seed(double v, int k) { Dual2N<N> r = d2nconst<N>(v);
#pragma unroll
        for (int i = 0; i < N; ++i) if (i == k) r.g[i] = 1.0; return r; }
    static NLLS_DEV Dual2N<N> seedw(double v, int k, double w) { Dual2N<N> r = d2nconst<N>(v);
#pragma unroll
        for (int i = 0; i < N; ++i) if (i == k) r.g[i] = w; return r; }
};

// ------------------------------------------------------------------------------------------------
// variable kinds
// ------------------------------------------------------------------------------------------------
NLLS_HD constexpr int var_storage(int kind, int dim) {
    return (kind == NLLS_VAR_EUCLIDEAN || kind == NLLS_VAR_DYNAMIC) ? dim
         : (kind == NLLS_VAR_ZERO_TO_INF || kind == NLLS_VAR_ZERO_TO_ONE) ? 1
         : kind == NLLS_VAR_CONTAMINATED_GAUSSIAN ? 3
         : kind == NLLS_VAR_POSE_SO3 ? 12 : -1;
}
NLLS_HD constexpr int var_dof(int kind, int dim) {   // nvars(): src/variable.jl:4,9,21,28; robustadaptive.jl:21
    return (kind == NLLS_VAR_EUCLIDEAN || kind == NLLS_VAR_DYNAMIC) ? dim
         : (kind == NLLS_VAR_ZERO_TO_INF || kind == NLLS_VAR_ZERO_TO_ONE) ? 1
         : kind == NLLS_VAR_CONTAMINATED_GAUSSIAN ? 3
         : kind == NLLS_VAR_POSE_SO3 ? 6 : -1;
}

NLLS_DEV double zti_update(double v, double d) { return (v > 0 ? v : DBL_MIN) * exp(d); }          // variable.jl:22
NLLS_DEV double zto_update(double v, double d) {                                                   // variable.jl:29-32
    double val = (v > 0 ? v : DBL_MIN) * exp(d);
    return val < INFINITY ? val / (1 + (val - v)) : 1.0;
}
NLLS_DEV void so3_exp(const double* w, double* E) {   // Rodrigues, column-major 3x3 (new kind)
    double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], A, B;
    if (th2 < 1e-12) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; }
    else { double th = sqrt(th2); A = sin(th) / th; B = (1.0 - cos(th)) / th2; }
    const double K[9] = {0, w[2], -w[1], -w[2], 0, w[0], w[1], -w[0], 0};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double k2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) k2 += K[r + 3 * k] * K[k + 3 * c];
            E[r + 3 * c] = (r == c ? 1.0 : 0.0) + A * K[r + 3 * c] + B * k2;
        }
}
// update(var, step): the real retraction (src/linearsystem.jl:206-213 -> src/variable.jl)
NLLS_DEV void var_update_real(int kind, int dim, const double* in, const double* d, double* out) {
    switch (kind) {
    case NLLS_VAR_EUCLIDEAN: for (int i = 0; i < dim; ++i) out[i] = in[i] + d[i]; break;
    case NLLS_VAR_ZERO_TO_INF: out[0] = zti_update(in[0], d[0]); break;
    case NLLS_VAR_ZERO_TO_ONE: out[0] = zto_update(in[0], d[0]); break;
    case NLLS_VAR_CONTAMINATED_GAUSSIAN: {   // robustadaptive.jl:22, then the ordering of :13-15
        double a = zti_update(in[0], d[0]), b = zti_update(in[1], d[1]), w = zto_update(in[2], d[2]);
        if (!(a >= b)) { double t = a; a = b; b = t; }
        out[0] = a; out[1] = b; out[2] = w; break; }
    case NLLS_VAR_POSE_SO3: {
        double E[9], R[9]; so3_exp(d, E);
        for (int i = 0; i < 9; ++i) R[i] = in[i];
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) {
            double s = 0; for (int k = 0; k < 3; ++k) s += R[r + 3 * k] * E[k + 3 * c];
            out[r + 3 * c] = s; }
        for (int i = 0; i < 3; ++i) out[9 + i] = in[9 + i] + d[3 + i];
        break; }
    }
}
// update(var, dualzeros) (src/autodiff.jl:57-61): storage of the variable as T, seeded from `start`
// (start < 0: no partials wanted for this slot).  KIND/DIM are compile-time.
template <int KIND, int DIM, class T>
NLLS_DEV void var_load(const double* v, int start, T* out) {
    using L = Lift<T>;
    if constexpr (KIND == NLLS_VAR_EUCLIDEAN) {
#pragma unroll
        for (int i = 0; i < DIM; ++i) out[i] = L::seed(v[i], start < 0 ? -1 : start + i);
    } else if constexpr (KIND == NLLS_VAR_ZERO_TO_INF) {
        double b = v[0] > 0 ? v[0] : DBL_MIN; out[0] = L::seedw(b, start, b);
    } else if constexpr (KIND == NLLS_VAR_ZERO_TO_ONE) {
        double b = v[0] > 0 ? v[0] : DBL_MIN; T val = L::seedw(b, start, b); out[0] = val / (val + (1.0 - v[0]));
    } else if constexpr (KIND == NLLS_VAR_CONTAMINATED_GAUSSIAN) {
        var_load<NLLS_VAR_ZERO_TO_INF, 1, T>(v + 0, start < 0 ? -1 : start + 0, out + 0);
        var_load<NLLS_VAR_ZERO_TO_INF, 1, T>(v + 1, start < 0 ? -1 : start + 1, out + 1);
        var_load<NLLS_VAR_ZERO_TO_ONE, 1, T>(v + 2, start < 0 ? -1 : start + 2, out + 2);
    } else if constexpr (KIND == NLLS_VAR_POSE_SO3) {
        // R*(I + [d]x): the exact first-order behaviour of R*expm([d]x) at d = 0
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double R0 = v[r], R1 = v[r + 3], R2 = v[r + 6];
            T c0 = L::c(R0), c1 = L::c(R1), c2 = L::c(R2);
            if (start >= 0) {
                c0 = c0 + L::seedw(0.0, start + 2, R1) - L::seedw(0.0, start + 1, R2);
                c1 = c1 - L::seedw(0.0, start + 2, R0) + L::seedw(0.0, start + 0, R2);
                c2 = c2 + L::seedw(0.0, start + 1, R0) - L::seedw(0.0, start + 0, R1);
            }
            out[r] = c0; out[r + 3] = c1; out[r + 6] = c2;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) out[9 + i] = L::seed(v[9 + i], start < 0 ? -1 : start + 3 + i);
    }
}

// ------------------------------------------------------------------------------------------------
// residual kinds: computeresidual() bodies, generic in the scalar type
// ------------------------------------------------------------------------------------------------
constexpr int MAXST = 12;   // largest variable storage (POSE_SO3)
constexpr int MAX_SLOTS = 10;   // variables per cost block: MAX_ARGS of the reference (src/NLLSsolver.jl:28).  The built-in kinds declare SK / SD with four entries; a kind with more slots (a user kind) declares as many as it has

template <int KIND> struct Res;

template <> struct Res<NLLS_RES_BA_AFFINE> {   // test/optimizeba.jl:4 + src/residual.jl:13
    static constexpr int NDEPS = 2, M = 2, NDATA = 2, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, 0, 0};
    static constexpr int SD[4] = {6, 3, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        const T* c = sv[0]; const T* X = sv[1];
        r[0] = c[0] * X[0] + c[1] * X[1] + c[2] * X[2] - data[0];
        r[1] = c[3] * X[0] + c[4] * X[1] + c[5] * X[2] - data[1];
    }
};
template <> struct Res<NLLS_RES_ROSENBROCK_A> {   // test/functional.jl:12
    static constexpr int NDEPS = 1, M = 1, NDATA = 1, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, 0, 0, 0};
    static constexpr int SD[4] = {1, 0, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) { r[0] = (1.0 - sv[0][0]) * data[0]; }
};
template <> struct Res<NLLS_RES_ROSENBROCK_B> {   // test/functional.jl:24
    static constexpr int NDEPS = 2, M = 1, NDATA = 1, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, 0, 0};
    static constexpr int SD[4] = {1, 1, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) { r[0] = (sv[0][0] * sv[0][0] - sv[1][0]) * data[0]; }
};
template <> struct Res<NLLS_RES_ROSENBROCK_2D> {   // examples/rosenbrock.jl:19
    static constexpr int NDEPS = 1, M = 2, NDATA = 2, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, 0, 0, 0};
    static constexpr int SD[4] = {2, 0, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        const T* x = sv[0]; r[0] = (1.0 - x[0]) * data[0]; r[1] = (x[0] * x[0] - x[1]) * data[1]; }
};
template <> struct Res<NLLS_RES_CURVE_EXP4> {   // BASELINE config 2
    static constexpr int NDEPS = 4, M = 1, NDATA = 2, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN};
    static constexpr int SD[4] = {1, 1, 1, 1};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        double t = data[0], y = data[1];
        r[0] = sv[0][0] * dexp(sv[1][0] * t) + sv[2][0] * t + sv[3][0] - y; }
};
template <> struct Res<NLLS_RES_ADAPTIVE_MEAN> {   // test/adaptivecost.jl:11 (sv[] excludes the kernel)
    static constexpr int NDEPS = 2, M = 1, NDATA = 1, ADAPT = 1;
    static constexpr int SK[4] = {NLLS_VAR_CONTAMINATED_GAUSSIAN, NLLS_VAR_EUCLIDEAN, 0, 0};
    static constexpr int SD[4] = {3, 1, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) { r[0] = sv[0][0] - data[0]; }
};
template <class T> NLLS_DEV void pinhole(const double* data, const T* P, const T* X, T* r) {
    T Y[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Y[i] = P[i] * X[0] + P[i + 3] * X[1] + P[i + 6] * X[2] + P[9 + i];
    r[0] = Y[0] / Y[2] - data[0]; r[1] = Y[1] / Y[2] - data[1];
}
// Residual AND Jacobian of the pinhole model in closed form (round 5; the optional `jac` of a residual kind -- BlockGH takes it instead of pushing `eval` through
// Dual<9>, which stays the generic path and the check: nlls_check_analytic).  Tangent of the pose: R <- R expm([w]x), t <- t + tau (var_load / var_update_real), so
//   Y = R X + t,   dY/dw = R [e_k x X] = (R2 X1 - R1 X2 | R0 X2 - R2 X0 | R1 X0 - R0 X1)   (Rk = column k of R),   dY/dtau = I,   dY/dX = R,
//   r = (Y0 / Y2, Y1 / Y2) - data,   dr/dY = [[1, 0, -u], [0, 1, -v]] / Y2   with (u, v) = (Y0, Y1) / Y2:   J = dr/dY [dY/dw | I | R].
NLLS_DEV void pinhole_jac(const double* data, const double* P, const double* X, double* r, double (*J)[9]) {
    double Y[3], Mw[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double R0 = P[i], R1 = P[i + 3], R2 = P[i + 6];
        Y[i] = R0 * X[0] + R1 * X[1] + R2 * X[2] + P[9 + i];
        Mw[i][0] = R2 * X[1] - R1 * X[2]; Mw[i][1] = R0 * X[2] - R2 * X[0]; Mw[i][2] = R1 * X[0] - R0 * X[1];
    }
    const double iz = 1.0 / Y[2], u = Y[0] * iz, v = Y[1] * iz;
    r[0] = u - data[0]; r[1] = v - data[1];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        J[0][k] = iz * (Mw[0][k] - u * Mw[2][k]);         J[1][k] = iz * (Mw[1][k] - v * Mw[2][k]);
        J[0][6 + k] = iz * (P[3 * k] - u * P[3 * k + 2]); J[1][6 + k] = iz * (P[3 * k + 1] - v * P[3 * k + 2]);
    }
    J[0][3] = iz; J[0][4] = 0.0; J[0][5] = -u * iz;
    J[1][3] = 0.0; J[1][4] = iz; J[1][5] = -v * iz;
}
template <> struct Res<NLLS_RES_BA_SO3> {   // new kind
    static constexpr int NDEPS = 2, M = 2, NDATA = 2, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_POSE_SO3, NLLS_VAR_EUCLIDEAN, 0, 0};
    static constexpr int SD[4] = {6, 3, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) { pinhole(data, sv[0], sv[1], r); }
    static NLLS_DEV void jac(const double* data, const double (*st)[MAXST], double* r, double (*J)[9]) { pinhole_jac(data, st[0], st[1], r, J); }
};
template <> struct Res<NLLS_RES_BA_SO3_ADAPTIVE> {   // new kind
    static constexpr int NDEPS = 3, M = 2, NDATA = 2, ADAPT = 1;
    static constexpr int SK[4] = {NLLS_VAR_CONTAMINATED_GAUSSIAN, NLLS_VAR_POSE_SO3, NLLS_VAR_EUCLIDEAN, 0};
    static constexpr int SD[4] = {3, 6, 3, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) { pinhole(data, sv[0], sv[1], r); }
    static NLLS_DEV void jac(const double* data, const double (*st)[MAXST], double* r, double (*J)[9]) { pinhole_jac(data, st[1], st[2], r, J); }   // (st[] includes the kernel's slot)
};

template <> struct Res<NLLS_RES_LINEAR3> {   // test/nonsquaredcost.jl:4-14: X w - y, data = (y[3], X[9] column-major)
    static constexpr int NDEPS = 1, M = 3, NDATA = 12, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, 0, 0, 0};
    static constexpr int SD[4] = {3, 0, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        const T* w = sv[0];
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i] = w[0] * data[3 + i] + w[1] * data[6 + i] + w[2] * data[9 + i] - data[i];
    }
};
template <> struct Res<NLLS_COST_LINEAR3> {   // test/nonsquaredcost.jl:28-37: a NON-SQUARED cost, computecost = y'w (r[0] carries the cost's value)
    static constexpr int NDEPS = 1, M = 1, NDATA = 3, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, 0, 0, 0};
    static constexpr int SD[4] = {3, 0, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        const T* w = sv[0]; r[0] = w[0] * data[0] + w[1] * data[1] + w[2] * data[2];
    }
};
template <> struct Res<NLLS_RES_SCALE_MIX> {   // standalone bounded scalars (src/variable.jl:18-32): s * (w a + (1 - w) b) - y
    static constexpr int NDEPS = 2, M = 1, NDATA = 3, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_ZERO_TO_INF, NLLS_VAR_ZERO_TO_ONE, 0, 0};
    static constexpr int SD[4] = {1, 1, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        const T s = sv[0][0], w = sv[1][0]; r[0] = s * (w * data[0] + (1.0 - w) * data[1]) - data[2];
    }
};
}  // namespace nlls
// user residual kinds, added at BUILD time (include/nlls_amd.h, NLLS_RES_USER0 .. 7): the header specialises nlls::Res<> and defines NLLS_USER_RES(X)
#ifdef NLLS_USER_KINDS_HEADER
#include NLLS_USER_KINDS_HEADER
#endif
#ifndef NLLS_USER_RES
#define NLLS_USER_RES(X)
#endif
namespace nlls {
// kinds whose block is an AbstractCost (value / gradient / Hessian of computecost itself, src/autodiff.jl:144-159), not half a squared residual norm
template <int KIND> constexpr bool is_cost_kind = (KIND == NLLS_COST_LINEAR3);

// compile-time helpers over a residual kind
template <int KIND> struct ResInfo {
    using R = Res<KIND>;
    static constexpr int NS = R::NDEPS - R::ADAPT;                     // slots seen by computeresidual
    static constexpr int dof(int s) { return var_dof(R::SK[s], R::SD[s]); }
    static constexpr int sto(int s) { return var_storage(R::SK[s], R::SD[s]); }
    static constexpr int np() { int n = 0; for (int s = R::ADAPT; s < R::NDEPS; ++s) n += dof(s); return n; }
    static constexpr int NP = np();                                    // free dof of the non-kernel slots
    // start of slot s among the non-kernel partials
    static constexpr int joff(int s) { int n = 0; for (int t = R::ADAPT; t < s; ++t) n += dof(t); return n; }
};

// ------------------------------------------------------------------------------------------------
// robust kernels
// ------------------------------------------------------------------------------------------------
struct RobustSpec { int kind; double p0, p1; };

NLLS_DEV double robustify_fixed(const RobustSpec& k, double cost) {   // src/robust.jl:11,26,47,71
    int base = k.kind & 0xF; double c;
    if (base == NLLS_ROBUST_HUBER || base == NLLS_ROBUST_HUBER2O) { double w2 = k.p0 * k.p0; c = cost < w2 ? cost : sqrt(cost) * (k.p0 * 2) - w2; }
    else if (base == NLLS_ROBUST_GEMAN_MCCLURE) { double w2 = k.p0 * k.p0; c = cost * w2 / (cost + w2); }
    else c = cost;
    if (k.kind & NLLS_ROBUST_SCALED) c *= k.p1;
    return c;
}
NLLS_DEV void robustifydcost_fixed(const RobustSpec& k, double cost, double& rho, double& d1, double& d2) {   // src/robust.jl:12,28-31,48-55,72-77
    int base = k.kind & 0xF;
    if (base == NLLS_ROBUST_HUBER || base == NLLS_ROBUST_HUBER2O) {
        double w = k.p0, w2 = w * w;
        if (cost < w2) { rho = cost; d1 = 1; d2 = 0; }
        else { double sq = sqrt(cost); rho = sq * (w * 2) - w2; d1 = w / sq; d2 = base == NLLS_ROBUST_HUBER2O ? (-0.5 * w) / (cost * sq) : 0.0; }
    } else if (base == NLLS_ROBUST_GEMAN_MCCLURE) {
        double w2 = k.p0 * k.p0, r = 1.0 / (cost + w2), w = w2 * r, ww = w * w;
        rho = cost * w; d1 = ww; d2 = -2 * ww * r;
    } else { rho = cost; d1 = 1; d2 = 0; }
    if (k.kind & NLLS_ROBUST_SCALED) { rho *= k.p1; d1 *= k.p1; d2 *= k.p1; }
}
// ContaminatedGaussian from storage (1/s1, 1/s2, w)   src/robustadaptive.jl:12-33
NLLS_DEV double cg_robustify(const double* st, double cost) {
    double s1sq = st[0] * st[0], s2sq = st[1] * st[1], hd = 0.5 * (s2sq - s1sq), hs2 = 0.5 * s2sq;
    return cost * hs2 - log(st[2] * st[0] * exp(cost * hd) + (1 - st[2]) * st[1]);
}
NLLS_DEV void cg_robustifydcost(const double* st, double cost, double& rho, double& d1, double& d2) {
    double s1sq = st[0] * st[0], s2sq = st[1] * st[1], hd = 0.5 * (s2sq - s1sq), hs2 = 0.5 * s2sq;
    double c = cost * hs2, s = st[2] * st[0] * exp(cost * hd), t = (1 - st[2]) * st[1], den = 1 / (s + t);
    s *= hd;
    rho = c + log(den); d1 = hs2 - s * den; d2 = -s * hd * t * den * den;
}
// autorobustifydkernel  src/autodiff.jl:164-165: value/gradient/hessian w.r.t. (kernel dof 1..3, cost)
NLLS_DEV Dual2 cg_robustifydkernel(const double* st, double cost) {
    double b0 = st[0] > 0 ? st[0] : DBL_MIN, b1 = st[1] > 0 ? st[1] : DBL_MIN, b2 = st[2] > 0 ? st[2] : DBL_MIN;
    Dual2 is1 = d2const(b0); is1.g[0] = b0; is1.h[0][0] = b0;            // ZeroToInf: b*exp(x)
    Dual2 is2 = d2const(b1); is2.g[1] = b1; is2.h[1][1] = b1;
    Dual2 val = d2const(b2); val.g[2] = b2; val.h[2][2] = b2;            // ZeroToOne: val/(1+(val-v))
    Dual2 den = val; den.v += 1.0 - st[2];
    Dual2 w = d2mul(val, d2recip(den));
    Dual2 c = d2const(cost); c.g[3] = 1.0;
    Dual2 s1sq = d2mul(is1, is1), s2sq = d2mul(is2, is2);
    Dual2 hd = d2scale(d2add(s2sq, d2scale(s1sq, -1.0)), 0.5), hs2 = d2scale(s2sq, 0.5);
    Dual2 a = d2mul(d2mul(w, is1), d2exp(d2mul(c, hd)));
    Dual2 omw = d2scale(w, -1.0); omw.v += 1.0;
    Dual2 b = d2mul(omw, is2);
    return d2add(d2mul(c, hs2), d2scale(d2log(d2add(a, b)), -1.0));
}

// The same value / gradient / Hessian in closed form (round 5): the second-order duals above carry 21 doubles through every operation -- twelve d2mul's,
// most of the registers of the adaptive kinds' accumulate kernels.  With  u = log a,  v = log b  (a = w is1 exp(c hd), b = (1 - w) is2, S = a + b):
//   rho = c hs2 - log S,   (log S)_i = pa u_i + pb v_i,   (log S)_ij = pa (u_i u_j + u_ij) + pb (v_i v_j + v_ij) - (log S)_i (log S)_j,   pa = a / S, pb = b / S
// and u, v are sums of terms in ONE variable each (x0, x1, x2 the tangent steps of (1/s1, 1/s2, w) through their retractions at 0, x3 = c), so their
// derivatives are a handful of scalars.  Kept beside the dual-number version, which is the generic statement of src/autodiff.jl:164-165 and the
// check: nlls_check_analytic compares the two on the device (tests/test_gpu_parity.py).
NLLS_DEV Dual2 cg_robustifydkernel_closed(const double* st, double c) {
    const double b0 = st[0] > 0 ? st[0] : DBL_MIN, b1 = st[1] > 0 ? st[1] : DBL_MIN, b2 = st[2] > 0 ? st[2] : DBL_MIN;
    const double w = b2 / (b2 + (1.0 - st[2])), omw = 1.0 - w, wo = w * omw;      // ZeroToOne: val / (1 + (val - v)) at step 0; d w / d x2 = w (1 - w)
    const double s1 = b0 * b0, s2 = b1 * b1, hd = 0.5 * (s2 - s1), hs2 = 0.5 * s2;
    const double a = w * b0 * exp(c * hd), b = omw * b1, iS = 1.0 / (a + b), pa = a * iS, pb = b * iS;
    // u = log w + log is1 + c hd;  v = log(1 - w) + log is2
    const double u0 = 1.0 - c * s1, u1 = c * s2, u2 = omw, u3 = hd;
    const double u00 = -2.0 * c * s1, u11 = 2.0 * c * s2, u22 = -wo, u03 = -s1, u13 = s2;
    const double v1 = 1.0, v2 = -w, v22 = -wo;
    const double L0 = pa * u0, L1 = pa * u1 + pb * v1, L2 = pa * u2 + pb * v2, L3 = pa * u3;
    Dual2 k;
    k.v = c * hs2 - log(a + b);
    k.g[0] = -L0; k.g[1] = c * s2 - L1; k.g[2] = -L2; k.g[3] = hs2 - L3;
    const double h00 = -(pa * (u0 * u0 + u00) - L0 * L0);
    const double h01 = -(pa * (u0 * u1) - L0 * L1);
    const double h02 = -(pa * (u0 * u2) - L0 * L2);
    const double h03 = -(pa * (u0 * u3 + u03) - L0 * L3);
    const double h11 = 2.0 * c * s2 - (pa * (u1 * u1 + u11) + pb * (v1 * v1) - L1 * L1);
    const double h12 = -(pa * (u1 * u2) + pb * (v1 * v2) - L1 * L2);
    const double h13 = s2 - (pa * (u1 * u3 + u13) - L1 * L3);
    const double h22 = -(pa * (u2 * u2 + u22) + pb * (v2 * v2 + v22) - L2 * L2);
    const double h23 = -(pa * (u2 * u3) - L2 * L3);
    const double h33 = -(pa * (u3 * u3) - L3 * L3);
    k.h[0][0] = h00; k.h[0][1] = h01; k.h[0][2] = h02; k.h[0][3] = h03;
    k.h[1][0] = h01; k.h[1][1] = h11; k.h[1][2] = h12; k.h[1][3] = h13;
    k.h[2][0] = h02; k.h[2][1] = h12; k.h[2][2] = h22; k.h[2][3] = h23;
    k.h[3][0] = h03; k.h[3][1] = h13; k.h[3][2] = h23; k.h[3][3] = h33;
    return k;
}

// ------------------------------------------------------------------------------------------------
// per-block evaluation  (src/residual.jl:44-111, src/autodiff.jl:81-93)
// ------------------------------------------------------------------------------------------------
// Residual value only: computerescost  src/residual.jl:49-55.  voff[s] = storage offset of slot s.
template <int KIND>
NLLS_DEV double block_cost(const double* __restrict__ vars, const uint32_t* voff, const double* data, const RobustSpec& rk) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    double sv[I::NS > 0 ? I::NS : 1][MAXST];
    [&]<int... S>(std::integer_sequence<int, S...>) {
        (var_load<R::SK[S + R::ADAPT], R::SD[S + R::ADAPT], double>(vars + voff[S + R::ADAPT], -1, sv[S]), ...);
    }(std::make_integer_sequence<int, I::NS>{});
    double r[R::M]; R::template eval<double>(data, sv, r);
    if constexpr (is_cost_kind<KIND>) return r[0];              // computecost: the value itself (src/cost.jl, AbstractCost)
    double s = 0;
#pragma unroll
    for (int m = 0; m < R::M; ++m) s += r[m] * r[m];
    if constexpr (R::ADAPT) return 0.5 * cg_robustify(vars + voff[0], s);
    else return 0.5 * robustify_fixed(rk, s);
}

// ... the same from variable storage already in registers (one row per getvars() slot, kernel included): the matrix-free trial's back-substitution forms the trial
// point's variables itself and takes their cost in the same pass
template <int KIND>
NLLS_DEV double block_cost_st(const double (*st)[MAXST], const double* data, const RobustSpec& rk) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    double sv[I::NS > 0 ? I::NS : 1][MAXST];
    [&]<int... S>(std::integer_sequence<int, S...>) {
        (var_load<R::SK[S + R::ADAPT], R::SD[S + R::ADAPT], double>(st[S + R::ADAPT], -1, sv[S]), ...);
    }(std::make_integer_sequence<int, I::NS>{});
    double r[R::M]; R::template eval<double>(data, sv, r);
    if constexpr (is_cost_kind<KIND>) return r[0];
    double s = 0;
#pragma unroll
    for (int m = 0; m < R::M; ++m) s += r[m] * r[m];
    if constexpr (R::ADAPT) return 0.5 * cg_robustify(st[0], s);
    else return 0.5 * robustify_fixed(rk, s);
}

// Everything the accumulate kernels need from one block, with ALL variables treated as free
// (the reference's varflags specialisations only drop rows/columns of g and H; the kept entries
// are identical -- src/residual.jl:57-111).
template <class R> struct ResJacCols { static constexpr int value = [] { int n = 0; for (int s = R::ADAPT; s < R::NDEPS; ++s) n += var_dof(R::SK[s], R::SD[s]); return n > 0 ? n : 1; }(); };
// does the residual kind bring its own Jacobian (`jac`: residual and Jacobian w.r.t. the tangent of update() in closed form)?
template <class R> concept HasJac = requires(const double* d, const double (*st)[MAXST], double* r, double (*J)[ResJacCols<R>::value]) { R::jac(d, st, r, J); };
// ANALYTIC = false: everything through dual numbers (src/autodiff.jl as written) -- the reference statement the closed forms are checked against (nlls_check_analytic)
template <int KIND, bool ANALYTIC = true>
struct BlockGH {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    static constexpr int NP = I::NP, M = R::M;
    double J[M][NP];        // jacobian
    double Jw[M][NP], gw[NP];   // rho' J and 2 rho'' g: H(i, j) = sum_m Jw[m][i] J[m][j] + gw[i] g[j]  (residual.jl:91-101 with the common factors taken once per block)
    double g[NP];           // UN-weighted J'r (residual.jl:73)
    double cost;            // 0.5 * rho   (residual.jl:110)
    double dc, d2c;         // rho', rho'' (residual.jl:78 or :82-84)
    double dck[3];          // d rho / d kernel   (adaptive, kernel optimised)
    double d2ck[3][4];      // d2 rho / d kernel d(kernel, cost)
    double hc[is_cost_kind<KIND> ? NP : 1][is_cost_kind<KIND> ? NP : 1];   // AbstractCost kinds: the Hessian of the cost (g carries its gradient)

    // raw storage of the block's variables (one row per getvars() slot, kernel included), so that callers can issue
    // the gathers of several blocks before evaluating any of them
    static constexpr int NSLOTS = R::NDEPS;
    static NLLS_DEV void load(const double* __restrict__ vars, const uint32_t* voff, double (*st)[MAXST]) {
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
#pragma unroll
                for (int q = 0; q < var_storage(R::SK[S], R::SD[S]); ++q) st[S][q] = vars[voff[S] + q]; }(), ...);
        }(std::make_integer_sequence<int, R::NDEPS>{});
    }
    // the same without slot SKIP (a heavy tile's own variable is the same for every entry: gathered once per tile)
    template <int SKIP>
    static NLLS_DEV void load_skip(const double* __restrict__ vars, const uint32_t* voff, double (*st)[MAXST]) {
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
                if constexpr (S != SKIP) {
#pragma unroll
                    for (int q = 0; q < var_storage(R::SK[S], R::SD[S]); ++q) st[S][q] = vars[voff[S] + q]; } }(), ...);
        }(std::make_integer_sequence<int, R::NDEPS>{});
    }
    template <int ONLY>
    static NLLS_DEV void load_only(const double* __restrict__ vars, const uint32_t* voff, double (*st)[MAXST]) {
#pragma unroll
        for (int q = 0; q < var_storage(R::SK[ONLY], R::SD[ONLY]); ++q) st[ONLY][q] = vars[voff[ONLY] + q];
    }
    template <int ONLY>
    static NLLS_DEV void copy_only(const double (*from)[MAXST], double (*st)[MAXST]) {
#pragma unroll
        for (int q = 0; q < var_storage(R::SK[ONLY], R::SD[ONLY]); ++q) st[ONLY][q] = from[ONLY][q];
    }
    // forces the gathered values into registers at this point of the program, i.e. places their s_waitcnt here
    static NLLS_DEV void pin(double (*st)[MAXST]) {
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
#pragma unroll
                for (int q = 0; q < var_storage(R::SK[S], R::SD[S]); ++q) { double v = st[S][q]; asm volatile("" : "+v"(v)); st[S][q] = v; } }(), ...);
        }(std::make_integer_sequence<int, R::NDEPS>{});
    }
    NLLS_DEV void compute(const double* __restrict__ vars, const uint32_t* voff, const double* data, const RobustSpec& rk, bool kernel_free) {
        double st[R::NDEPS][MAXST]; load(vars, voff, st); compute_st(st, data, rk, kernel_free);
    }
    NLLS_DEV void compute_st(const double (*st)[MAXST], const double* data, const RobustSpec& rk, bool kernel_free) {
        if constexpr (is_cost_kind<KIND>) {                     // computecostgradhess, src/autodiff.jl:144-159: second-order duals through update()
            using T2 = Dual2N<NP>;
            T2 sv2[I::NS > 0 ? I::NS : 1][MAXST];
            [&]<int... S>(std::integer_sequence<int, S...>) {
                (var_load<R::SK[S], R::SD[S], T2>(st[S], I::joff(S), sv2[S]), ...);
            }(std::make_integer_sequence<int, I::NS>{});
            T2 c2[1]; R::template eval<T2>(data, sv2, c2);
            cost = c2[0].v; dc = 1.0; d2c = 0.0;
#pragma unroll
            for (int i = 0; i < NP; ++i) { g[i] = c2[0].g[i];
#pragma unroll
                for (int j = 0; j < NP; ++j) hc[i][j] = c2[0].h[i][j]; }
            return;
        }
        double rv[M];
        if constexpr (ANALYTIC && HasJac<R>) {
            R::jac(data, st, rv, J);
        } else {
            using T = Dual<NP>;
            T sv[I::NS > 0 ? I::NS : 1][MAXST];
            [&]<int... S>(std::integer_sequence<int, S...>) {
                (var_load<R::SK[S + R::ADAPT], R::SD[S + R::ADAPT], T>(st[S + R::ADAPT], I::joff(S + R::ADAPT), sv[S]), ...);
            }(std::make_integer_sequence<int, I::NS>{});
            T r[M]; R::template eval<T>(data, sv, r);
#pragma unroll
            for (int m = 0; m < M; ++m) { rv[m] = r[m].v;
#pragma unroll
                for (int j = 0; j < NP; ++j) J[m][j] = r[m].d[j]; }
        }
        double c = 0;
#pragma unroll
        for (int m = 0; m < M; ++m) c += rv[m] * rv[m];
#pragma unroll
        for (int j = 0; j < NP; ++j) { double s = 0;
#pragma unroll
            for (int m = 0; m < M; ++m) s += J[m][j] * rv[m]; g[j] = s; }
        double rho;
        if constexpr (R::ADAPT) {
            if (kernel_free) {                                  // residual.jl:79-88
                Dual2 k; if constexpr (ANALYTIC) k = cg_robustifydkernel_closed(st[0], c); else k = cg_robustifydkernel(st[0], c);
                rho = k.v; dc = k.g[3]; d2c = k.h[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { dck[i] = k.g[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) d2ck[i][j] = k.h[i][j]; }
            } else cg_robustifydcost(st[0], c, rho, dc, d2c);   // residual.jl:76-78
        } else robustifydcost_fixed(rk, c, rho, dc, d2c);
        cost = 0.5 * rho;
        const double d2c2 = 2 * d2c;
#pragma unroll
        for (int j = 0; j < NP; ++j) { gw[j] = d2c2 * g[j];
#pragma unroll
            for (int m = 0; m < M; ++m) Jw[m][j] = dc * J[m][j]; }
    }
    // local H / g entries over the non-kernel dof (residual.jl:91-101)
    NLLS_DEV double H(int i, int j) const {
        if constexpr (is_cost_kind<KIND>) return hc[i][j];
        double s = gw[i] * g[j];
#pragma unroll
        for (int m = 0; m < M; ++m) s += Jw[m][i] * J[m][j];
        return s; }
    NLLS_DEV double G(int i) const { return g[i] * dc; }
    // kernel border (residual.jl:86-88,103-107): d2/dkernel_k dvar_i, d2/dkernel^2, d/dkernel
    NLLS_DEV double Hkv(int k, int i) const { return g[i] * d2ck[k][3]; }
    NLLS_DEV double Hkk(int k, int l) const { return d2ck[k][l]; }
    NLLS_DEV double Gk(int k) const { return dck[k]; }
};

}  // namespace nlls
